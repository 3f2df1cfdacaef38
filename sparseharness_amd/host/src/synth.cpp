// synth.cpp -- seeded synthetic CSR generators for the benchmark configs of
// BASELINE.json: power-law rows / uniform columns and Graph500
// R-MAT.  The reference has no generator (it only reads MatrixMarket files,
// src/sparse_matrix.cpp:11-70); these produce, directly in memory, the CSR view
// its row builder would produce from an equivalent file: duplicates kept,
// integer-valued weights in [1,16] (so the reference's int truncation,
// src/sparse_matrix.cpp:107, is a no-op).
//
// Every random draw is a pure function of (seed, index) (splitmix64 counter
// hash), so results do not depend on the number of OpenMP threads.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <vector>

#include "sh_host.h"

static inline uint64_t mix(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static inline uint64_t h2(uint64_t seed, uint64_t i) { return mix(mix(seed) ^ (i * 0xD1342543DE82EF95ull + 1)); }
static inline double u01(uint64_t h) { return (double)(h >> 11) * (1.0 / 9007199254740992.0); }

extern "C" {

// Power-law row degrees P(d) ~ d^-exponent on [1, dmax], rescaled so that the
// degrees sum to exactly nnz; uniform columns; rows in random (hash) order.
int sh_synth_powerlaw(int64_t rows, int64_t cols, int64_t nnz, double exponent, int64_t dmax,
                      uint64_t seed, int32_t *row_ptr, int32_t *col_idx, float *val) {
  if (rows <= 0 || cols <= 0 || nnz < 0 || !row_ptr || (nnz && (!col_idx || !val)) || exponent <= 1.0)
    return -1;
  if (dmax < 1) dmax = 1;
  std::vector<double> raw(rows);
  const double a1 = 1.0 - exponent, top = std::pow((double)dmax, a1);
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < rows; r++) {
    double u = u01(h2(seed, (uint64_t)r));
    raw[r] = std::pow((top - 1.0) * u + 1.0, 1.0 / a1);  // inverse CDF, in [1, dmax]
  }
  // find the scale whose floored degrees sum to <= nnz, then hand out the remainder
  auto total = [&](double sc) {
    int64_t t = 0;
#pragma omp parallel for reduction(+ : t) schedule(static)
    for (int64_t r = 0; r < rows; r++)
      t += (int64_t)std::floor(raw[r] * sc);
    return t;
  };
  double lo = 0.0, hi = 1.0;
  while (total(hi) < nnz) hi *= 2.0;
  for (int it = 0; it < 80; it++) {
    double mid = 0.5 * (lo + hi);
    if (total(mid) <= nnz) lo = mid; else hi = mid;
  }
  std::vector<int64_t> deg(rows);
  int64_t sum = 0;
  for (int64_t r = 0; r < rows; r++) { deg[r] = (int64_t)std::floor(raw[r] * lo); sum += deg[r]; }
  int64_t rem = nnz - sum;           // >= 0
  for (int64_t k = 0; rem > 0; k++) { // spread the remainder over hashed rows
    int64_t r = (int64_t)(h2(seed ^ 0xABCDEFull, (uint64_t)k) % (uint64_t)rows);
    deg[r]++; rem--;
  }
  row_ptr[0] = 0;
  for (int64_t r = 0; r < rows; r++) {
    int64_t nx = (int64_t)row_ptr[r] + deg[r];
    if (nx > INT32_MAX) return -2;
    row_ptr[r + 1] = (int32_t)nx;
  }
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < nnz; i++) {
    col_idx[i] = (int32_t)(h2(seed + 1, (uint64_t)i) % (uint64_t)cols);
    val[i] = (float)(1 + (int)(h2(seed + 2, (uint64_t)i) & 15));
  }
  return 0;
}

// Graph500 R-MAT: 2^scale vertices, edge_factor * 2^scale directed edges,
// quadrant probabilities (a,b,c,1-a-b-c), vertex ids permuted by a seeded
// Fisher-Yates (seed+1) when permute != 0.  Duplicate edges are kept.  Entries
// of a row are ordered by (column, weight) so the result is deterministic.
int sh_synth_rmat(int scale, int edge_factor, double a, double b, double c, uint64_t seed,
                  int permute, int32_t *row_ptr, int32_t *col_idx, float *val) {
  if (scale < 1 || scale > 30 || edge_factor < 1 || !row_ptr || !col_idx || !val)
    return -1;
  const int64_t n = (int64_t)1 << scale, m = n * edge_factor;
  if (m > INT32_MAX - 8) return -2;
  std::vector<int32_t> perm(n);
  std::iota(perm.begin(), perm.end(), 0);
  if (permute)
    for (int64_t i = n - 1; i > 0; i--) {
      int64_t j = (int64_t)(h2(seed + 1, (uint64_t)i) % (uint64_t)(i + 1));
      std::swap(perm[i], perm[j]);
    }
  const double ab = a + b, abc = a + b + c;
  std::vector<uint64_t> key(m);   // (col << 32) | weight bits, grouped by row below
  std::vector<int32_t> erow(m);
  std::vector<int32_t> cnt(n + 1, 0);
#pragma omp parallel for schedule(static)
  for (int64_t e = 0; e < m; e++) {
    int64_t i = 0, j = 0;
    for (int l = 0; l < scale; l++) {
      double u = u01(h2(seed, (uint64_t)e * 64 + (uint64_t)l));
      int ib = u >= ab, jb = (u >= a && u < ab) || u >= abc;
      i = (i << 1) | ib;
      j = (j << 1) | jb;
    }
    int32_t r = perm[i], cc = perm[j];
    float w = (float)(1 + (int)(h2(seed + 2, (uint64_t)e) & 15));
    uint32_t wb;
    memcpy(&wb, &w, 4);
    erow[e] = r;
    key[e] = ((uint64_t)(uint32_t)cc << 32) | wb;
#pragma omp atomic
    cnt[r + 1]++;
  }
  row_ptr[0] = 0;
  for (int64_t r = 0; r < n; r++) row_ptr[r + 1] = row_ptr[r] + cnt[r + 1];
  std::vector<uint64_t> sorted(m);
  std::vector<int32_t> fill(n, 0);
  for (int64_t e = 0; e < m; e++)   // serial scatter keeps memory traffic simple; order fixed by sort below
    sorted[(int64_t)row_ptr[erow[e]] + fill[erow[e]]++] = key[e];
#pragma omp parallel for schedule(dynamic, 1024)
  for (int64_t r = 0; r < n; r++)
    std::sort(sorted.begin() + row_ptr[r], sorted.begin() + row_ptr[r + 1]);
#pragma omp parallel for schedule(static)
  for (int64_t p = 0; p < m; p++) {
    col_idx[p] = (int32_t)(sorted[p] >> 32);
    uint32_t wb = (uint32_t)sorted[p];
    memcpy(&val[p], &wb, 4);
  }
  return 0;
}

} // extern "C"
