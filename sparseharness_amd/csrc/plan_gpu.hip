// plan_gpu.hip -- the x-tiled plan's layout built ON THE DEVICE from the CSR arrays (VERDICT r2 "missing" #4; the
// reference's counterpart is SparseMatrix::cl_encode on the host, src/sparse_matrix.cpp:122-399).
//
// The host builder (plan_host.h::build_tiled_plan) walks the matrix bin by bin; here the same layout falls out of
// sorts and scans (rocPRIM) plus one thread per entry / piece / strip:
//
//   A-order  all entries sorted by (row, tile), stable: a (row, tile) RUN is contiguous, its entries in CSR order.
//            Per entry: index inside its run, the run's length, its ROLE (first / second entry of a folded pair,
//            single, or entry of a heavy row).  The exclusive scan of "this entry stands for a product" IS the light
//            row-offset array (lrp) and, minus the bin's first product, the slot of every product in its bin image.
//   bins     greedy cut over lrp on the host (10 M steps, sequential by nature), the only O(rows) host work.
//   T-order  the light entries of A-order sorted by tile, stable: (tile, row, CSR order) = the order of the stream.
//            A (bin, tile) PIECE is contiguous; two scans give every entry its rank among the piece's pairs / singles,
//            and the piece's layout (pack_piece) then fixes its stream position and product index in closed form.
//   pieces   scans over the piece table in (tile, bin) order give stream / P positions, in (bin, tile) order the
//            bin-major tables (pslot, ptab, gblk).
//   heavy    the runs of heavy rows are the (row, tile) cells; sorted by tile they get their place in the tile's
//            heavy run, which fixes the wave boundaries and hence the partial slots (gdest).
//
// The result equals the host builder's byte for byte (tests/test_parity_gpu.py compares every array), so everything
// the host builder's emulator and parity tests establish holds for it too.
#include "plan_common.h"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>

#include <cstring>
#include <string>
#include <vector>

namespace sh {

void TiledDevArrays::release() {
  for (void *p : {(void *)tcol, (void *)pslot, (void *)tcode, (void *)tval, (void *)gdest, (void *)gblk, (void *)obase, (void *)lrp, (void *)ptab, (void *)ptile})
    if (p) (void)hipFree(p);
  *this = TiledDevArrays();
}

namespace {

constexpr int PBS = 256;
// rocPRIM's device-wide primitives under the names the builder uses (two-call convention: temporary storage size first)
template <class In, class Out>
hipError_t excl_sum(void *t, size_t &b, In in, Out out, int64_t n, hipStream_t s) {
  using T = typename std::iterator_traits<Out>::value_type;
  return rocprim::exclusive_scan(t, b, in, out, T(0), (size_t)n, rocprim::plus<T>(), s);
}
template <class In, class Out>
hipError_t incl_sum(void *t, size_t &b, In in, Out out, int64_t n, hipStream_t s) {
  using T = typename std::iterator_traits<Out>::value_type;
  return rocprim::inclusive_scan(t, b, in, out, (size_t)n, rocprim::plus<T>(), s);
}
template <class In, class Out>
hipError_t incl_max(void *t, size_t &b, In in, Out out, int64_t n, hipStream_t s) {
  using T = typename std::iterator_traits<Out>::value_type;
  return rocprim::inclusive_scan(t, b, in, out, (size_t)n, rocprim::maximum<T>(), s);
}
template <class K, class V>
hipError_t sort_pairs(void *t, size_t &b, const K *kin, K *kout, const V *vin, V *vout, int64_t n, int bit0, int bit1, hipStream_t s) {
  return rocprim::radix_sort_pairs(t, b, kin, kout, vin, vout, (size_t)n, (unsigned)bit0, (unsigned)bit1, s);
}
constexpr uint32_t ROLE_SINGLE = 0, ROLE_FIRST = 1, ROLE_SECOND = 2, ROLE_HEAVY = 3;
inline dim3 grid_for(int64_t n) { return dim3((unsigned)std::max<int64_t>(1, (n + PBS - 1) / PBS)); }
inline int bits_for(uint64_t maxval) { int b = 1; while (b < 64 && (maxval >> b) != 0) b++; return b; }
#define GID ((int64_t)blockIdx.x * PBS + threadIdx.x)

// temporaries: freed when the builder returns
struct DevPool {
  std::vector<void *> ptrs;
  hipError_t err = hipSuccess;
  template <class T> T *get(size_t n, bool zero = false) {
    void *p = nullptr;
    if (err != hipSuccess) return nullptr;
    const size_t bytes = std::max<size_t>(n, 1) * sizeof(T) + 64;
    err = hipMalloc(&p, bytes);
    if (err != hipSuccess) return nullptr;
    ptrs.push_back(p);
    if (zero) err = hipMemsetAsync(p, 0, bytes, stream);
    return (T *)p;
  }
  hipStream_t stream = nullptr;
  ~DevPool() { for (void *p : ptrs) (void)hipFree(p); }
};

__global__ void k_row_heads(const int32_t *__restrict__ rp, int64_t rows, uint32_t *__restrict__ head) {
  const int64_t r = GID;
  if (r < rows && rp[r + 1] > rp[r]) head[rp[r]] = (uint32_t)r;
}
__global__ void k_iota(uint32_t *__restrict__ v, int64_t n) {
  const int64_t i = GID;
  if (i < n) v[i] = (uint32_t)i;
}
__global__ void k_iota64(uint64_t *__restrict__ v, int64_t n) {
  const int64_t i = GID;
  if (i < n) v[i] = (uint64_t)i;
}
__global__ void k_fill16(uint16_t *__restrict__ v, int64_t n, uint16_t x) {
  const int64_t i = GID;
  if (i < n) v[i] = x;
}
// key = row << tb | tile (out-of-range columns: tile 0, as the host's tile_of)
__global__ void k_keys(const uint32_t *__restrict__ row_of, const int32_t *__restrict__ ci, int64_t n, int64_t cols, int tb,
                       uint64_t *__restrict__ key, uint32_t *__restrict__ val) {
  const int64_t i = GID;
  if (i >= n) return;
  const int32_t c = ci[i];
  const uint32_t t = ((uint32_t)c < (uint32_t)cols) ? (uint32_t)(c / TCOLS) : 0u;
  key[i] = ((uint64_t)row_of[i] << tb) | t;
  val[i] = (uint32_t)i;
}
__global__ void k_run_starts(const uint64_t *__restrict__ key, int64_t n, uint32_t *__restrict__ start_idx) {
  const int64_t i = GID;
  if (i < n) start_idx[i] = (i == 0 || key[i] != key[i - 1]) ? (uint32_t)i : 0u;
}
// role of every entry; rep[i] = the entry stands for a light product; hstart[i] = first entry of a heavy (row, tile) cell;
// klen[run start] = entries of the run
__global__ void k_roles(const uint64_t *__restrict__ key, const uint32_t *__restrict__ rs, const int32_t *__restrict__ rp, int64_t n,
                        int tb, int64_t heavy_thr, int fold, uint32_t *__restrict__ rep, uint32_t *__restrict__ hstart,
                        uint32_t *__restrict__ klen, uint8_t *__restrict__ role_out) {
  const int64_t i = GID;
  if (i >= n) return;
  const uint64_t k = key[i];
  const int64_t row = (int64_t)(k >> tb);
  const bool heavy = (int64_t)rp[row + 1] - rp[row] >= heavy_thr;
  const uint32_t idx = (uint32_t)i - rs[i];
  const bool next_same = i + 1 < n && key[i + 1] == k;
  uint32_t role;
  if (heavy) role = ROLE_HEAVY;
  else if (!fold) role = ROLE_SINGLE;
  else if (idx & 1u) role = ROLE_SECOND;
  else role = next_same ? ROLE_FIRST : ROLE_SINGLE;
  rep[i] = (role == ROLE_FIRST || role == ROLE_SINGLE) ? 1u : 0u;
  hstart[i] = (heavy && idx == 0u) ? 1u : 0u;
  if (!next_same) klen[rs[i]] = idx + 1u;
  role_out[i] = (uint8_t)role;
}
// info = role | (run length odd) << 2 | index inside the run << 3
__global__ void k_info(const uint32_t *__restrict__ rs, const uint32_t *__restrict__ klen, const uint8_t *__restrict__ role, int64_t n,
                       uint32_t *__restrict__ info) {
  const int64_t i = GID;
  if (i >= n) return;
  const uint32_t s = rs[i];
  info[i] = (uint32_t)role[i] | ((klen[s] & 1u) << 2) | (((uint32_t)i - s) << 3);
}
__global__ void k_lrp(const int32_t *__restrict__ rp, const uint32_t *__restrict__ repscan, int64_t rows, int64_t heavy_thr,
                      uint32_t *__restrict__ lrp) {
  const int64_t r = GID;
  if (r > rows) return;
  uint32_t v = repscan[rp[r]];
  if (r < rows && (int64_t)rp[r + 1] - rp[r] >= heavy_thr) v |= 0x80000000u;
  lrp[r] = v;
}
__global__ void k_bin_of_row(const int32_t *__restrict__ bin_r0, int32_t n_bins, int64_t rows, uint32_t *__restrict__ bin_of) {
  const int64_t r = GID;
  if (r >= rows) return;
  int32_t lo = 0, hi = n_bins;   // bin_r0[lo] <= r < bin_r0[hi] (bin_r0[n_bins] = rows)
  while (hi - lo > 1) {
    const int32_t mid = (lo + hi) >> 1;
    if ((int64_t)bin_r0[mid] <= r) lo = mid; else hi = mid;
  }
  bin_of[r] = (uint32_t)lo;
}
// sort key of the T-order: the tile; heavy entries get CT and end up behind all light ones
__global__ void k_tile_keys(const uint64_t *__restrict__ keyA, const uint8_t *__restrict__ role, int64_t n, int tb, uint32_t CT,
                            uint32_t *__restrict__ kout, uint32_t *__restrict__ vout) {
  const int64_t i = GID;
  if (i >= n) return;
  kout[i] = role[i] == ROLE_HEAVY ? CT : (uint32_t)(keyA[i] & ((1ull << tb) - 1ull));
  vout[i] = (uint32_t)i;
}
__global__ void k_piece_flags(const uint32_t *__restrict__ tileB, const uint32_t *__restrict__ iB, const uint64_t *__restrict__ keyA,
                              const uint32_t *__restrict__ bin_of, const uint32_t *__restrict__ info, int64_t L, int tb,
                              uint32_t *__restrict__ pflag, uint64_t *__restrict__ cnt, uint32_t *__restrict__ binT) {
  const int64_t u = GID;
  if (u >= L) return;
  const uint32_t i = iB[u];
  const uint32_t bin = bin_of[keyA[i] >> tb], tile = tileB[u];
  bool st = u == 0;
  if (!st) st = tileB[u - 1] != tile || bin_of[keyA[iB[u - 1]] >> tb] != bin;
  pflag[u] = st ? 1u : 0u;
  const uint32_t role = info[i] & 3u;
  cnt[u] = role == ROLE_FIRST ? (1ull << 32) : (role == ROLE_SINGLE ? 1ull : 0ull);
  binT[u] = bin;
}
__global__ void k_piece_firsts(const uint32_t *__restrict__ pflag, const uint32_t *__restrict__ pid1, const uint32_t *__restrict__ tileB,
                               const uint32_t *__restrict__ binT, int64_t L, int64_t NP, uint32_t *__restrict__ p_first,
                               uint32_t *__restrict__ p_tile, uint32_t *__restrict__ p_bin) {
  const int64_t u = GID;
  if (u == 0) p_first[NP] = (uint32_t)L;
  if (u >= L || !pflag[u]) return;
  const uint32_t p = pid1[u] - 1u;
  p_first[p] = (uint32_t)u;
  p_tile[p] = tileB[u];
  p_bin[p] = binT[u];
}
__global__ void k_piece_sizes(const uint32_t *__restrict__ p_first, const uint64_t *__restrict__ cscan, int64_t NP,
                              uint32_t *__restrict__ p_np, uint32_t *__restrict__ p_ns, uint64_t *__restrict__ p_g4,
                              uint64_t *__restrict__ p_prods) {
  const int64_t p = GID;
  if (p >= NP) return;
  const uint64_t d = cscan[p_first[p + 1]] - cscan[p_first[p]];
  const uint32_t np = (uint32_t)(d >> 32), ns = (uint32_t)d;
  const PiecePack pk = pack_piece(np, ns);
  p_np[p] = np; p_ns[p] = ns;
  p_g4[p] = 4ull * (uint64_t)pk.groups;
  p_prods[p] = (uint64_t)pk.products;
}
// per tile / per bin: the scan values at its first element and behind its last one (the arrays are zero for tiles /
// bins without elements).  key[] is ascending over the NP elements; s1, s2 have NP + 1 entries.
__global__ void k_bounds(const uint32_t *__restrict__ key, const uint64_t *__restrict__ s1, const uint64_t *__restrict__ s2, int64_t NP,
                         uint64_t *__restrict__ b1, uint64_t *__restrict__ e1, uint64_t *__restrict__ b2, uint64_t *__restrict__ e2) {
  const int64_t p = GID;
  if (p >= NP) return;
  const uint32_t t = key[p];
  if (p == 0 || key[p - 1] != t) { b1[t] = s1[p]; b2[t] = s2[p]; }
  if (p == NP - 1 || key[p + 1] != t) { e1[t] = s1[p + 1]; e2[t] = s2[p + 1]; }
}
__global__ void k_piece_spos(const uint32_t *__restrict__ p_tile, const uint64_t *__restrict__ sS, const uint64_t *__restrict__ tbS,
                             const uint64_t *__restrict__ run_start, int64_t NP, uint32_t *__restrict__ p_spos) {
  const int64_t p = GID;
  if (p >= NP) return;
  const uint32_t t = p_tile[p];
  p_spos[p] = (uint32_t)(run_start[t] + (sS[p] - tbS[t]));
}
__global__ void k_gather64(const uint64_t *__restrict__ src, const uint32_t *__restrict__ idx, int64_t n, uint64_t *__restrict__ dst) {
  const int64_t k = GID;
  if (k < n) dst[k] = src[idx[k]];
}
// B-order (pieces by bin, then tile): where the piece's products start in the bin-major product order
__global__ void k_piece_off(const uint32_t *__restrict__ pB, const uint64_t *__restrict__ offB, int64_t NP, uint32_t *__restrict__ p_off) {
  const int64_t k = GID;
  if (k < NP) p_off[pB[k]] = (uint32_t)offB[k];
}

// value word -> dictionary code (n_words sorted words in w[], their codes in c[]); raw values: n_words == 0
// (bits: 4 / 8 -> one byte per entry in code8[], nibbles packed afterwards; 16 -> two bytes per entry)
struct Coding { const uint32_t *words; const uint16_t *codes; int n_words; int bits; };
__device__ __forceinline__ uint16_t code_of(const uint32_t *w, const uint16_t *c, int n, uint32_t v) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (w[mid] < v) lo = mid + 1; else hi = mid;
  }
  return c[lo];
}
// the light stream: one thread per light entry in T-order
__global__ void k_fill_light(int64_t L, const uint32_t *__restrict__ iB, const uint32_t *__restrict__ pid1, const uint32_t *__restrict__ p_first,
                             const uint64_t *__restrict__ cscan, const uint32_t *__restrict__ p_np, const uint32_t *__restrict__ p_ns,
                             const uint32_t *__restrict__ p_spos, const uint32_t *__restrict__ p_off, const uint32_t *__restrict__ binT,
                             const uint32_t *__restrict__ info, const uint32_t *__restrict__ repscan, const uint32_t *__restrict__ jA,
                             const int32_t *__restrict__ ci, const uint32_t *__restrict__ val, const RowBin *__restrict__ bins, int64_t cols,
                             Coding cd, uint16_t *__restrict__ tcol, uint8_t *__restrict__ code8, uint32_t *__restrict__ tval,
                             uint16_t *__restrict__ pslot) {
  // a dictionary of up to VDICT words is staged in LDS; a larger one (two-byte codes) is searched where it lies (16 KB: cache hits)
  __shared__ uint32_t sw_l[VDICT];
  __shared__ uint16_t sc_l[VDICT];
  const bool dict_in_lds = cd.n_words <= VDICT;
  if (dict_in_lds)
    for (int k = threadIdx.x; k < cd.n_words; k += PBS) { sw_l[k] = cd.words[k]; sc_l[k] = cd.codes[k]; }
  const uint32_t *sw = dict_in_lds ? sw_l : cd.words;
  const uint16_t *sc = dict_in_lds ? sc_l : cd.codes;
  __syncthreads();
  const int64_t u = GID;
  if (u >= L) return;
  const uint32_t i = iB[u], inf = info[i];
  const uint32_t role = inf & 3u, kodd = (inf >> 2) & 1u, idx = inf >> 3;
  const uint32_t p = pid1[u] - 1u;
  const uint64_t c = cscan[u] - cscan[p_first[p]];
  const uint32_t pf = (uint32_t)(c >> 32), sg = (uint32_t)c;   // pair-firsts / singles of the piece in front of this entry
  const int64_t np = p_np[p], ns = p_ns[p], S = p_spos[p];
  const int64_t blocks = (np + 3) / 4;
  const int64_t front = (blocks > 0 && ((S >> 2) & 1)) ? 1 : 0;   // a singles group in front: the pair blocks start on an even group index
  int64_t q, o;
  bool fold_flag = false;
  if (role == ROLE_SINGLE) {
    if (front && sg < 4u) { q = S + sg; o = sg; }
    else {
      const int64_t s2 = (int64_t)sg - (front ? (ns < 4 ? ns : 4) : 0), spare = 4 * blocks - np;
      if (s2 < spare) {                       // a free column of the last pair block (its B entry stays padding)
        const int64_t kk = np - 4 * (blocks - 1) + s2;
        q = S + 4 * front + 8 * (blocks - 1) + kk; o = 4 * front + 4 * (blocks - 1) + kk;
      } else {
        const int64_t s3 = s2 - spare;
        q = S + 4 * front + 8 * blocks + s3; o = 4 * front + 4 * blocks + s3;
      }
    }
  } else {
    const int64_t pi = role == ROLE_FIRST ? pf : (int64_t)pf - 1;
    q = S + 4 * front + 8 * (pi / 4) + (pi % 4) + (role == ROLE_SECOND ? 4 : 0);
    o = 4 * front + 4 * (pi / 4) + (pi % 4);
    fold_flag = role == ROLE_SECOND && (pi % 4) == 0;
  }
  const uint32_t j = jA[i];
  const int32_t col = ci[j];
  uint16_t tc = ((uint32_t)col < (uint32_t)cols) ? (uint16_t)(col % TCOLS) : TCOL_IDENTITY;
  if (fold_flag) tc |= TCOL_FOLD;
  tcol[q] = tc;
  if (cd.n_words) {
    const uint16_t code = code_of(sw, sc, cd.n_words, val[j]);
    if (cd.bits == 16) reinterpret_cast<uint16_t *>(code8)[q] = code;
    else code8[q] = (uint8_t)code;
  }
  else tval[q] = val[j];
  if (role != ROLE_SECOND) {
    // slot of the product in its bin's image: row-major product order (a run's pairs, then its single) -- except that
    // a single sitting in the singles group in FRONT of the pair blocks was handed its slot before the run's pairs
    int64_t slot = (int64_t)repscan[i] - bins[binT[u]].csr0;
    const bool single_in_front = front && sg < 4u;
    if (role == ROLE_FIRST && kodd && single_in_front) slot += 1;
    if (role == ROLE_SINGLE && single_in_front) slot -= idx / 2;
    pslot[(int64_t)p_off[p] + o] = (uint16_t)slot;
  }
}
// piece tables, one thread per piece in B-order (= its index in ptab[])
__global__ void k_piece_tables(const uint32_t *__restrict__ pB, const uint64_t *__restrict__ offB, const uint32_t *__restrict__ p_bin,
                               const uint32_t *__restrict__ p_tile, const uint64_t *__restrict__ sP, const RowBin *__restrict__ bins, int64_t NP,
                               int32_t *__restrict__ ptab, uint16_t *__restrict__ ptile, uint32_t *__restrict__ gblk) {
  const int64_t k = GID;
  if (k >= NP) return;
  const uint32_t p = pB[k];
  ptile[k] = (uint16_t)p_tile[p];
  const RowBin b = bins[p_bin[p]];
  const int64_t g_in_bin = ((int64_t)offB[k] - b.pstart) / 4;
  ptab[k] = (int32_t)((int64_t)(sP[p] / 4) - g_in_bin);
  atomicOr(&gblk[((int64_t)b.gb0 + g_in_bin / 64) * 4 + (g_in_bin % 64) / 32], 1u << (g_in_bin % 32));
}
__global__ void k_gblk_before(const RowBin *__restrict__ bins, int64_t n_bins, uint32_t *__restrict__ gblk) {
  const int64_t bi = GID;
  if (bi >= n_bins) return;
  const RowBin b = bins[bi];
  int64_t nblk = ((int64_t)b.n / 4 + 63) / 64;
  if (nblk < 1) nblk = 1;
  uint32_t before = 0;
  for (int64_t j = 0; j < nblk; j++) {
    uint32_t *rec = &gblk[((int64_t)b.gb0 + j) * 4];
    rec[2] = before;
    before += (uint32_t)(__popc(rec[0]) + __popc(rec[1]));
  }
}
// storing groups (no fold flag) of every 64-group block of the light runs: one wave per block
__global__ void k_block_counts(const uint16_t *__restrict__ tcol, const uint64_t *__restrict__ ob0, const uint64_t *__restrict__ run_start,
                               const uint64_t *__restrict__ run_len, int CT, int64_t NB, uint32_t *__restrict__ bcnt) {
  const int64_t B = GID >> 6;
  const int lane = threadIdx.x & 63;
  if (B >= NB) return;   // (whole waves: PBS is a multiple of 64)
  int lo = 0, hi = CT;   // ob0[lo] <= B < ob0[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((int64_t)ob0[mid] <= B) lo = mid; else hi = mid;
  }
  const int64_t rel = (B - (int64_t)ob0[lo]) * 256 + lane * 4;
  const bool valid = rel < (int64_t)run_len[lo];
  const bool stores = valid && !(tcol[(int64_t)run_start[lo] + rel] & TCOL_FOLD);
  const uint64_t m = __ballot(stores);
  if (lane == 0) bcnt[B] = (uint32_t)__popcll(m);
}
__global__ void k_scale4(const uint32_t *__restrict__ s, int64_t n, uint32_t *__restrict__ out) {
  const int64_t i = GID;
  if (i < n) out[i] = 4u * s[i];
}

// ---- heavy rows
__global__ void k_cells(const uint32_t *__restrict__ hstart, const uint32_t *__restrict__ hscan, const uint64_t *__restrict__ keyA,
                        const uint32_t *__restrict__ klen, int64_t n, int tb, uint32_t *__restrict__ c_tile, uint32_t *__restrict__ c_row,
                        uint32_t *__restrict__ c_cnt) {
  const int64_t i = GID;
  if (i >= n || !hstart[i]) return;
  const uint32_t c = hscan[i];
  c_tile[c] = (uint32_t)(keyA[i] & ((1ull << tb) - 1ull));
  c_row[c] = (uint32_t)(keyA[i] >> tb);
  c_cnt[c] = klen[i];
}
__global__ void k_cell_padded(const uint32_t *__restrict__ cT, const uint32_t *__restrict__ c_cnt, int64_t NC, uint64_t *__restrict__ padT) {
  const int64_t k = GID;
  if (k < NC) padT[k] = (uint64_t)((c_cnt[cT[k]] + HSTRIP - 1) / HSTRIP * HSTRIP);
}
// per cell (by its id): place inside the tile's heavy run, place in the heavy part of the stream, partials
__global__ void k_cell_pos(const uint32_t *__restrict__ cT, const uint32_t *__restrict__ tileT, const uint64_t *__restrict__ hposT,
                           const uint64_t *__restrict__ hbegin, int64_t NC, uint64_t *__restrict__ c_pos, uint64_t *__restrict__ c_glob,
                           uint32_t *__restrict__ c_nparts) {
  const int64_t k = GID;
  if (k >= NC) return;
  const uint32_t c = cT[k];
  const uint64_t pos = hposT[k] - hbegin[tileT[k]], padded = hposT[k + 1] - hposT[k];
  c_pos[c] = pos;
  c_glob[c] = hposT[k];
  const uint64_t k0 = pos / HSTRIP, k1 = (pos + padded) / HSTRIP - 1;
  c_nparts[c] = (uint32_t)(k1 / 64 - k0 / 64 + 1);
}
__global__ void k_row_first_flags(const uint32_t *__restrict__ c_row, int64_t NC, uint32_t *__restrict__ f) {
  const int64_t c = GID;
  if (c < NC) f[c] = (c == 0 || c_row[c] != c_row[c - 1]) ? 1u : 0u;
}
__global__ void k_heavy_rows(const uint32_t *__restrict__ c_row, const uint32_t *__restrict__ rf, const uint32_t *__restrict__ rfscan,
                             const uint32_t *__restrict__ c_part, int64_t NC, LongRow *__restrict__ hv) {
  const int64_t c = GID;
  if (c >= NC) return;
  const uint32_t hi = rfscan[c] - 1u;   // inclusive scan of the row-first flags
  if (rf[c]) { hv[hi].row = (int32_t)c_row[c]; hv[hi].slot0 = (int32_t)c_part[c]; hv[hi].pad = 0; }
  if (c == NC - 1 || c_row[c + 1] != c_row[c]) hv[hi].nslots = (int32_t)c_part[c + 1];   // made relative to slot0 below
}
__global__ void k_heavy_rows_fix(LongRow *__restrict__ hv, int64_t n) {
  const int64_t h = GID;
  if (h < n) hv[h].nslots -= hv[h].slot0;
}
// gdest: one thread per heavy strip (strips are numbered in the order of the heavy part of the stream)
__global__ void k_gdest(const uint64_t *__restrict__ hposT, const uint32_t *__restrict__ cT, const uint64_t *__restrict__ c_pos,
                        const uint32_t *__restrict__ c_part, int64_t NC, int64_t NS, uint32_t *__restrict__ gdest) {
  const int64_t s = GID;
  if (s >= NS) return;
  const uint64_t at = (uint64_t)s * HSTRIP;
  int64_t lo = 0, hi = NC;   // hposT[lo] <= at < hposT[hi]
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (hposT[mid] <= at) lo = mid; else hi = mid;
  }
  const uint32_t c = cT[lo];
  const uint64_t pos = c_pos[c], padded = hposT[lo + 1] - hposT[lo], q = at - hposT[lo];
  const uint64_t krel = (pos + q) / HSTRIP, k0 = pos / HSTRIP;
  const uint32_t part = c_part[c] + (uint32_t)(krel / 64 - k0 / 64);
  const uint64_t part_first = (krel / 64 > k0 / 64) ? (krel / 64) * 64 : k0;   // first strip of this partial
  const bool last = q + HSTRIP >= padded || (krel + 1) % 64 == 0;
  gdest[s] = part | ((uint32_t)(krel - part_first) << GD_DIST_SHIFT) | (last ? GD_LAST : 0u);
}
__global__ void k_fill_heavy(int64_t n, const uint8_t *__restrict__ role, const uint32_t *__restrict__ rs, const uint32_t *__restrict__ hscan,
                             const uint64_t *__restrict__ c_glob, int64_t heavy_base, const uint32_t *__restrict__ jA,
                             const int32_t *__restrict__ ci, const uint32_t *__restrict__ val, int64_t cols, Coding cd,
                             uint16_t *__restrict__ tcol, uint8_t *__restrict__ code8, uint32_t *__restrict__ tval) {
  // a dictionary of up to VDICT words is staged in LDS; a larger one (two-byte codes) is searched where it lies (16 KB: cache hits)
  __shared__ uint32_t sw_l[VDICT];
  __shared__ uint16_t sc_l[VDICT];
  const bool dict_in_lds = cd.n_words <= VDICT;
  if (dict_in_lds)
    for (int k = threadIdx.x; k < cd.n_words; k += PBS) { sw_l[k] = cd.words[k]; sc_l[k] = cd.codes[k]; }
  const uint32_t *sw = dict_in_lds ? sw_l : cd.words;
  const uint16_t *sc = dict_in_lds ? sc_l : cd.codes;
  __syncthreads();
  const int64_t i = GID;
  if (i >= n || role[i] != ROLE_HEAVY) return;
  const uint32_t s = rs[i];
  const int64_t q = heavy_base + (int64_t)c_glob[hscan[s]] + ((int64_t)i - s);
  const uint32_t j = jA[i];
  const int32_t col = ci[j];
  tcol[q] = ((uint32_t)col < (uint32_t)cols) ? (uint16_t)(col % TCOLS) : TCOL_IDENTITY;
  if (cd.n_words) {
    const uint16_t code = code_of(sw, sc, cd.n_words, val[j]);
    if (cd.bits == 16) reinterpret_cast<uint16_t *>(code8)[q] = code;
    else code8[q] = (uint8_t)code;
  }
  else tval[q] = val[j];
}
__global__ void k_pack_nibbles(const uint8_t *__restrict__ code8, int64_t nbytes, uint8_t *__restrict__ tcode) {
  const int64_t b = GID;
  if (b < nbytes) tcode[b] = (uint8_t)(code8[2 * b] | (code8[2 * b + 1] << 4));
}

// ---- distinct value words (<= VDICT, else overflow).  table: DT slots of (1 << 32 | word), 0 = empty; ctl[0] = words
// in the table, ctl[1] = overflow.
constexpr int DT = 2048, DL = 1024;
__device__ __forceinline__ uint32_t dhash(uint32_t w) { return (w * 2654435761u) >> 16; }
__global__ void k_distinct(const uint32_t *__restrict__ val, int64_t n, unsigned long long *__restrict__ table, uint32_t *__restrict__ ctl) {
  __shared__ unsigned long long loc[DL];
  __shared__ int lcount, lover;
  for (int k = threadIdx.x; k < DL; k += PBS) loc[k] = 0ull;
  if (threadIdx.x == 0) { lcount = 0; lover = 0; }
  __syncthreads();
  constexpr int PER = 64;   // values per thread
  const int64_t base = (int64_t)blockIdx.x * PBS * PER;
  if (ctl[1] == 0u) {
    for (int k = 0; k < PER; k++) {
      const int64_t i = base + (int64_t)k * PBS + threadIdx.x;
      if (i >= n || lover) break;
      const unsigned long long e = (1ull << 32) | val[i];
      uint32_t h = dhash(val[i]) & (DL - 1);
      for (int probes = 0; probes < DL; probes++) {
        const unsigned long long old = atomicCAS(&loc[h], 0ull, e);
        if (old == e) break;
        if (old == 0ull) { if (atomicAdd(&lcount, 1) >= VDICT) lover = 1; break; }
        h = (h + 1) & (DL - 1);
      }
    }
  }
  __syncthreads();
  if (lover) { if (threadIdx.x == 0) ctl[1] = 1u; return; }
  for (int k = threadIdx.x; k < DL; k += PBS) {
    const unsigned long long e = loc[k];
    if (e == 0ull) continue;
    uint32_t h = dhash((uint32_t)e) & (DT - 1);
    for (int probes = 0; probes < DT; probes++) {
      const unsigned long long old = atomicCAS(&table[h], 0ull, e);
      if (old == e) break;
      if (old == 0ull) { if (atomicAdd(&ctl[0], 1u) >= (uint32_t)VDICT) ctl[1] = 1u; break; }
      h = (h + 1) & (DT - 1);
    }
  }
}

// The same for up to VDICT16 words, straight into a table of DT16 slots in device memory (launched only when the table
// above overflowed: a matrix with a few hundred or thousand distinct values hits existing entries almost always, and one
// with more overflows within the first blocks; later blocks return at once).
constexpr int DT16 = 16384;
__global__ void k_distinct16(const uint32_t *__restrict__ val, int64_t n, unsigned long long *__restrict__ table, uint32_t *__restrict__ ctl) {
  constexpr int PER = 64;
  const int64_t base = (int64_t)blockIdx.x * PBS * PER;
  for (int k = 0; k < PER; k++) {
    const int64_t i = base + (int64_t)k * PBS + threadIdx.x;
    if (i >= n || __hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    const unsigned long long e = (1ull << 32) | val[i];
    uint32_t h = dhash(val[i]) & (DT16 - 1);
    for (int probes = 0; probes < DT16; probes++) {
      unsigned long long old = __hip_atomic_load(&table[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == 0ull) old = atomicCAS(&table[h], 0ull, e);
      if (old == e) break;
      if (old == 0ull) { if (atomicAdd(&ctl[0], 1u) >= (uint32_t)VDICT16) ctl[1] = 1u; break; }
      h = (h + 1) & (DT16 - 1);
    }
  }
}

// ---- the bit-blocked (or,and) layout
// key = the entry's cell ((row range, column block, sub-range), as the host's cell_of), dead entries (zero value or column
// out of range) = ncell: they sort behind all live ones
__global__ void k_bits_keys(const uint32_t *__restrict__ row_of, const int32_t *__restrict__ ci, const uint32_t *__restrict__ val, int64_t n,
                            int64_t cols, int32_t n_ct, uint32_t ncell, uint32_t *__restrict__ key, uint32_t *__restrict__ idx) {
  const int64_t j = GID;
  if (j >= n) return;
  const int32_t c = ci[j];
  const uint32_t r = row_of[j];
  const bool live = val[j] != 0u && (uint32_t)c < (uint32_t)cols;
  key[j] = live ? (uint32_t)((((int64_t)(r / BITS_BR) * n_ct + c / BITS_BC) * BITS_NSUB) + (r % BITS_BR) / BITS_SUB) : ncell;
  idx[j] = (uint32_t)j;
}
__global__ void k_bits_bounds(const uint32_t *__restrict__ key, int64_t n, uint32_t ncell, uint32_t *__restrict__ cstart, uint32_t *__restrict__ cend) {
  const int64_t i = GID;
  if (i >= n) return;
  const uint32_t k = key[i];
  if (k >= ncell) return;
  if (i == 0 || key[i - 1] != k) cstart[k] = (uint32_t)i;
  if (i == n - 1 || key[i + 1] != k) cend[k] = (uint32_t)i + 1u;
}
// entry word = column inside the block | row inside the sub-range << 19; a cell's last entry also writes the padding
__global__ void k_bits_fill(const uint32_t *__restrict__ key, const uint32_t *__restrict__ jS, const uint32_t *__restrict__ row_of,
                            const int32_t *__restrict__ ci, int64_t n, uint32_t ncell, const uint32_t *__restrict__ cstart,
                            const uint32_t *__restrict__ cend, const uint32_t *__restrict__ start, uint32_t *__restrict__ ent) {
  const int64_t i = GID;
  if (i >= n) return;
  const uint32_t k = key[i];
  if (k >= ncell) return;
  const uint32_t j = jS[i];
  const uint32_t w = (uint32_t)(ci[j] % BITS_BC) | ((row_of[j] % BITS_SUB) << 19);
  const uint32_t pos = start[k] + ((uint32_t)i - cstart[k]);
  ent[pos] = w;
  if ((uint32_t)i + 1u == cend[k]) {
    const uint32_t cnt = cend[k] - cstart[k], end = start[k] + ((cnt + 7u) & ~7u);
    for (uint32_t q = pos + 1; q < end; q++) ent[q] = w;
  }
}

} // namespace

// One scratch buffer for all rocPRIM calls, grown on demand.
struct CubTemp {
  void *p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    const hipError_t r = hipMalloc(&p, n + 256);
    if (r == hipSuccess) cap = n;
    return r;
  }
  ~CubTemp() { if (p) (void)hipFree(p); }
};

int build_tiled_plan_gpu(hipStream_t stream, int64_t rows, int64_t cols, int64_t nnz, const int32_t *h_rp, const int32_t *d_rp,
                          const int32_t *d_ci, const uint32_t *d_val, const sh_plan_options &opt, int n_cus, TiledHost &H,
                          TiledDevArrays &D, std::string &why) {
  const int CT = (int)std::max<int64_t>(1, (cols + TCOLS - 1) / TCOLS);
  if (CT > 65535 || nnz <= 0 || rows <= 0) { why = "not applicable"; return 0; }
  const bool fold = opt.fold != 0 && TCOL_FOLD != 0;
  const int64_t per_tile = std::max(1, opt.heavy_per_tile);
  const int64_t heavy_thr = std::min<int64_t>(TBIN / 4, std::max<int64_t>(512, per_tile * CT));
  auto is_heavy = [&](int64_t r) { return (int64_t)h_rp[r + 1] - h_rp[r] >= heavy_thr; };
  const int64_t n = nnz;
  const int tb = bits_for((uint64_t)CT - 1), rb = bits_for((uint64_t)rows - 1);

  DevPool pool;
  pool.stream = stream;
  CubTemp tmp;
  hipError_t herr = hipSuccess;
  const char *hwhat = "";
#define GT(call)                                                                                     \
  do {                                                                                               \
    herr = (call);                                                                                   \
    if (herr != hipSuccess) { hwhat = #call; goto hip_failed; }                                      \
  } while (0)
#define POOL_OK()                                                                                    \
  do {                                                                                               \
    if (pool.err != hipSuccess) { herr = pool.err; hwhat = "hipMalloc (temporaries)"; goto hip_failed; } \
  } while (0)
#define LAUNCH(kernel, count, ...)                                                                   \
  do {                                                                                               \
    if ((count) > 0) hipLaunchKernelGGL(kernel, grid_for(count), dim3(PBS), 0, stream, __VA_ARGS__);  \
  } while (0)
  // rocPRIM calls: size query, then the call
#define CUB(...)                                                                                     \
  do {                                                                                               \
    size_t _b = 0;                                                                                   \
    void *_t = nullptr;                                                                              \
    { auto _call = [&](void *d_temp_storage, size_t &temp_storage_bytes) { return __VA_ARGS__; };    \
      GT(_call(_t, _b));                                                                             \
      GT(tmp.reserve(_b));                                                                           \
      _t = tmp.p;                                                                                    \
      GT(_call(_t, _b)); }                                                                           \
  } while (0)

#ifdef SH_PLAN_EMULATE
#define PHASE(name) do { if (getenv("SH_BUILD_TIMES")) { (void)hipStreamSynchronize(stream); lap(name); } } while (0)
#else
#define PHASE(name) do { } while (0)
#endif
  lap(nullptr);
  // All declarations up front: the error path is a goto.
  uint32_t *row_of = nullptr, *head = nullptr, *valA_in = nullptr, *jA = nullptr, *start_idx = nullptr, *rs = nullptr, *rep = nullptr,
           *hstart = nullptr, *klen = nullptr, *info = nullptr, *repscan = nullptr, *hscan = nullptr, *bin_of = nullptr,
           *keyB_in = nullptr, *valB_in = nullptr, *tileB = nullptr, *iB = nullptr, *pflag = nullptr, *pid1 = nullptr, *binT = nullptr,
           *p_first = nullptr, *p_tile = nullptr, *p_bin = nullptr, *p_np = nullptr, *p_ns = nullptr, *p_spos = nullptr, *p_off = nullptr,
           *piota = nullptr, *pB = nullptr, *p_bin_sorted = nullptr, *bcnt = nullptr, *bscan = nullptr;
  uint64_t *keyA_in = nullptr, *keyA = nullptr, *cnt = nullptr, *cscan = nullptr, *p_g4 = nullptr, *p_prods = nullptr, *sS = nullptr, *sP = nullptr,
           *tbS = nullptr, *teS = nullptr, *tbP = nullptr, *teP = nullptr, *d_run_start = nullptr, *d_run_len = nullptr, *d_ob0 = nullptr,
           *prodsB = nullptr, *offB = nullptr, *bk0 = nullptr, *bk1 = nullptr, *bo0 = nullptr, *bo1 = nullptr, *kidx = nullptr;
  uint8_t *role = nullptr, *code8 = nullptr;
  int32_t *d_bin_r0 = nullptr;
  RowBin *d_bins = nullptr;
  std::vector<int64_t> run_len((size_t)CT, 0), run_plen((size_t)CT, 0), run_start((size_t)CT, 0), hrel((size_t)CT, 0), heavy_start((size_t)CT, 0),
      ob0((size_t)CT + 1, 0);
  std::vector<uint64_t> h64a, h64b, h64c, h64d;
  int64_t L = 0, NP = 0, NC = 0, n_heavy = 0, n_pieces_total = 0, n_blocks_total = 0, p_off_total = 0, heavy_total = 0, heavy_base = 0;
  // heavy-side arrays
  uint32_t *c_tile = nullptr, *c_row = nullptr, *c_cnt = nullptr, *ciota = nullptr, *cT = nullptr, *tileT = nullptr, *c_nparts = nullptr,
           *c_part = nullptr, *rf = nullptr, *rfscan = nullptr;
  uint64_t *padT = nullptr, *hposT = nullptr, *hb = nullptr, *he = nullptr, *hb2 = nullptr, *he2 = nullptr, *c_pos = nullptr, *c_glob = nullptr;
  LongRow *d_hv = nullptr;
  unsigned long long *dtable = nullptr;
  uint32_t *dctl = nullptr, *d_words = nullptr;
  uint16_t *d_codes = nullptr;
  Coding cd{nullptr, nullptr, 0, 0};
  ValSet dict;
  bool coded = false;
  uint32_t u32tmp = 0;

  // ---- A-order: entries by (row, tile), stable
  row_of = pool.get<uint32_t>((size_t)n);
  head = pool.get<uint32_t>((size_t)n, true);
  keyA_in = pool.get<uint64_t>((size_t)n); keyA = pool.get<uint64_t>((size_t)n);
  valA_in = pool.get<uint32_t>((size_t)n); jA = pool.get<uint32_t>((size_t)n);
  POOL_OK();
  LAUNCH(k_row_heads, rows, d_rp, rows, head);
  CUB(incl_max(d_temp_storage, temp_storage_bytes, head, row_of, n, stream));
  LAUNCH(k_keys, n, row_of, d_ci, n, cols, tb, keyA_in, valA_in);
  CUB(sort_pairs(d_temp_storage, temp_storage_bytes, keyA_in, keyA, valA_in, jA, n, 0, tb + rb, stream));
  PHASE("  A-order sort");
  // runs, roles
  start_idx = head;   // (head is free again)
  rs = row_of;        // (so is row_of: the keys carry the rows)
  LAUNCH(k_run_starts, n, keyA, n, start_idx);
  CUB(incl_max(d_temp_storage, temp_storage_bytes, start_idx, rs, n, stream));
  rep = pool.get<uint32_t>((size_t)n + 1, true);
  hstart = pool.get<uint32_t>((size_t)n + 1, true);
  klen = pool.get<uint32_t>((size_t)n);
  role = pool.get<uint8_t>((size_t)n);
  info = valA_in;     // (free after the sort)
  repscan = pool.get<uint32_t>((size_t)n + 1);
  hscan = pool.get<uint32_t>((size_t)n + 1);
  POOL_OK();
  LAUNCH(k_roles, n, keyA, rs, d_rp, n, tb, heavy_thr, fold ? 1 : 0, rep, hstart, klen, role);
  LAUNCH(k_info, n, rs, klen, role, n, info);
  CUB(excl_sum(d_temp_storage, temp_storage_bytes, rep, repscan, n + 1, stream));
  CUB(excl_sum(d_temp_storage, temp_storage_bytes, hstart, hscan, n + 1, stream));
  PHASE("  runs, roles, scans");
  // lrp (a final array) and the row bins (host: a greedy cut is sequential)
  D.n_lrp = (size_t)rows + 1;
  GT(hipMalloc((void **)&D.lrp, D.n_lrp * 4 + 16));   // (+16: phase 2 copies the offsets of a bin in 16-byte groups, the last one may reach past the array)
  LAUNCH(k_lrp, rows + 1, d_rp, repscan, rows, heavy_thr, D.lrp);
  H.lrp.resize((size_t)rows + 1);
  GT(hipMemcpyAsync(H.lrp.data(), D.lrp, D.n_lrp * 4, hipMemcpyDeviceToHost, stream));
  GT(hipMemcpyAsync(&u32tmp, hscan + n, 4, hipMemcpyDeviceToHost, stream));
  GT(hipStreamSynchronize(stream));
  NC = u32tmp;
  {
    auto light_off = [&](int64_t r) { return (int64_t)(H.lrp[(size_t)r] & 0x7FFFFFFFu); };
    const int64_t bin_target = std::max<int64_t>(TBIN / 4, (int64_t)TBIN - 3ll * CT);
    for (int64_t r = 0; r < rows;) {
      int64_t r1 = r + 1;
      while (r1 < rows && r1 - r < TBIN_ROWS && light_off(r1 + 1) - light_off(r) <= bin_target) r1++;
      RowBin b{};
      b.r0 = (int32_t)r; b.nr = (int32_t)(r1 - r); b.csr0 = (int32_t)light_off(r);
      H.bins.push_back(b);
      r = r1;
    }
    int64_t le = 0;
    for (int64_t r = 0; r < rows; r++) {
      if (is_heavy(r)) n_heavy++;
      else le += h_rp[r + 1] - h_rp[r];
    }
    H.light_entries = le;
    L = le;
  }
  {
    const int64_t n_bins = (int64_t)H.bins.size();
    std::vector<int32_t> r0((size_t)n_bins + 1);
    for (int64_t b = 0; b < n_bins; b++) r0[(size_t)b] = H.bins[(size_t)b].r0;
    r0[(size_t)n_bins] = (int32_t)rows;
    d_bin_r0 = pool.get<int32_t>((size_t)n_bins + 1);
    bin_of = pool.get<uint32_t>((size_t)rows);
    POOL_OK();
    GT(hipMemcpyAsync(d_bin_r0, r0.data(), ((size_t)n_bins + 1) * 4, hipMemcpyHostToDevice, stream));
    LAUNCH(k_bin_of_row, rows, d_bin_r0, (int32_t)n_bins, rows, bin_of);
    GT(hipStreamSynchronize(stream));   // r0 dies here
  }

  PHASE("  lrp, bins (host), bin_of");
  // ---- T-order: light entries by tile, stable
  keyB_in = pool.get<uint32_t>((size_t)n); valB_in = pool.get<uint32_t>((size_t)n);
  tileB = pool.get<uint32_t>((size_t)n); iB = pool.get<uint32_t>((size_t)n);
  POOL_OK();
  LAUNCH(k_tile_keys, n, keyA, role, n, tb, (uint32_t)CT, keyB_in, valB_in);
  CUB(sort_pairs(d_temp_storage, temp_storage_bytes, keyB_in, tileB, valB_in, iB, n, 0, bits_for((uint64_t)CT), stream));
  if (L > 0) {
    pflag = keyB_in;   // (free after the sort)
    pid1 = valB_in;
    cnt = pool.get<uint64_t>((size_t)L + 1, true);
    cscan = pool.get<uint64_t>((size_t)L + 1);
    binT = pool.get<uint32_t>((size_t)L);
    POOL_OK();
    LAUNCH(k_piece_flags, L, tileB, iB, keyA, bin_of, info, L, tb, pflag, cnt, binT);
    CUB(incl_sum(d_temp_storage, temp_storage_bytes, pflag, pid1, L, stream));
    CUB(excl_sum(d_temp_storage, temp_storage_bytes, cnt, cscan, L + 1, stream));
    GT(hipMemcpyAsync(&u32tmp, pid1 + (L - 1), 4, hipMemcpyDeviceToHost, stream));
    GT(hipStreamSynchronize(stream));
    NP = u32tmp;
  }
  PHASE("  T-order sort, piece flags");
  // ---- the piece table in (tile, bin) order
  p_first = pool.get<uint32_t>((size_t)NP + 1); p_tile = pool.get<uint32_t>((size_t)NP + 1); p_bin = pool.get<uint32_t>((size_t)NP + 1);
  p_np = pool.get<uint32_t>((size_t)NP + 1); p_ns = pool.get<uint32_t>((size_t)NP + 1);
  p_spos = pool.get<uint32_t>((size_t)NP + 1); p_off = pool.get<uint32_t>((size_t)NP + 1);
  p_g4 = pool.get<uint64_t>((size_t)NP + 1, true); p_prods = pool.get<uint64_t>((size_t)NP + 1, true);
  sS = pool.get<uint64_t>((size_t)NP + 1); sP = pool.get<uint64_t>((size_t)NP + 1);
  tbS = pool.get<uint64_t>((size_t)CT + 1, true); teS = pool.get<uint64_t>((size_t)CT + 1, true);
  tbP = pool.get<uint64_t>((size_t)CT + 1, true); teP = pool.get<uint64_t>((size_t)CT + 1, true);
  d_run_start = pool.get<uint64_t>((size_t)CT + 1); d_run_len = pool.get<uint64_t>((size_t)CT + 1); d_ob0 = pool.get<uint64_t>((size_t)CT + 2);
  POOL_OK();
  if (NP > 0) {
    LAUNCH(k_piece_firsts, L, pflag, pid1, tileB, binT, L, NP, p_first, p_tile, p_bin);
    LAUNCH(k_piece_sizes, NP, p_first, cscan, NP, p_np, p_ns, p_g4, p_prods);
  }
  CUB(excl_sum(d_temp_storage, temp_storage_bytes, p_g4, sS, NP + 1, stream));
  CUB(excl_sum(d_temp_storage, temp_storage_bytes, p_prods, sP, NP + 1, stream));
  LAUNCH(k_bounds, NP, p_tile, sS, sP, NP, tbS, teS, tbP, teP);
  h64a.resize((size_t)CT); h64b.resize((size_t)CT); h64c.resize((size_t)CT); h64d.resize((size_t)CT);
  GT(hipMemcpyAsync(h64a.data(), tbS, (size_t)CT * 8, hipMemcpyDeviceToHost, stream));
  GT(hipMemcpyAsync(h64b.data(), teS, (size_t)CT * 8, hipMemcpyDeviceToHost, stream));
  GT(hipMemcpyAsync(h64c.data(), tbP, (size_t)CT * 8, hipMemcpyDeviceToHost, stream));
  GT(hipMemcpyAsync(h64d.data(), teP, (size_t)CT * 8, hipMemcpyDeviceToHost, stream));
  GT(hipStreamSynchronize(stream));
  {
    int64_t pos = 0;
    for (int t = 0; t < CT; t++) {
      run_len[(size_t)t] = (int64_t)(h64b[(size_t)t] - h64a[(size_t)t]);
      run_plen[(size_t)t] = (int64_t)(h64d[(size_t)t] - h64c[(size_t)t]);
      run_start[(size_t)t] = pos;
      pos += (run_len[(size_t)t] + 255) & ~int64_t(255);   // every tile's light run starts on a multiple of 64 groups
      ob0[(size_t)t + 1] = ob0[(size_t)t] + (((run_len[(size_t)t] + 255) & ~int64_t(255)) / 256);
    }
    H.light_len = pos;
    if (pos > INT32_MAX - 8) { why = "stream exceeds int32 indexing"; return 0; }
    std::vector<uint64_t> a((size_t)CT), b((size_t)CT), c((size_t)CT + 1);
    for (int t = 0; t < CT; t++) { a[(size_t)t] = (uint64_t)run_start[(size_t)t]; b[(size_t)t] = (uint64_t)run_len[(size_t)t]; }
    for (int t = 0; t <= CT; t++) c[(size_t)t] = (uint64_t)ob0[(size_t)t];
    GT(hipMemcpyAsync(d_run_start, a.data(), (size_t)CT * 8, hipMemcpyHostToDevice, stream));
    GT(hipMemcpyAsync(d_run_len, b.data(), (size_t)CT * 8, hipMemcpyHostToDevice, stream));
    GT(hipMemcpyAsync(d_ob0, c.data(), ((size_t)CT + 1) * 8, hipMemcpyHostToDevice, stream));
    GT(hipStreamSynchronize(stream));
  }
  LAUNCH(k_piece_spos, NP, p_tile, sS, tbS, d_run_start, NP, p_spos);
  // ---- B-order: pieces by bin (stable: tiles ascending inside a bin)
  {
    const int64_t n_bins = (int64_t)H.bins.size();
    piota = pool.get<uint32_t>((size_t)NP + 1); pB = pool.get<uint32_t>((size_t)NP + 1); p_bin_sorted = pool.get<uint32_t>((size_t)NP + 1);
    prodsB = pool.get<uint64_t>((size_t)NP + 1, true); offB = pool.get<uint64_t>((size_t)NP + 1);
    kidx = pool.get<uint64_t>((size_t)NP + 1);
    bk0 = pool.get<uint64_t>((size_t)n_bins + 1, true); bk1 = pool.get<uint64_t>((size_t)n_bins + 1, true);
    bo0 = pool.get<uint64_t>((size_t)n_bins + 1, true); bo1 = pool.get<uint64_t>((size_t)n_bins + 1, true);
    d_bins = pool.get<RowBin>((size_t)n_bins + 1);
    POOL_OK();
    if (NP > 0) {
      LAUNCH(k_iota, NP, piota, NP);
      CUB(sort_pairs(d_temp_storage, temp_storage_bytes, p_bin, p_bin_sorted, piota, pB, NP, 0,
                                             bits_for((uint64_t)std::max<int64_t>(n_bins, 1) - 1), stream));
      LAUNCH(k_gather64, NP, p_prods, pB, NP, prodsB);
    }
    CUB(excl_sum(d_temp_storage, temp_storage_bytes, prodsB, offB, NP + 1, stream));
    if (NP > 0) {
      // kidx[k] = k as a 64-bit "scan", so that k_bounds yields the bins' first piece / one past their last piece
      // (s1 = kidx, s2 = offB)
      LAUNCH(k_iota64, NP + 1, kidx, NP + 1);
      LAUNCH(k_bounds, NP, p_bin_sorted, kidx, offB, NP, bk0, bk1, bo0, bo1);
      LAUNCH(k_piece_off, NP, pB, offB, NP, p_off);
    }
    std::vector<uint64_t> k0((size_t)n_bins), k1((size_t)n_bins), o0((size_t)n_bins), o1((size_t)n_bins);
    GT(hipMemcpyAsync(k0.data(), bk0, (size_t)n_bins * 8, hipMemcpyDeviceToHost, stream));
    GT(hipMemcpyAsync(k1.data(), bk1, (size_t)n_bins * 8, hipMemcpyDeviceToHost, stream));
    GT(hipMemcpyAsync(o0.data(), bo0, (size_t)n_bins * 8, hipMemcpyDeviceToHost, stream));
    GT(hipMemcpyAsync(o1.data(), bo1, (size_t)n_bins * 8, hipMemcpyDeviceToHost, stream));
    GT(hipStreamSynchronize(stream));
    for (int64_t bi = 0; bi < n_bins; bi++) {
      RowBin &b = H.bins[(size_t)bi];
      const int64_t bn = (int64_t)(o1[(size_t)bi] - o0[(size_t)bi]);
      if (bn > TBIN) { why = "a bin exceeds TBIN products"; return 0; }
      if (p_off_total + bn > INT32_MAX) { why = "P exceeds int32 indexing"; return 0; }
      b.n = (int32_t)bn;
      b.pstart = (int32_t)p_off_total;
      p_off_total += bn;
      b.pt0 = (int32_t)n_pieces_total;
      b.gb0 = (int32_t)n_blocks_total;
      n_pieces_total += (int64_t)(k1[(size_t)bi] - k0[(size_t)bi]);
      n_blocks_total += std::max<int64_t>(1, (bn / 4 + 63) / 64);
    }
    H.p_len = p_off_total;
    H.tile_fill = (n_bins > 0) ? (double)n_pieces_total / ((double)n_bins * CT) : 1.0;
    GT(hipMemcpyAsync(d_bins, H.bins.data(), (size_t)n_bins * sizeof(RowBin), hipMemcpyHostToDevice, stream));
  }

  PHASE("  piece tables, B-order");
  // ---- heavy rows: cells = (row, tile) runs of heavy rows
  c_tile = pool.get<uint32_t>((size_t)NC + 1); c_row = pool.get<uint32_t>((size_t)NC + 1); c_cnt = pool.get<uint32_t>((size_t)NC + 1);
  ciota = pool.get<uint32_t>((size_t)NC + 1); cT = pool.get<uint32_t>((size_t)NC + 1); tileT = pool.get<uint32_t>((size_t)NC + 1);
  c_nparts = pool.get<uint32_t>((size_t)NC + 1, true); c_part = pool.get<uint32_t>((size_t)NC + 1);
  rf = pool.get<uint32_t>((size_t)NC + 1); rfscan = pool.get<uint32_t>((size_t)NC + 1);
  padT = pool.get<uint64_t>((size_t)NC + 1, true); hposT = pool.get<uint64_t>((size_t)NC + 1);
  hb = pool.get<uint64_t>((size_t)CT + 1, true); he = pool.get<uint64_t>((size_t)CT + 1, true);
  hb2 = pool.get<uint64_t>((size_t)CT + 1, true); he2 = pool.get<uint64_t>((size_t)CT + 1, true);
  c_pos = pool.get<uint64_t>((size_t)NC + 1); c_glob = pool.get<uint64_t>((size_t)NC + 1);
  d_hv = pool.get<LongRow>((size_t)n_heavy + 1);
  POOL_OK();
  if (NC > 0) {
    LAUNCH(k_cells, n, hstart, hscan, keyA, klen, n, tb, c_tile, c_row, c_cnt);
    LAUNCH(k_iota, NC, ciota, NC);
    CUB(sort_pairs(d_temp_storage, temp_storage_bytes, c_tile, tileT, ciota, cT, NC, 0, tb, stream));
    LAUNCH(k_cell_padded, NC, cT, c_cnt, NC, padT);
  }
  CUB(excl_sum(d_temp_storage, temp_storage_bytes, padT, hposT, NC + 1, stream));
  if (NC > 0) {
    LAUNCH(k_bounds, NC, tileT, hposT, hposT, NC, hb, he, hb2, he2);
    LAUNCH(k_cell_pos, NC, cT, tileT, hposT, hb, NC, c_pos, c_glob, c_nparts);
  }
  // (cells are numbered in A-order = (row, tile): the scan of their partial counts is slot0 + part0 of the host builder)
  CUB(excl_sum(d_temp_storage, temp_storage_bytes, c_nparts, c_part, NC + 1, stream));
  if (NC > 0) {
    LAUNCH(k_row_first_flags, NC, c_row, NC, rf);
    CUB(incl_sum(d_temp_storage, temp_storage_bytes, rf, rfscan, NC, stream));
    LAUNCH(k_heavy_rows, NC, c_row, rf, rfscan, c_part, NC, d_hv);
    LAUNCH(k_heavy_rows_fix, n_heavy, d_hv, n_heavy);
  }
  {
    GT(hipMemcpyAsync(h64a.data(), hb, (size_t)CT * 8, hipMemcpyDeviceToHost, stream));
    GT(hipMemcpyAsync(h64b.data(), he, (size_t)CT * 8, hipMemcpyDeviceToHost, stream));
    uint64_t tot = 0;
    GT(hipMemcpyAsync(&tot, hposT + NC, 8, hipMemcpyDeviceToHost, stream));
    GT(hipMemcpyAsync(&u32tmp, c_part + NC, 4, hipMemcpyDeviceToHost, stream));
    H.heavy.resize((size_t)n_heavy);
    if (n_heavy > 0) GT(hipMemcpyAsync(H.heavy.data(), d_hv, (size_t)n_heavy * sizeof(LongRow), hipMemcpyDeviceToHost, stream));
    GT(hipStreamSynchronize(stream));
    heavy_total = (int64_t)tot;
    if ((int64_t)u32tmp > (int64_t)GD_SLOT_MASK) { why = "too many heavy partials"; return 0; }
    H.n_partials = (int32_t)u32tmp;
    for (int t = 0; t < CT; t++) hrel[(size_t)t] = (int64_t)(h64b[(size_t)t] - h64a[(size_t)t]);
    int64_t total = (H.light_len + HSTRIP - 1) / HSTRIP * HSTRIP;
    heavy_base = total;
    H.heavy_base = heavy_base;
    for (int t = 0; t < CT; t++) { heavy_start[(size_t)t] = total; total += hrel[(size_t)t]; }
    if (total != heavy_base + heavy_total) { why = "internal: heavy runs do not add up"; return -1; }
    if (total > INT32_MAX - 8) { why = "stream exceeds int32 indexing"; return 0; }
    H.stream_len = total;
    if (H.stream_len > nnz + nnz / 4 + 4096 + 256ll * CT) { why = "padding would cost more than 25 %"; return 0; }
  }

  PHASE("  heavy cells");
  // ---- value dictionary
  coded = opt.value_coding >= 0;
  if (coded) {
    dtable = pool.get<unsigned long long>((size_t)DT, true);
    dctl = pool.get<uint32_t>(4, true);
    POOL_OK();
    hipLaunchKernelGGL(k_distinct, dim3((unsigned)((n + (int64_t)PBS * 64 - 1) / ((int64_t)PBS * 64))), dim3(PBS), 0, stream, d_val, n, dtable, dctl);
    std::vector<unsigned long long> tab((size_t)DT);
    uint32_t ctl[4] = {0, 0, 0, 0};
    GT(hipMemcpyAsync(tab.data(), dtable, (size_t)DT * 8, hipMemcpyDeviceToHost, stream));
    GT(hipMemcpyAsync(ctl, dctl, 16, hipMemcpyDeviceToHost, stream));
    GT(hipStreamSynchronize(stream));
    std::vector<uint32_t> words;
    bool overflow = ctl[1] != 0u || ctl[0] > (uint32_t)VDICT;
    if (!overflow)
      for (unsigned long long e : tab) if (e) words.push_back((uint32_t)e);
    if (overflow && opt.value_coding == 0) {   // more than VDICT words: up to VDICT16 of them still make two-byte codes
      unsigned long long *dtable16 = pool.get<unsigned long long>((size_t)DT16, true);
      uint32_t *dctl16 = pool.get<uint32_t>(4, true);
      POOL_OK();
      hipLaunchKernelGGL(k_distinct16, dim3((unsigned)((n + (int64_t)PBS * 64 - 1) / ((int64_t)PBS * 64))), dim3(PBS), 0, stream, d_val, n, dtable16, dctl16);
      std::vector<unsigned long long> tab16((size_t)DT16);
      GT(hipMemcpyAsync(tab16.data(), dtable16, (size_t)DT16 * 8, hipMemcpyDeviceToHost, stream));
      GT(hipMemcpyAsync(ctl, dctl16, 16, hipMemcpyDeviceToHost, stream));
      GT(hipStreamSynchronize(stream));
      overflow = ctl[1] != 0u || ctl[0] > (uint32_t)VDICT16;
      if (!overflow)
        for (unsigned long long e : tab16) if (e) words.push_back((uint32_t)e);
    }
    decide_value_coding(words, overflow, opt, H.code_bits, dict);
    coded = !dict.overflow;
  }
  if (coded) {
    H.vdict = dict.list;
    H.vdict_used = (int)H.vdict.size();
    // sorted words and their codes for the device lookup
    std::vector<std::pair<uint32_t, uint16_t>> wc;
    for (size_t k = 0; k < dict.list.size(); k++) wc.emplace_back(dict.list[k], (uint16_t)k);
    std::sort(wc.begin(), wc.end());
    std::vector<uint32_t> ws; std::vector<uint16_t> cs;
    for (auto &pr : wc) { ws.push_back(pr.first); cs.push_back(pr.second); }
    d_words = pool.get<uint32_t>(VDICT16); d_codes = pool.get<uint16_t>(VDICT16);
    POOL_OK();
    GT(hipMemcpyAsync(d_words, ws.data(), ws.size() * 4, hipMemcpyHostToDevice, stream));
    GT(hipMemcpyAsync(d_codes, cs.data(), cs.size() * 2, hipMemcpyHostToDevice, stream));
    GT(hipStreamSynchronize(stream));
    cd = Coding{d_words, d_codes, (int)ws.size(), H.code_bits};
    H.vdict.resize(dict_words(H.code_bits), 0u);
  }

  PHASE("  dictionary");
  // ---- the final arrays
  D.n_tcol = (size_t)H.stream_len;
  D.n_gdest = (size_t)(H.stream_len - heavy_base) / HSTRIP + 1;
  D.n_pslot = (size_t)H.p_len;
  D.n_gblk = (size_t)(n_blocks_total + 1) * 4;
  D.n_ptab = (size_t)n_pieces_total + 1;
  D.n_obase = (size_t)ob0[(size_t)CT] + 1;
  GT(hipMalloc((void **)&D.tcol, D.n_tcol * 2 + SLACK_WIDE));
  GT(hipMalloc((void **)&D.gdest, D.n_gdest * 4 + SLACK_WIDE));
  GT(hipMalloc((void **)&D.pslot, D.n_pslot * 2 + SLACK_WIDE));
  GT(hipMalloc((void **)&D.gblk, D.n_gblk * 4 + SLACK_WIDE));
  GT(hipMalloc((void **)&D.ptab, D.n_ptab * 4 + SLACK_WIDE));
  GT(hipMalloc((void **)&D.ptile, D.n_ptab * 2 + SLACK_WIDE));
  GT(hipMalloc((void **)&D.obase, D.n_obase * 4 + SLACK_WIDE));
  if (coded) {
    D.n_tcode = tcode_bytes(H.code_bits, H.stream_len);
    GT(hipMalloc((void **)&D.tcode, D.n_tcode + SLACK_TCODE));
    if (H.code_bits == 4) { code8 = pool.get<uint8_t>((size_t)H.stream_len, true); POOL_OK(); }
    else { code8 = D.tcode; GT(hipMemsetAsync(D.tcode, 0, D.n_tcode, stream)); }
  } else {
    D.n_tval = (size_t)H.stream_len;
    GT(hipMalloc((void **)&D.tval, D.n_tval * 4 + SLACK_WIDE));
    GT(hipMemsetAsync(D.tval, 0, D.n_tval * 4, stream));
  }
  LAUNCH(k_fill16, (int64_t)D.n_tcol, D.tcol, (int64_t)D.n_tcol, TCOL_IDENTITY);
  GT(hipMemsetAsync(D.gdest, 0, D.n_gdest * 4, stream));
  GT(hipMemsetAsync(D.pslot, 0xFF, D.n_pslot * 2, stream));
  GT(hipMemsetAsync(D.gblk, 0, D.n_gblk * 4, stream));
  GT(hipMemsetAsync(D.ptab, 0, D.n_ptab * 4, stream));
  GT(hipMemsetAsync(D.ptile, 0, D.n_ptab * 2, stream));
  GT(hipMemsetAsync(D.obase, 0, D.n_obase * 4, stream));
  PHASE("  final arrays: malloc + clear");
  // light stream, slots, piece tables
  if (L > 0 && NP > 0) {
    LAUNCH(k_fill_light, L, L, iB, pid1, p_first, cscan, p_np, p_ns, p_spos, p_off, binT, info, repscan, jA, d_ci, d_val, d_bins, cols, cd,
           D.tcol, code8, D.tval, D.pslot);
    LAUNCH(k_piece_tables, NP, pB, offB, p_bin, p_tile, sP, d_bins, NP, D.ptab, D.ptile, D.gblk);
  }
  LAUNCH(k_gblk_before, (int64_t)H.bins.size(), d_bins, (int64_t)H.bins.size(), D.gblk);
  // heavy stream and gdest
  if (NC > 0) {
    LAUNCH(k_gdest, heavy_total / HSTRIP, hposT, cT, c_pos, c_part, NC, heavy_total / HSTRIP, D.gdest);
    LAUNCH(k_fill_heavy, n, n, role, rs, hscan, c_glob, heavy_base, jA, d_ci, d_val, cols, cd, D.tcol, code8, D.tval);
  }
  if (coded && H.code_bits == 4) LAUNCH(k_pack_nibbles, (int64_t)D.n_tcode, code8, (int64_t)D.n_tcode, D.tcode);
  PHASE("  fills");
  // obase: P position of the first product of every 64 stream groups
  {
    const int64_t NB = ob0[(size_t)CT];
    bcnt = pool.get<uint32_t>((size_t)NB + 1, true); bscan = pool.get<uint32_t>((size_t)NB + 1);
    POOL_OK();
    if (NB > 0) {
      LAUNCH(k_block_counts, NB * 64, D.tcol, d_ob0, d_run_start, d_run_len, CT, NB, bcnt);
      CUB(excl_sum(d_temp_storage, temp_storage_bytes, bcnt, bscan, NB + 1, stream));
      LAUNCH(k_scale4, NB + 1, bscan, NB + 1, D.obase);   // (the entry behind the last block = all products: obase[b + 1] - obase[b] = products of block b)
      // every tile's products must end where the next tile's begin (the host builder's consistency check)
      GT(hipMemcpyAsync(&u32tmp, bscan + NB, 4, hipMemcpyDeviceToHost, stream));
      GT(hipStreamSynchronize(stream));
      if (4ll * (int64_t)u32tmp != H.p_len) { why = "internal: fold flags and product counts disagree"; return -1; }
    }
  }
  GT(hipStreamSynchronize(stream));
  GT(hipGetLastError());
  PHASE("  obase");
  cut_work_items(CT, run_start, run_len, heavy_start, hrel, ob0, opt, n_cus, H);
  return 1;

hip_failed:
  why = std::string(hwhat) + ": " + hipGetErrorString(herr);
  (void)hipGetLastError();
  return -1;
#undef GT
#undef PHASE
#undef POOL_OK
#undef LAUNCH
#undef CUB
}

int build_bits_plan_gpu(hipStream_t stream, int64_t rows, int64_t cols, int64_t nnz, const int32_t *d_rp, const int32_t *d_ci,
                        const uint32_t *d_val, BitsHost &H, uint32_t **d_ent, std::string &why) {
  H.n_rr = (int32_t)std::max<int64_t>(1, (rows + BITS_BR - 1) / BITS_BR);
  H.n_ct = (int32_t)std::max<int64_t>(1, (cols + BITS_BC - 1) / BITS_BC);
  const int64_t ncell = (int64_t)H.n_rr * H.n_ct * BITS_NSUB, n = nnz;
  if (ncell > (int64_t)1 << 27 || nnz <= 0 || rows <= 0) { why = "not applicable"; return 0; }
  DevPool pool;
  pool.stream = stream;
  CubTemp tmp;
  hipError_t herr = hipSuccess;
  const char *hwhat = "";
#define GT(call)                                                                                     \
  do {                                                                                               \
    herr = (call);                                                                                   \
    if (herr != hipSuccess) { hwhat = #call; goto hip_failed; }                                      \
  } while (0)
#define POOL_OK()                                                                                    \
  do {                                                                                               \
    if (pool.err != hipSuccess) { herr = pool.err; hwhat = "hipMalloc (temporaries)"; goto hip_failed; } \
  } while (0)
#define LAUNCH(kernel, count, ...)                                                                   \
  do {                                                                                               \
    if ((count) > 0) hipLaunchKernelGGL(kernel, grid_for(count), dim3(PBS), 0, stream, __VA_ARGS__);  \
  } while (0)
#define CUB(...)                                                                                     \
  do {                                                                                               \
    size_t _b = 0;                                                                                   \
    void *_t = nullptr;                                                                              \
    { auto _call = [&](void *d_temp_storage, size_t &temp_storage_bytes) { return __VA_ARGS__; };    \
      GT(_call(_t, _b));                                                                             \
      GT(tmp.reserve(_b));                                                                           \
      _t = tmp.p;                                                                                    \
      GT(_call(_t, _b)); }                                                                           \
  } while (0)
  uint32_t *head = nullptr, *row_of = nullptr, *key_in = nullptr, *idx_in = nullptr, *key = nullptr, *jS = nullptr, *cstart = nullptr,
           *cend = nullptr, *d_start = nullptr;
  std::vector<uint32_t> hs((size_t)ncell), he((size_t)ncell), st32((size_t)ncell + 1);
  std::vector<int64_t> cnt((size_t)ncell + 1, 0), start((size_t)ncell + 1, 0);
  head = pool.get<uint32_t>((size_t)n, true); row_of = pool.get<uint32_t>((size_t)n);
  key_in = pool.get<uint32_t>((size_t)n); idx_in = pool.get<uint32_t>((size_t)n);
  key = pool.get<uint32_t>((size_t)n); jS = pool.get<uint32_t>((size_t)n);
  cstart = pool.get<uint32_t>((size_t)ncell + 1, true); cend = pool.get<uint32_t>((size_t)ncell + 1, true);
  d_start = pool.get<uint32_t>((size_t)ncell + 1);
  POOL_OK();
  LAUNCH(k_row_heads, rows, d_rp, rows, head);
  CUB(incl_max(d_temp_storage, temp_storage_bytes, head, row_of, n, stream));
  LAUNCH(k_bits_keys, n, row_of, d_ci, d_val, n, cols, H.n_ct, (uint32_t)ncell, key_in, idx_in);
  // stable: inside a cell the entries keep the order of the CSR walk (rows ascending), so the sub-range of an entry
  // follows from its position
  CUB(sort_pairs(d_temp_storage, temp_storage_bytes, key_in, key, idx_in, jS, n, 0, bits_for((uint64_t)ncell), stream));
  LAUNCH(k_bits_bounds, n, key, n, (uint32_t)ncell, cstart, cend);
  GT(hipMemcpyAsync(hs.data(), cstart, (size_t)ncell * 4, hipMemcpyDeviceToHost, stream));
  GT(hipMemcpyAsync(he.data(), cend, (size_t)ncell * 4, hipMemcpyDeviceToHost, stream));
  GT(hipStreamSynchronize(stream));
  for (int64_t k = 0; k < ncell; k++) cnt[(size_t)k] = (int64_t)he[(size_t)k] - (int64_t)hs[(size_t)k];
  if (!bits_starts_and_items(H, cnt, start)) { why = "entry array exceeds int32 indexing"; return 0; }
  for (int64_t k = 0; k <= ncell; k++) st32[(size_t)k] = (uint32_t)start[(size_t)k];
  H.ent_len = start[(size_t)ncell] + 8;
  GT(hipMemcpyAsync(d_start, st32.data(), ((size_t)ncell + 1) * 4, hipMemcpyHostToDevice, stream));
  GT(hipMalloc((void **)d_ent, (size_t)H.ent_len * 4 + SLACK_WIDE));
  GT(hipMemsetAsync(*d_ent, 0, (size_t)H.ent_len * 4, stream));
  LAUNCH(k_bits_fill, n, key, jS, row_of, d_ci, n, (uint32_t)ncell, cstart, cend, d_start, *d_ent);
  GT(hipStreamSynchronize(stream));
  GT(hipGetLastError());
  return 1;
hip_failed:
  why = std::string(hwhat) + ": " + hipGetErrorString(herr);
  (void)hipGetLastError();
  if (*d_ent) { (void)hipFree(*d_ent); *d_ent = nullptr; }
  return -1;
#undef GT
#undef POOL_OK
#undef LAUNCH
#undef CUB
}

} // namespace sh
