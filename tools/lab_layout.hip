// tools/lab_layout.hip -- does WHERE the (bin, tile) pieces of the product array P lie change how fast
// phase 2 can gather them?  Phase 2 reads, per row bin, one ~208-byte piece out of every column tile's run
// of P.  With P tile-major (today) the 306 pieces of a bin are 1.5 MB apart: 306 different pages per bin.
// A bin-BLOCKED layout [block of BB bins][tile][bin in block] keeps them inside BB * 64 KiB.
// One persistent 256-thread workgroup per CU walks "bins" exactly like spmv_tiled_phase2s's loaders
// (bins b0, b0 + G, ...; neighbouring bins on one XCD) with 16-byte lanes, 16 loads in flight per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t r = (x); if (r != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r), __LINE__); exit(1);} } while (0)

constexpr int TILES = 306;

// layout: 0 tile-major, 1 bin-blocked (BB), 2 bin-major (sequential per bin)
template <int PG>   // 16-byte groups per piece
__global__ __launch_bounds__(256) void gather_bins(const uint4 *P, int n_bins, int layout, int BB, float *sink) {
  const int G = gridDim.x;
  const int b0 = (G % 8 == 0) ? (int)(blockIdx.x % 8) * (G / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const int tid = threadIdx.x;
  constexpr int NG = TILES * PG;                  // groups per bin
  constexpr int IT = (NG + 255) / 256;
  uint32_t a = 0;
  const size_t tile_stride = (size_t)n_bins * PG;   // tile-major: a tile's run
  for (int b = b0; b < n_bins; b += G) {
    uint4 v[IT];
#pragma unroll
    for (int k = 0; k < IT; k++) {
      const int g = min(tid + k * 256, NG - 1);
      const int piece = g / PG, w = g % PG;
      size_t at;
      if (layout == 0) at = (size_t)piece * tile_stride + (size_t)b * PG + w;
      else if (layout == 1) at = (size_t)(b / BB) * ((size_t)TILES * BB * PG) + (size_t)piece * (BB * PG) + (size_t)(b % BB) * PG + w;
      else at = (size_t)b * NG + g;
      v[k] = P[at];
    }
#pragma unroll
    for (int k = 0; k < IT; k++) a += v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
  }
  if (a == 0x12345678u) *sink = 1.f;
}

template <int PG>
static void run(const uint4 *buf, float *sink, hipEvent_t e0, hipEvent_t e1, size_t bytes, int grid) {
  int n_bins = (int)(bytes / 16 / ((size_t)TILES * PG));
  n_bins -= n_bins % 1024;   // every block of every blocked layout is whole: the largest address is below n_bins * TILES * PG
  if ((size_t)n_bins * TILES * PG * 16 > bytes || n_bins <= 0) { printf("geometry error\n"); exit(1); }
  struct Cfg { int layout, BB; const char *name; };
  for (Cfg c : {Cfg{0, 0, "tile-major"}, Cfg{1, 16, "blocked-16"}, Cfg{1, 64, "blocked-64"}, Cfg{1, 256, "blocked-256"}, Cfg{1, 1024, "blocked-1024"}, Cfg{2, 0, "bin-major"}}) {
    std::vector<float> t;
    for (int rep = 0; rep < 7; rep++) {
      CK(hipEventRecord(e0));
      gather_bins<PG><<<grid, 256>>>(buf, n_bins, c.layout, c.BB, sink);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    printf("%-6d wg=%-4d %-14s %8.1f us %8.2f TB/s  (%d bins)\n", PG * 16, grid, c.name, t[3] * 1e3, (double)n_bins * TILES * PG * 16 / (t[3] * 1e-3) / 1e12, n_bins);
  }
}

int main() {
  const size_t bytes = 480ull << 20;
  uint4 *buf; float *sink;
  CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 64)); CK(hipMemset(buf, 1, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%-6s %-14s %11s %13s\n", "pieceB", "layout", "time", "rate");
  for (int grid : {256, 512}) {
    run<13>(buf, sink, e0, e1, bytes, grid);
    run<16>(buf, sink, e0, e1, bytes, grid);
    run<27>(buf, sink, e0, e1, bytes, grid);
  }
  return 0;
}
