// tools/lab_pieces.hip -- how fast does the chip read a large buffer in PIECES of S bytes whose
// order is scrambled (phase 2 reads the product array P as ~215-byte pieces, one per column
// tile, each 2.6 MB away from the next)?  Each wave reads whole pieces with 16-byte lanes; a
// multiplicative hash permutes the piece order so that consecutive pieces are far apart.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <utility>
#define CK(x) do { hipError_t r = (x); if (r != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r), __LINE__); exit(1);} } while (0)

// piece p (of n_pieces, each S16 16-byte words) sits at permuted position; lanes cover 16-byte words
__global__ void rd_pieces(const uint4 *buf, uint64_t n_pieces, int S16, int scramble, uint64_t mult, float *sink) {
  const uint64_t total16 = n_pieces * S16;
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  uint32_t a = 0;
  for (; i < total16; i += stride) {
    const uint64_t piece = i / S16, w = i % S16;
    const uint64_t pp = scramble ? (piece * mult) % n_pieces : piece;   // mult coprime with n_pieces
    const uint4 v = buf[pp * S16 + w];
    a += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (a == 0x12345678u) *sink = 1.f;
}

// the same for WRITES: a piece of S16 data words at every STRIDE16 words (STRIDE16 > S16 leaves the rest of the slot alone)
__global__ void wr_pieces(uint4 *buf, uint64_t n_pieces, int S16, int STRIDE16, int scramble, uint64_t mult) {
  const uint64_t total16 = n_pieces * S16;
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < total16; i += stride) {
    const uint64_t piece = i / S16, w = i % S16;
    const uint64_t pp = scramble ? (piece * mult) % n_pieces : piece;
    buf[pp * STRIDE16 + w] = make_uint4((uint32_t)i, 2u, 3u, 4u);
  }
}

int main() {
  const size_t bytes = 768ull << 20;
  uint4 *buf; float *sink;
  CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 64)); CK(hipMemset(buf, 1, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%-10s %-10s %10s\n", "piece_B", "order", "TB/s");
  for (int S : {64, 128, 208, 224, 256, 512, 1024, 4096, 65536}) {
    const int S16 = S / 16;
    uint64_t n_pieces = bytes / S;
    if (n_pieces % 2 == 0) n_pieces -= 1;                    // odd count: any odd multiplier not sharing a factor works
    uint64_t mult = 2654435761ull % n_pieces; while (std::__gcd(mult, n_pieces) != 1) mult++;
    for (int scramble : {0, 1}) {
      std::vector<float> t;
      for (int rep = 0; rep < 6; rep++) {
        CK(hipEventRecord(e0));
        rd_pieces<<<256 * 8, 256>>>(buf, n_pieces, S16, scramble, mult, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
      }
      std::sort(t.begin(), t.end());
      printf("%-10d %-10s %10.2f\n", S, scramble ? "scrambled" : "sequential", (double)n_pieces * S / (t[2] * 1e-3) / 1e12);
    }
  }
  printf("writes: %-8s %-8s %-10s %10s\n", "piece_B", "slot_B", "order", "TB/s");
  for (auto cfg : {std::pair<int, int>{208, 208}, {208, 224}, {208, 256}, {224, 224}, {224, 256}, {256, 256}, {1024, 1024}, {65536, 65536}}) {
    const int S16 = cfg.first / 16, ST16 = cfg.second / 16;
    uint64_t n_pieces = bytes / cfg.second;
    if (n_pieces % 2 == 0) n_pieces -= 1;
    uint64_t mult = 2654435761ull % n_pieces; while (std::__gcd(mult, n_pieces) != 1) mult++;
    for (int scramble : {0, 1}) {
      std::vector<float> t;
      for (int rep = 0; rep < 6; rep++) {
        CK(hipEventRecord(e0));
        wr_pieces<<<256 * 8, 256>>>(buf, n_pieces, S16, ST16, scramble, mult);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
      }
      std::sort(t.begin(), t.end());
      printf("writes: %-8d %-8d %-10s %10.2f\n", cfg.first, cfg.second, scramble ? "scrambled" : "sequential", (double)n_pieces * cfg.first / (t[2] * 1e-3) / 1e12);
    }
  }
  return 0;
}
