// tools/lab_gather.hip -- microbenchmarks that size the SpMV design space on
// MI355X (not part of the product): streaming ceiling, random 4-byte gather
// rate vs table size (L2 / Infinity Cache / HBM), LDS gather rate, LDS float
// atomic rate.  Build: hipcc --offload-arch=gfx950 -O3 tools/lab_gather.hip -o tools/lab_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t r = (x); if (r != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r), __LINE__); exit(1);} } while (0)

__device__ inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

__global__ void fill_idx(int32_t *idx, size_t n, uint32_t mod, uint32_t seed) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) idx[i] = (int32_t)(hash32((uint32_t)i * 2654435761u + seed) % mod);
}
__global__ void fill_f(float *p, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) p[i] = (float)(i & 15);
}
// float4 copy
__global__ void copy4(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n4) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t st = (size_t)gridDim.x * blockDim.x;
  for (; i < n4; i += st) b[i] = a[i];
}
// read-only stream of two arrays (like col+val), 16B/lane, reduce to keep alive
__global__ void read2(const int4 *__restrict__ a, const float4 *__restrict__ b, size_t n4, float *sink) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t st = (size_t)gridDim.x * blockDim.x;
  float acc = 0;
  for (; i < n4; i += st) { int4 c = a[i]; float4 v = b[i]; acc += v.x + v.y + v.z + v.w + (float)(c.x ^ c.y ^ c.z ^ c.w); }
  if (acc == 1.2345f) *sink = acc;
}
// random gather: idx streamed 16B/lane, 4 gathers per lane-iteration
__global__ void gather4(const int4 *__restrict__ idx, const float *__restrict__ x, size_t n4, float *sink) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t st = (size_t)gridDim.x * blockDim.x;
  float acc = 0;
  for (; i + st < n4; i += 2 * st) {
    int4 c = idx[i]; int4 d = idx[i + st];
    acc += x[c.x] + x[c.y] + x[c.z] + x[c.w] + x[d.x] + x[d.y] + x[d.z] + x[d.w];
  }
  if (acc == 1.2345f) *sink = acc;
}
// LDS gather: x tile (TILE floats) in LDS, indices streamed
template <int TILE>
__global__ __launch_bounds__(1024) void lds_gather(const int4 *__restrict__ idx, const float *__restrict__ x, size_t n4, float *sink) {
  extern __shared__ float xs[];
  for (int i = threadIdx.x; i < TILE; i += blockDim.x) xs[i] = x[i];
  __syncthreads();
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t st = (size_t)gridDim.x * blockDim.x;
  float acc = 0;
  for (; i < n4; i += st) { int4 c = idx[i]; acc += xs[c.x & (TILE - 1)] + xs[c.y & (TILE - 1)] + xs[c.z & (TILE - 1)] + xs[c.w & (TILE - 1)]; }
  if (acc == 1.2345f) *sink = acc;
}
// LDS float atomic add at random addresses, values + indices streamed
template <int TILE>
__global__ __launch_bounds__(1024) void lds_atomic(const int4 *__restrict__ idx, const float4 *__restrict__ v, size_t n4, float *out) {
  extern __shared__ float ys[];
  for (int i = threadIdx.x; i < TILE; i += blockDim.x) ys[i] = 0;
  __syncthreads();
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t st = (size_t)gridDim.x * blockDim.x;
  for (; i < n4; i += st) {
    int4 c = idx[i]; float4 p = v[i];
    atomicAdd(&ys[c.x & (TILE - 1)], p.x); atomicAdd(&ys[c.y & (TILE - 1)], p.y);
    atomicAdd(&ys[c.z & (TILE - 1)], p.z); atomicAdd(&ys[c.w & (TILE - 1)], p.w);
  }
  __syncthreads();
  if (threadIdx.x == 0 && ys[0] == 1.2345f) out[0] = ys[0];
}

template <class F> float timeit(F f, int reps = 5) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  std::vector<float> t;
  for (int r = 0; r < reps; r++) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms); }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

int main() {
  const size_t N = 200u * 1000 * 1000;   // "nnz"
  int32_t *idx; float *val, *x, *sink, *dst;
  CK(hipMalloc(&idx, N * 4)); CK(hipMalloc(&val, N * 4)); CK(hipMalloc(&dst, N * 4));
  CK(hipMalloc(&x, 512u << 20)); CK(hipMalloc(&sink, 64));
  fill_f<<<2048, 256>>>(val, N); fill_f<<<2048, 256>>>(x, (512u << 20) / 4);
  CK(hipDeviceSynchronize());
  const int G = 256 * 8;
  float ms = timeit([&] { copy4<<<G, 256>>>((const float4 *)val, (float4 *)dst, N / 4); });
  printf("copy4      %8.3f ms  %7.1f GB/s (r+w)\n", ms, 2.0 * N * 4 / ms / 1e6);
  ms = timeit([&] { read2<<<G, 256>>>((const int4 *)idx, (const float4 *)val, N / 4, sink); });
  printf("read2      %8.3f ms  %7.1f GB/s (col+val stream only)\n", ms, 2.0 * N * 4 / ms / 1e6);
  for (size_t kb : {256, 1024, 2048, 4096, 8192, 16384, 40000, 160000, 500000}) {
    uint32_t elems = (uint32_t)(kb * 1024 / 4);
    fill_idx<<<2048, 256>>>(idx, N, elems, 17);
    CK(hipDeviceSynchronize());
    for (int g : {G, 256 * 16}) {
      ms = timeit([&] { gather4<<<g, 256>>>((const int4 *)idx, x, N / 4, sink); });
      printf("gather4 table %7zu KB grid %5d  %8.3f ms  %7.2f Ggather/s\n", kb, g, ms, N / ms / 1e6);
    }
  }
  // sorted-ish locality: indices within a window
  fill_idx<<<2048, 256>>>(idx, N, 32768, 3); CK(hipDeviceSynchronize());
  ms = timeit([&] { lds_gather<32768><<<256, 1024, 32768 * 4>>>((const int4 *)idx, x, N / 4, sink); });
  printf("lds_gather 128KB tile, 256 WG x1024   %8.3f ms  %7.2f Ggather/s\n", ms, N / ms / 1e6);
  ms = timeit([&] { lds_gather<16384><<<512, 1024, 16384 * 4>>>((const int4 *)idx, x, N / 4, sink); });
  printf("lds_gather  64KB tile, 512 WG x1024   %8.3f ms  %7.2f Ggather/s\n", ms, N / ms / 1e6);
  ms = timeit([&] { lds_atomic<32768><<<256, 1024, 32768 * 4>>>((const int4 *)idx, (const float4 *)val, N / 4, sink); });
  printf("lds_atomic 128KB tile, 256 WG x1024   %8.3f ms  %7.2f Gadd/s\n", ms, N / ms / 1e6);
  ms = timeit([&] { lds_atomic<16384><<<512, 1024, 16384 * 4>>>((const int4 *)idx, (const float4 *)val, N / 4, sink); });
  printf("lds_atomic  64KB tile, 512 WG x1024   %8.3f ms  %7.2f Gadd/s\n", ms, N / ms / 1e6);
  return 0;
}
