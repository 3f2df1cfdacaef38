// tools/lab_copy.hip -- what does a read+write stream reach on this chip?  (phase 1 of the tiled
// plan reads 0.73 GB and writes 0.60 GB per SpMV.)  Variants: loads in flight per thread, threads
// per workgroup, workgroups per CU, plain / non-temporal stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t r = (x); if (r != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r), __LINE__); exit(1);} } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ void copyk(const v4f *__restrict__ src, v4f *__restrict__ dst, size_t n4) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    v4f v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = src[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride);
      else dst[i + u * stride] = v[u];
    }
  }
  for (; i < n4; i += stride) dst[i] = src[i];
}

template <int U, bool NT> float run(const v4f *s, v4f *d, size_t n4, int wgs, int threads) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> t;
  for (int rep = 0; rep < 7; rep++) {
    CK(hipEventRecord(e0));
    copyk<U, NT><<<wgs, threads>>>(s, d, n4);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  return t[2];
}

int main() {
  const size_t bytes = 640ull << 20, n4 = bytes / 16;
  v4f *s, *d; CK(hipMalloc(&s, bytes)); CK(hipMalloc(&d, bytes)); CK(hipMemset(s, 1, bytes)); CK(hipMemset(d, 0, bytes));
  printf("%-8s %-8s %-6s %-4s %10s\n", "wgs", "threads", "unroll", "nt", "TB/s(r+w)");
  for (int threads : {256, 1024})
    for (int per_cu : {1, 2, 4, 8}) {
      const int wgs = 256 * per_cu;
      if (threads == 1024 && per_cu > 2) continue;
      float a = run<1, false>(s, d, n4, wgs, threads), b = run<4, false>(s, d, n4, wgs, threads), c = run<8, false>(s, d, n4, wgs, threads),
            e = run<4, true>(s, d, n4, wgs, threads);
      printf("%-8d %-8d %-6d %-4d %10.2f\n", wgs, threads, 1, 0, 2.0 * bytes / (a * 1e-3) / 1e12);
      printf("%-8d %-8d %-6d %-4d %10.2f\n", wgs, threads, 4, 0, 2.0 * bytes / (b * 1e-3) / 1e12);
      printf("%-8d %-8d %-6d %-4d %10.2f\n", wgs, threads, 8, 0, 2.0 * bytes / (c * 1e-3) / 1e12);
      printf("%-8d %-8d %-6d %-4d %10.2f\n", wgs, threads, 4, 1, 2.0 * bytes / (e * 1e-3) / 1e12);
    }
  return 0;
}
