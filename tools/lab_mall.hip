// tools/lab_mall.hip -- does a buffer written by one kernel and read by the next stay in
// the 256 MiB Infinity Cache?  (sizing experiment for the tiled plan's product array P)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t r = (x); if (r != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r), __LINE__); exit(1);} } while (0)
__global__ void wr(float4 *p, size_t n4, float v) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i < n4; i += st) p[i] = make_float4(v, v, v, v);
}
__global__ void rd(const float4 *p, size_t n4, float *sink) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  float a = 0;
  for (; i < n4; i += st) { float4 v = p[i]; a += v.x + v.y + v.z + v.w; }
  if (a == 1.2345f) *sink = a;
}
// stream "other" traffic between the write and the read (like phase 1's input stream)
__global__ void rd2(const float4 *p, size_t n4, float *sink) { 
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  float a = 0;
  for (; i < n4; i += st) { float4 v = p[i]; a += v.x * v.y; }
  if (a == 1.2345f) *sink = a;
}
int main() {
  float *buf, *other, *sink;
  CK(hipMalloc(&buf, 1ull << 30)); CK(hipMalloc(&other, 1ull << 30)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(other, 0, 1ull << 30));
  hipEvent_t e[4]; for (auto &x : e) CK(hipEventCreate(&x));
  for (size_t mb : {16, 32, 64, 96, 128, 192, 256, 512, 1024}) {
    for (size_t other_mb : {(size_t)0, mb * 2}) {
      size_t n4 = mb * 1024 * 1024 / 16, o4 = other_mb * 1024 * 1024 / 16;
      if (other_mb > 1024) continue;
      std::vector<float> tw, tr, to;
      for (int rep = 0; rep < 7; rep++) {
        CK(hipEventRecord(e[0])); wr<<<2048, 256>>>((float4 *)buf, n4, (float)rep);
        CK(hipEventRecord(e[1])); if (o4) rd2<<<2048, 256>>>((const float4 *)other, o4, sink);
        CK(hipEventRecord(e[2])); rd<<<2048, 256>>>((const float4 *)buf, n4, sink);
        CK(hipEventRecord(e[3])); CK(hipEventSynchronize(e[3]));
        float a, b, c; CK(hipEventElapsedTime(&a, e[0], e[1])); CK(hipEventElapsedTime(&b, e[1], e[2])); CK(hipEventElapsedTime(&c, e[2], e[3]));
        tw.push_back(a); to.push_back(b); tr.push_back(c);
      }
      std::sort(tw.begin(), tw.end()); std::sort(tr.begin(), tr.end()); std::sort(to.begin(), to.end());
      printf("buf %5zu MB, other-stream %5zu MB between: write %7.1f GB/s  other %7.1f GB/s  read-back %7.1f GB/s\n", mb, other_mb,
             mb * 1.048576 / tw[3], other_mb ? other_mb * 1.048576 / to[3] : 0.0, mb * 1.048576 / tr[3]);
    }
  }
  return 0;
}
