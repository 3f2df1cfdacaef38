// tools/lab_slab.hip -- is the two-phase plan bound by HBM or by the fabric in front of the
// 256 MiB Infinity Cache?  Decides whether it pays to run phase 2 of a row slab right behind
// phase 1 of the same slab, so that the products P are re-read (and overwritten) while they
// are still cache-resident.
//
//   part 1  ceilings: stream-read / stream-write a buffer of S MB over and over (S <= ~200 MB
//           stays in the Infinity Cache), persistent 256 x 1024 launch
//   part 2  the plan's traffic mix (per 16 B of P: 24 B of matrix stream read, 16 B P written,
//           16 B P read) as
//             two-pass   pass 1 reads 2/3 of M and writes all of P, pass 2 reads P and the rest
//             slabs      one persistent launch, per slab: read M part, write P slab s,
//                        read P slab s-1 (another workgroup's stripe: no L2 hits);
//                        P laid out linearly (every slab its own addresses) or as a ring of
//                        two slabs (addresses reused: dirty lines can be overwritten in cache)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t r = (x); if (r != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r), __LINE__); exit(1);} } while (0)

constexpr int NT = 1024;

__global__ __launch_bounds__(NT) void rd_loop(const float4 *p, size_t n4, int reps, float *sink) {
  float a = 0;
  const size_t per = n4 / gridDim.x;   // float4 per workgroup
  for (int r = 0; r < reps; r++) {
    // rotate the stripes so that a workgroup never re-reads what its own XCD's L2 holds
    const float4 *q = p + ((blockIdx.x + (size_t)r * 3) % gridDim.x) * per;
    for (size_t i = threadIdx.x; i + 3 * NT < per; i += 4 * NT) {
      float4 v0 = q[i], v1 = q[i + NT], v2 = q[i + 2 * NT], v3 = q[i + 3 * NT];
      a += v0.x + v1.y + v2.z + v3.w;
    }
  }
  if (a == 1.2345f) *sink = a;
}
__global__ __launch_bounds__(NT) void wr_loop(float4 *p, size_t n4, int reps) {
  const size_t per = n4 / gridDim.x;
  for (int r = 0; r < reps; r++) {
    float4 *q = p + ((blockIdx.x + (size_t)r * 3) % gridDim.x) * per;
    const float f = (float)r;
    for (size_t i = threadIdx.x; i + 3 * NT < per; i += 4 * NT) {
      q[i] = make_float4(f, f, f, f); q[i + NT] = make_float4(f, f, f, f);
      q[i + 2 * NT] = make_float4(f, f, f, f); q[i + 3 * NT] = make_float4(f, f, f, f);
    }
  }
}

// One persistent launch.  Slab s of P occupies P[(s * slab4) % ring4 ...).  Per inner iteration a
// thread reads mr float4 of M, writes 2 float4 of P (slab s) and reads 2 float4 of P (slab s-1).
// SC1: P goes through write-through (sc1) stores and sc1 loads, as an in-launch hand-off needs
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st16(float4 *p, float4 v, bool sc1) {
  if (sc1) { const v4f w = {v.x, v.y, v.z, v.w}; asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(w) : "memory"); }
  else *p = v;
}
// (both loads of an iteration fly together; one wait behind the second)
__device__ __forceinline__ void ld16x2(const float4 *p, const float4 *q, float4 &a, float4 &b) {
  v4f va, vb;
  asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(va), "=&v"(vb) : "v"(p), "v"(q) : "memory");
  a = make_float4(va.x, va.y, va.z, va.w); b = make_float4(vb.x, vb.y, vb.z, vb.w);
}
template <int MR, bool DO_W, bool DO_R, bool SC1 = false>
__global__ __launch_bounds__(NT) void slab_loop(const float4 *M, float4 *P, size_t m4_per_slab, size_t slab4, size_t ring4,
                                                int s0, int s1, float *sink) {
  float a = 0;
  const int G = gridDim.x, w = blockIdx.x;
  const size_t pper = slab4 / G, mper = m4_per_slab / G;
  for (int s = s0; s < s1; s++) {
    const float4 *m = M + (size_t)s * m4_per_slab + (size_t)w * mper;
    float4 *pw = P + ((size_t)s * slab4) % ring4 + (size_t)w * pper;
    const float4 *pr = P + ((size_t)(s + (DO_W ? -1 : 0)) * slab4) % ring4 + (size_t)((w + 3) % G) * pper;
    const bool rd = DO_R && (s > 0 || !DO_W);
    const size_t iters = pper / (2 * NT);
    for (size_t it = 0; it < iters; it++) {
      const size_t ip = it * 2 * NT + threadIdx.x, im = it * MR * NT + threadIdx.x;
      float4 v[MR > 0 ? MR : 1], r0 = make_float4(0, 0, 0, 0), r1 = r0;
#pragma unroll
      for (int k = 0; k < MR; k++) v[k] = (im + k * NT < mper) ? m[im + k * NT] : make_float4(0, 0, 0, 0);
      if (rd) { if constexpr (SC1) { ld16x2(pr + ip, pr + ip + NT, r0, r1); } else { r0 = pr[ip]; r1 = pr[ip + NT]; } }
      float t = 0;
      v[0] = MR > 0 ? v[0] : make_float4(0, 0, 0, 0);
#pragma unroll
      for (int k = 0; k < MR; k++) t += v[k].x;
      if constexpr (DO_W) { st16(pw + ip, make_float4(t, t, t, t), SC1); st16(pw + ip + NT, make_float4(t, v[0].y, t, t), SC1); }
      a += r0.x + r1.y + t;
    }
  }
  if (a == 1.2345f) *sink = a;
}

int main() {
  const size_t GB = 1ull << 30, MB = 1ull << 20;
  float *M, *P, *sink;
  CK(hipMalloc(&M, 2 * GB)); CK(hipMalloc(&P, GB)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(M, 0, 2 * GB)); CK(hipMemset(P, 0, GB));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](auto fn) {
    std::vector<float> t;
    for (int rep = 0; rep < 5; rep++) {
      CK(hipEventRecord(e0)); fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[2];
  };
  printf("part 1: ceilings (persistent 256 x 1024, buffer swept `reps` times per launch)\n");
  for (size_t mb : {32, 64, 128, 192, 256, 512, 2048}) {
    const size_t n4 = mb * MB / 16;
    const int reps = (int)std::max<size_t>(2, 4096 / mb);
    float tr = timeit([&] { rd_loop<<<256, NT>>>((const float4 *)M, n4, reps, sink); });
    float tw = timeit([&] { wr_loop<<<256, NT>>>((float4 *)M, n4, reps); });
    printf("  buffer %5zu MB x %3d sweeps: read %6.2f TB/s   write %6.2f TB/s\n", mb, reps,
           mb * 1.048576e-3 * reps / tr, mb * 1.048576e-3 * reps / tw);
  }
  CK(hipMemset(M, 0, 2 * GB));
  printf("part 2: plan mix, P = 512 MB written + read, M = 768 MB read (1.79 GB per SpMV-equivalent)\n");
  const size_t Ptot = 512 * MB, Mtot = 768 * MB;
  {
    // two-pass: pass 1 = M (2/3... here MR=3 per 2 P float4 => all of M) + write; pass 2 = read P only
    const size_t slab4 = Ptot / 16, m4 = Mtot / 16;
    float t = timeit([&] {
      slab_loop<3, true, false><<<256, NT>>>((const float4 *)M, (float4 *)P, m4, slab4, slab4, 0, 1, sink);
      slab_loop<0, false, true><<<256, NT>>>((const float4 *)M, (float4 *)P, 0, slab4, slab4, 0, 1, sink);
    });
    printf("  two-pass (today's structure)           : %7.1f us  %5.2f TB/s\n", t * 1e3, (Ptot * 2 + Mtot) * 1e-9 / t);
  }
  for (int S : {4, 8, 16, 32, 64}) {
    const size_t slab4 = Ptot / 16 / S, m4 = Mtot / 16 / S;
    for (int ring : {0, 1}) {
      const size_t ring4 = ring ? 2 * slab4 : slab4 * S;
      // S + 1 steps: the last step only reads (slab S-1); emulate with one extra slab index (wraps in M/P: harmless)
      float t = timeit([&] {
        slab_loop<3, true, true><<<256, NT>>>((const float4 *)M, (float4 *)P, m4, slab4, ring4, 0, S, sink);
        slab_loop<0, false, true><<<256, NT>>>((const float4 *)M, (float4 *)P, 0, slab4, ring4, S - 1, S, sink);
      });
      printf("  %2d slabs of %5.1f MB P, %s: %7.1f us  %5.2f TB/s\n", S, Ptot / 1048576.0 / S, ring ? "ring of 2 slabs  " : "linear addresses ",
             t * 1e3, (Ptot * 2 + Mtot) * 1e-9 / t);
    }
  }
  printf("part 3: the same with write-through (sc1) P stores and sc1 P loads\n");
  for (int S : {8, 16, 32}) {
    const size_t slab4 = Ptot / 16 / S, m4 = Mtot / 16 / S;
    for (int ring : {0, 1}) {
      const size_t ring4 = ring ? 2 * slab4 : slab4 * S;
      float t = timeit([&] {
        slab_loop<3, true, true, true><<<256, NT>>>((const float4 *)M, (float4 *)P, m4, slab4, ring4, 0, S, sink);
        slab_loop<0, false, true, true><<<256, NT>>>((const float4 *)M, (float4 *)P, 0, slab4, ring4, S - 1, S, sink);
      });
      printf("  %2d slabs of %5.1f MB P, %s: %7.1f us  %5.2f TB/s\n", S, Ptot / 1048576.0 / S, ring ? "ring of 2 slabs  " : "linear addresses ",
             t * 1e3, (Ptot * 2 + Mtot) * 1e-9 / t);
    }
  }
  return 0;
}
