// semiring.hip.h -- the three semirings of sparseharness's apps as device
// functors.  Arithmetic restates the user functions embedded in the Lift
// kernels (reference: example/{spmv,sssp,bfs}/kernel5.json:3);
// the file is compiled with -ffp-contract=off so that mul and add stay two
// roundings as in the reference's C/OpenCL text.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sh {

// All elements are 4 bytes and travel as raw 32-bit words; T is the view.
struct PlusTimesF32 {
  using T = float;
  static constexpr int id = 0;
  __device__ static inline T identity() { return 0.0f; }
  static constexpr uint32_t identity_bits = 0u;   // (the same word for host code)
  // absorbing(x): mul(x, a) == identity() for every a the matrix may hold -- a column tile of such x words contributes
  // nothing and phase 1 of the tiled plan need not read its entries.  Not for (+,x): 0 * a is -0 for negative a.
  static constexpr bool has_absorbing = false;
  __device__ static inline bool absorbing(uint32_t) { return false; }
  __device__ static inline T mul(T x, T a) { return x * a; }   // mult(l,r) = l*r
  __device__ static inline T add(T acc, T p) { return acc + p; } // add(x,y) = x+y
  // doubleMultiplyAdd: (dpRes*alpha)+(rowIdxPair2*beta).  When beta == 0 the
  // y read is skipped and y*beta is taken as +0.0f (exact for finite y >= 0,
  // the only case the apps produce: app/spmv.cpp:118-120).
  __device__ static inline T epilogue(T dot, T alpha, T y, T beta, bool use_y) {
    return (dot * alpha) + (use_y ? (y * beta) : 0.0f);
  }
  __host__ __device__ static inline bool reads_y(T beta) { return beta != 0.0f; }
  // should_terminate_iteration of a float app (app/sssp.cpp:170)
  __device__ static inline bool differs(T in, T out, double delta) {
    return !((double)fabsf(in - out) < delta);
  }
};

struct MinPlusF32 {
  using T = float;
  static constexpr int id = 1;
  __device__ static inline T identity() { return 3.4028235E38f; }
  static constexpr uint32_t identity_bits = 0x7F7FFFFFu;
  // |x| == FLT_MAX: FLT_MAX + |a| rounds back to FLT_MAX while |a| < 2^103 (the engine checks the matrix' values)
  static constexpr bool has_absorbing = true;
  __device__ static inline bool absorbing(uint32_t xbits) { return (xbits & 0x7FFFFFFFu) == 0x7F7FFFFFu; }
  __device__ static inline T mul(T x, T a) { return fabsf(x) + fabsf(a); } // absadd
  __device__ static inline T add(T acc, T p) {                              // clmin
    return fabsf(acc) < fabsf(p) ? fabsf(acc) : fabsf(p);
  }
  __device__ static inline T epilogue(T dot, T alpha, T y, T beta, bool) {
    T a = fabsf(dot) + fabsf(alpha);
    T b = fabsf(y) + fabsf(beta);
    return fabsf(a) < fabsf(b) ? fabsf(a) : fabsf(b);
  }
  __host__ __device__ static inline bool reads_y(T) { return true; }
  __device__ static inline bool differs(T in, T out, double delta) {
    return !((double)fabsf(in - out) < delta);
  }
};

struct OrAndI32 {
  using T = int32_t;
  static constexpr int id = 2;
  __device__ static inline T identity() { return 0; }
  static constexpr uint32_t identity_bits = 0u;
  static constexpr bool has_absorbing = true;
  __device__ static inline bool absorbing(uint32_t xbits) { return xbits == 0u; }
  __device__ static inline T mul(T x, T a) { return (x != 0) && (a != 0); } // bool_and
  __device__ static inline T add(T acc, T p) { return (acc != 0) || (p != 0); } // bool_or
  __device__ static inline T epilogue(T dot, T alpha, T y, T beta, bool use_y) { // doubleAndOr
    T r1 = (dot != 0) && (alpha != 0);
    T r2 = use_y ? ((y != 0) && (beta != 0)) : 0;
    return r1 || r2;
  }
  __host__ __device__ static inline bool reads_y(T beta) { return beta != 0; }
  // app/bfs.cpp:167: exact equality
  __device__ static inline bool differs(T in, T out, double) { return in != out; }
};

// (max,min) on int32: the SCC app (reference: example/scc/kernel5.json:3 -- int_min, int_max,
// doubleMinMax; identity / padding value INT_MIN, app/scc.cpp:206).
struct MaxMinI32 {
  using T = int32_t;
  static constexpr int id = 3;
  __device__ static inline T identity() { return INT32_MIN; }
  static constexpr uint32_t identity_bits = 0x80000000u;
  static constexpr bool has_absorbing = true;
  __device__ static inline bool absorbing(uint32_t xbits) { return xbits == 0x80000000u; }
  __device__ static inline T mul(T x, T a) { return x < a ? x : a; }       // int_min
  __device__ static inline T add(T acc, T p) { return acc > p ? acc : p; } // int_max
  __device__ static inline T epilogue(T dot, T alpha, T y, T beta, bool use_y) { // doubleMinMax
    const T m1 = dot < alpha ? dot : alpha;
    const T m2 = use_y ? (y < beta ? y : beta) : INT32_MIN;   // min(y, INT_MIN) == INT_MIN for every y
    return m1 > m2 ? m1 : m2;
  }
  __host__ __device__ static inline bool reads_y(T beta) { return beta != INT32_MIN; }
  __device__ static inline bool differs(T in, T out, double) { return in != out; }  // app/scc.cpp:166
};

template <class T> __device__ inline T from_bits(uint32_t u) {
  return __builtin_bit_cast(T, u);
}
template <class T> __device__ inline uint32_t to_bits(T v) {
  return __builtin_bit_cast(uint32_t, v);
}

} // namespace sh
