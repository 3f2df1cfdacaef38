// kernels.hip.h -- CDNA4 (gfx950) CSR SpMV kernels, templated on the semiring.
//
// What they compute is the per-row contract of every Lift strategy in the
// reference (example/<algo>/kernel*.json:3, inventory in SURVEY.md 2.2):
//   out[r] = epilogue( (+)_j ( x_or_identity(col_j) (x) val_j ), alpha, y[r], beta )
// How they compute it is native: the matrix is streamed once, in CSR order,
// with 16-byte-per-lane coalesced loads; no padded ELLPACK, no global temp
// buffer, no re-reads.
//
// Schedule (built on upload, engine.hip): the row range is cut into
//   * stream blocks: consecutive rows whose non-zeros (<= NNZ_BLK, counted
//     from the 16-byte-aligned start) are staged as PRODUCTS in LDS by one
//     256-thread workgroup and then reduced per row out of LDS by 1..64 lanes
//     per row (wave64 __shfl_xor tree for the cross-lane part);
//   * long-row segments: rows longer than NNZ_BLK are split into SEG_NNZ-sized
//     segments, one workgroup each, whose partial results are combined in
//     segment order by spmv_long_fixup (deterministic, no float atomics).
// Both kinds live in ONE launch (blockIdx < n_stream selects the kind) so the
// heavy tail of a power-law matrix fills the chip together with the body.
//
// HBM-bound (0.23 flop/B): no MFMA by design.
#pragma once
#include "semiring.hip.h"

namespace sh {

constexpr int BS = 256;          // 4 wave64 per workgroup
constexpr int NNZ_BLK = 4096;    // products staged in LDS per stream block (16 KiB)
constexpr int ROWS_BLK = 1024;   // max rows per stream block (row_ptr slice in LDS)
constexpr int SEG_NNZ = 8192;    // non-zeros per long-row segment

struct CsrDev {
  const int32_t *row_ptr;
  const int32_t *col_idx;   // padded to a multiple of 4 (+4) entries
  const uint32_t *val;      // same padding
  int32_t rows;
  int32_t cols;
};

// Optional fused convergence test of the iterative apps
// (should_terminate_iteration, app/sssp.cpp:157-176 / app/bfs.cpp:154-174).
struct StepDev {
  int32_t *changed;         // nullptr: no test
  const uint32_t *prev;     // previous vector; row r compares prev[prev_off + r]
  int64_t prev_off;
  double delta;
};

struct LongSeg { int32_t row, s, e, slot; };
struct LongRow { int32_t row, slot0, nslots, pad; };

template <class SR>
__device__ inline typename SR::T gather_x(const uint32_t *__restrict__ x, int32_t c, int32_t cols) {
  // bounds ladder of kernel5.json:3: idx < 0 or idx >= VLength -> identity
  return ((uint32_t)c < (uint32_t)cols) ? from_bits<typename SR::T>(x[c]) : SR::identity();
}

template <class SR>
__device__ inline void finish_row(int32_t row, typename SR::T dot, const uint32_t *__restrict__ y,
                                  typename SR::T alpha, typename SR::T beta, bool use_y,
                                  uint32_t *__restrict__ out, const StepDev &st) {
  using T = typename SR::T;
  T yv = use_y ? from_bits<T>(y[row]) : SR::identity();
  T o = SR::epilogue(dot, alpha, yv, beta, use_y);
  out[row] = to_bits<T>(o);
  if (st.changed) {
    T in = from_bits<T>(st.prev[st.prev_off + row]);
    if (SR::differs(in, o, st.delta))
      *st.changed = 1;   // benign race: every writer stores 1
  }
}

template <class SR>
__global__ __launch_bounds__(BS) void spmv_csr_kernel(
    CsrDev A, const uint32_t *__restrict__ x, const uint32_t *__restrict__ y,
    typename SR::T alpha, typename SR::T beta, int use_y_i, uint32_t *__restrict__ out,
    const int32_t *__restrict__ blk_row, int32_t n_stream,
    const LongSeg *__restrict__ segs, uint32_t *__restrict__ partial, StepDev st) {
  using T = typename SR::T;
  __shared__ uint32_t prod[NNZ_BLK];
  __shared__ int32_t rp[ROWS_BLK + 1];
  __shared__ uint32_t wred[BS / 64];
  const int tid = threadIdx.x;
  const bool use_y = use_y_i != 0;
  const int b = blockIdx.x;

  if (b < n_stream) {
    // ------------------------------------------------------------ stream block
    const int2 rr = reinterpret_cast<const int2 *>(blk_row)[b];   // (first row, one-past-last row)
    const int r0 = rr.x;
    const int nr = rr.y - r0;
    for (int i = tid; i <= nr; i += BS)
      rp[i] = A.row_ptr[r0 + i];
    __syncthreads();
    const int s = rp[0], e = rp[nr];
    const int base = s & ~3;          // 16-byte aligned start (<= 3 foreign entries masked)
    // Phase 1: products -> LDS.  e - base <= NNZ_BLK by construction.
    constexpr int IT = NNZ_BLK / (BS * 4);
    int4 c[IT];
    uint4 v[IT];
#pragma unroll
    for (int k = 0; k < IT; k++) {
      const int i = base + (k * BS + tid) * 4;
      if (i < e) {
        c[k] = *reinterpret_cast<const int4 *>(A.col_idx + i);
        v[k] = *reinterpret_cast<const uint4 *>(A.val + i);
      }
    }
#pragma unroll
    for (int k = 0; k < IT; k++) {
      const int i = base + (k * BS + tid) * 4;
      if (i < e) {
        uint4 p;
        p.x = (i + 0 >= s && i + 0 < e) ? to_bits<T>(SR::mul(gather_x<SR>(x, c[k].x, A.cols), from_bits<T>(v[k].x))) : 0u;
        p.y = (i + 1 >= s && i + 1 < e) ? to_bits<T>(SR::mul(gather_x<SR>(x, c[k].y, A.cols), from_bits<T>(v[k].y))) : 0u;
        p.z = (i + 2 >= s && i + 2 < e) ? to_bits<T>(SR::mul(gather_x<SR>(x, c[k].z, A.cols), from_bits<T>(v[k].z))) : 0u;
        p.w = (i + 3 >= s && i + 3 < e) ? to_bits<T>(SR::mul(gather_x<SR>(x, c[k].w, A.cols), from_bits<T>(v[k].w))) : 0u;
        *reinterpret_cast<uint4 *>(&prod[i - base]) = p;
      }
    }
    __syncthreads();
    // Phase 2: per-row reduction out of LDS with lpr (1..64) lanes per row.
    int lpr = 1;
    while (lpr < 64 && nr * lpr * 2 <= BS)
      lpr <<= 1;
    const int g = tid / lpr, l = tid & (lpr - 1), ng = BS / lpr;
    for (int row = g; row < nr; row += ng) {
      T acc = SR::identity();
      const int je = rp[row + 1] - base;
      for (int j = rp[row] - base + l; j < je; j += lpr)
        acc = SR::add(acc, from_bits<T>(prod[j]));
      for (int o = lpr >> 1; o > 0; o >>= 1)
        acc = SR::add(acc, from_bits<T>(__shfl_xor(to_bits<T>(acc), o, 64)));
      if (l == 0)
        finish_row<SR>(r0 + row, acc, y, alpha, beta, use_y, out, st);
    }
  } else {
    // ------------------------------------------------------- long-row segment
    const LongSeg sg = segs[b - n_stream];
    const int s = sg.s, e = sg.e;
    const int base = s & ~3;
    T acc = SR::identity();
    for (int i0 = base + tid * 4; i0 < e; i0 += BS * 4 * 2) {
      int4 c[2];
      uint4 v[2];
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int i = i0 + k * BS * 4;
        if (i < e) {
          c[k] = *reinterpret_cast<const int4 *>(A.col_idx + i);
          v[k] = *reinterpret_cast<const uint4 *>(A.val + i);
        }
      }
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int i = i0 + k * BS * 4;
        if (i < e) {
          if (i + 0 >= s && i + 0 < e) acc = SR::add(acc, SR::mul(gather_x<SR>(x, c[k].x, A.cols), from_bits<T>(v[k].x)));
          if (i + 1 >= s && i + 1 < e) acc = SR::add(acc, SR::mul(gather_x<SR>(x, c[k].y, A.cols), from_bits<T>(v[k].y)));
          if (i + 2 >= s && i + 2 < e) acc = SR::add(acc, SR::mul(gather_x<SR>(x, c[k].z, A.cols), from_bits<T>(v[k].z)));
          if (i + 3 >= s && i + 3 < e) acc = SR::add(acc, SR::mul(gather_x<SR>(x, c[k].w, A.cols), from_bits<T>(v[k].w)));
        }
      }
    }
    for (int o = 32; o > 0; o >>= 1)
      acc = SR::add(acc, from_bits<T>(__shfl_xor(to_bits<T>(acc), o, 64)));
    if ((tid & 63) == 0)
      wred[tid >> 6] = to_bits<T>(acc);
    __syncthreads();
    if (tid == 0) {
      T t = from_bits<T>(wred[0]);
#pragma unroll
      for (int w = 1; w < BS / 64; w++)
        t = SR::add(t, from_bits<T>(wred[w]));
      partial[sg.slot] = to_bits<T>(t);
    }
  }
}

// Combine the segment partials of each long row in segment order.
template <class SR>
__global__ __launch_bounds__(64) void spmv_long_fixup(
    const LongRow *__restrict__ rows, int32_t n_long, const uint32_t *__restrict__ partial,
    const uint32_t *__restrict__ y, typename SR::T alpha, typename SR::T beta, int use_y_i,
    uint32_t *__restrict__ out, StepDev st) {
  using T = typename SR::T;
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n_long)
    return;
  const LongRow lr = rows[i];
  T acc = SR::identity();
  for (int k = 0; k < lr.nslots; k++)
    acc = SR::add(acc, from_bits<T>(partial[lr.slot0 + k]));
  finish_row<SR>(lr.row, acc, y, alpha, beta, use_y_i != 0, out, st);
}

} // namespace sh
