// bits.hip.h -- the (or,and) semiring on bits: a third execution plan for SH_OR_AND_I32 launches.
//
// The BFS app's kernel (example/bfs/kernel5.json:3) computes, per row,
//   out[r] = ( OR_j ( x[col_j] != 0  &&  a_rj != 0 ) && alpha ) || ( y[r] && beta )
// i.e. every product is ONE bit and only the positions of the non-zero x matter.  The float plans move 4 bytes
// per product and stage 4 bytes per column; here
//   * x is first condensed into a bitmap (one launch, 4 B read per column);
//   * the matrix is stored once more as bare coordinates, 4 bytes per entry with a non-zero value, cut into
//     BLOCKS of BR rows x BC columns: the block's slice of the x bitmap (BC / 8 = 64 KiB) and its slice of the
//     result bitmap (BR / 8 = 32 KiB) both sit in LDS, an entry is {column in the block: 19 bits, row in its
//     8192-row sub-range: 13 bits}, entries ordered by row so that the sub-range is known from the position;
//   * a workgroup streams a block's entries (16 B per lane), tests the x bit in LDS and ORs hits into the LDS
//     result bitmap (ds_or_b32, no return), then writes its 32 KiB partial bitmap;
//   * a last launch ORs the partial bitmaps of a row range's column blocks and applies the epilogue / the fused
//     convergence test / the row -> element mapping of the multi-GPU pieces, one thread per row.
// HBM bytes per iteration: 4 B per entry + ~17 B per row, against ~10 B per entry + 12 B per row of the x-tiled
// plan: R-MAT-23 (134 M entries) 0.68 GB instead of 1.21 GB algorithmic, and no product array.
// Bit-exact by construction (the semiring's values are 0 / 1).
#pragma once
#include "kernels.hip.h"

namespace sh {

constexpr int BITS_BC = 1 << 19;          // columns per block: 64 KiB of x bits in LDS
constexpr int BITS_BR = 1 << 18;          // rows per block: 32 KiB of result bits in LDS
constexpr int BITS_SUB = 1 << 13;         // rows per sub-range (the row field of an entry)
constexpr int BITS_NSUB = BITS_BR / BITS_SUB;   // 32 sub-ranges per block
constexpr int BITS_TBS = 1024;
constexpr uint32_t BITS_COL_MASK = BITS_BC - 1;
static_assert(BITS_BC == (1 << 19) && BITS_SUB == (1 << 13), "an entry is 19 + 13 bits");

// One work item: the entries [s, e) of block (rr, ct) -- multiples of 8 -- which cover the sub-ranges [sub0, sub1) of
// row range rr; soff = index in bsub[] of the item's sub-range offsets (sub1 - sub0 + 1 values, absolute entry
// positions, multiples of 8: a sub-range is padded to whole octets with copies of its last entry).
// Its partial result bitmap is partial[item * BITS_BR / 32 ...].
struct BitsItem { int32_t rr, ct, s, e, sub0, sub1, soff, pad; };

// x -> bitmap: bit c of xbits = (x[c] != 0).  A thread takes 4 consecutive columns (one 16-byte load), eight
// neighbouring lanes OR their nibbles into one 32-bit word (three xor-shuffles), the first of them stores it.
static __global__ __launch_bounds__(256) void bits_pack_x(const uint32_t *__restrict__ x, int32_t cols, uint32_t *__restrict__ xbits,
                                                   int64_t n_words32, const int32_t *gate) {
  if (gate != nullptr && *gate == 0) return;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n_words32 * 8) return;   // (whole waves: n_words32 is a multiple of 8)
  const int64_t c = t * 4;
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (c + 3 < cols && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    v = *reinterpret_cast<const uint4 *>(x + c);
  } else {
    if (c < cols) v.x = x[c];
    if (c + 1 < cols) v.y = x[c + 1];
    if (c + 2 < cols) v.z = x[c + 2];
    if (c + 3 < cols) v.w = x[c + 3];
  }
  const int lane = threadIdx.x & 63;
  uint32_t w = ((v.x != 0u ? 1u : 0u) | (v.y != 0u ? 2u : 0u) | (v.z != 0u ? 4u : 0u) | (v.w != 0u ? 8u : 0u)) << (4 * (lane & 7));
  w |= __shfl_xor(w, 1, 64);
  w |= __shfl_xor(w, 2, 64);
  w |= __shfl_xor(w, 4, 64);
  if ((lane & 7) == 0) xbits[t >> 3] = w;
}

static __global__ __launch_bounds__(BITS_TBS) void bits_blocks(const BitsItem *__restrict__ items, const uint32_t *__restrict__ ent,
                                                        const int32_t *__restrict__ bsub, const uint32_t *__restrict__ xbits,
                                                        uint32_t *__restrict__ partial, const int32_t *gate) {
  __shared__ uint32_t xs[BITS_BC / 32];
  __shared__ uint32_t os[BITS_BR / 32];
  __shared__ int32_t offs[BITS_NSUB + 1];   // first 16-byte group of every sub-range of the item (+ the end); even numbers
  if (gate != nullptr && *gate == 0) return;
  const BitsItem it = items[blockIdx.x];
  const int tid = threadIdx.x;
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(xbits + (size_t)it.ct * (BITS_BC / 32));
    constexpr int NI = BITS_BC / 32 / 4 / BITS_TBS;   // 4 x 16 B per thread
    uint4 t[NI];
#pragma unroll
    for (int k = 0; k < NI; k++) t[k] = src[tid + k * BITS_TBS];
    uint32_t nz = 0u;
#pragma unroll
    for (int k = 0; k < NI; k++) {
      reinterpret_cast<uint4 *>(xs)[tid + k * BITS_TBS] = t[k];
      nz |= t[k].x | t[k].y | t[k].z | t[k].w;
    }
    for (int i = tid; i < BITS_BR / 32; i += BITS_TBS) os[i] = 0u;
    if (tid <= it.sub1 - it.sub0) offs[tid] = bsub[it.soff + tid] >> 2;
    // No x bit set in this column block (the first BFS iterations: a handful of frontier vertices): every product is
    // 0, the item's partial bitmap is all zero and its entries need not be read at all.
    if (!__syncthreads_or(nz != 0u)) {
      uint4 *dst = reinterpret_cast<uint4 *>(partial + (size_t)blockIdx.x * (BITS_BR / 32));
      for (int i = tid; i < BITS_BR / 32 / 4; i += BITS_TBS) dst[i] = make_uint4(0u, 0u, 0u, 0u);
      return;
    }
  }
  // The item's entries are one 32-byte aligned stream (every sub-range is padded to a multiple of 8 entries by
  // repeating its last entry: OR does not mind).  A lane takes 8 CONSECUTIVE entries (two 16-byte loads): they are
  // ordered by row, so their hits fall into one or two 32-bit words of the result bitmap and are merged in registers
  // into at most two LDS atomics per 8 entries -- an LDS atomic costs by the active lane, and at a dense frontier one
  // atomic per hit made the launch twice as long (R-MAT-23 BFS: 428 vs 205 us per iteration).  A lane's octets come
  // in ascending order, so the sub-range of an octet -- the upper bits of its rows -- is found by moving a pointer
  // along offs[].  Two batches of U octets in flight per lane.
  constexpr int U = 2;
  const uint4 *e4 = reinterpret_cast<const uint4 *>(ent);
  const int o1 = it.e >> 3, nsub = it.sub1 - it.sub0;   // octets
  int sub = 0;
  auto load = [&](int o, uint4 (&w)[2 * U]) {
#pragma unroll
    for (int k = 0; k < U; k++) {
      const int ok = min(o + k * BITS_TBS, o1 - 1);
      w[2 * k] = e4[2 * ok];
      w[2 * k + 1] = e4[2 * ok + 1];
    }
  };
  auto consume = [&](int o, const uint4 (&w)[2 * U]) {
#pragma unroll
    for (int k = 0; k < U; k++) {
      const int ok = o + k * BITS_TBS;
      if (ok < o1) {
        while (sub + 1 < nsub && 2 * ok >= offs[sub + 1]) sub++;
        const uint32_t rbase = (uint32_t)(it.sub0 + sub) * BITS_SUB;
        const uint32_t ww[8] = {w[2 * k].x, w[2 * k].y, w[2 * k].z, w[2 * k].w, w[2 * k + 1].x, w[2 * k + 1].y, w[2 * k + 1].z, w[2 * k + 1].w};
        const uint32_t wb = (rbase + (ww[0] >> 19)) >> 5;   // word of the first entry's row
        uint32_t m0 = 0u, m1 = 0u;
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const uint32_t c = ww[i] & BITS_COL_MASK;
          if ((xs[c >> 5] >> (c & 31)) & 1u) {
            const uint32_t r = rbase + (ww[i] >> 19), d = (r >> 5) - wb;
            if (d == 0u) m0 |= 1u << (r & 31);
            else if (d == 1u) m1 |= 1u << (r & 31);
            else atomicOr(&os[r >> 5], 1u << (r & 31));   // (rows further apart: rare)
          }
        }
        if (m0) atomicOr(&os[wb], m0);
        if (m1) atomicOr(&os[wb + 1], m1);
      }
    }
  };
  uint4 wa[2 * U], wb2[2 * U];
  int o = (it.s >> 3) + tid;
  if (o < o1) {
    load(o, wa);
    for (;;) {
      load(o + U * BITS_TBS, wb2);
      consume(o, wa);
      o += U * BITS_TBS;
      if (o >= o1) break;
      load(o + U * BITS_TBS, wa);
      consume(o, wb2);
      o += U * BITS_TBS;
      if (o >= o1) break;
    }
  }
  __syncthreads();
  uint4 *dst = reinterpret_cast<uint4 *>(partial + (size_t)blockIdx.x * (BITS_BR / 32));
  for (int i = tid; i < BITS_BR / 32 / 4; i += BITS_TBS) dst[i] = reinterpret_cast<const uint4 *>(os)[i];
}

// Rows: OR of the partial bitmaps of the row range's items, then the epilogue.  A wave takes 64 words = 2048
// consecutive rows of one row range: lane L ORs word L of every item's partial bitmap (coalesced 256-byte loads),
// then the wave walks its rows 64 at a time -- the two words of a step come by readlane, row r is bit r & 31 of
// word r >> 5 -- so that y / the previous vector / out are accessed 256 bytes at a time.
constexpr int BITS_FIN_ROWS = 2048 * 4;   // rows per 256-thread workgroup
static_assert(BITS_BR % BITS_FIN_ROWS == 0, "a workgroup's rows lie in one row range");
static __global__ __launch_bounds__(256) void bits_finish(const int32_t *__restrict__ rr_item0, const uint32_t *__restrict__ partial,
                                                   int32_t rows, const uint32_t *__restrict__ y, int32_t alpha, int32_t beta,
                                                   int use_y_i, uint32_t *__restrict__ out, StepDev st) {
  if (gate_closed(st)) return;
  const int lane = threadIdx.x & 63;
  const int64_t row0 = (int64_t)blockIdx.x * BITS_FIN_ROWS + (int64_t)(threadIdx.x >> 6) * 2048;   // the wave's first row
  if (row0 >= rows) return;
  const int rr = (int)(row0 / BITS_BR);
  const uint32_t word0 = (uint32_t)(row0 % BITS_BR) >> 5;
  uint32_t w = 0u;
  const int i0 = rr_item0[rr], i1 = rr_item0[rr + 1];
  for (int it = i0; it < i1; it++)
    w |= partial[(size_t)it * (BITS_BR / 32) + word0 + lane];
  const bool use_y = use_y_i != 0;
#pragma unroll 4
  for (int j = 0; j < 32; j++) {
    const uint32_t wa = (uint32_t)__builtin_amdgcn_readlane((int)w, 2 * j), wb = (uint32_t)__builtin_amdgcn_readlane((int)w, 2 * j + 1);
    const int64_t r = row0 + 64 * j + lane;
    const int32_t dot = (int32_t)(((lane < 32 ? wa : wb) >> (lane & 31)) & 1u);
    if (r < rows)
      finish_row<OrAndI32>((int32_t)r, dot, y, alpha, beta, use_y, out, st);
  }
}

} // namespace sh
