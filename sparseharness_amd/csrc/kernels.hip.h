// kernels.hip.h -- CDNA4 (gfx950) CSR SpMV kernels, templated on the semiring.
//
// What they compute is the per-row contract of every Lift strategy in the
// reference (example/<algo>/kernel*.json:3, inventory in SURVEY.md 2.2):
//   out[r] = epilogue( (+)_j ( x_or_identity(col_j) (x) val_j ), alpha, y[r], beta )
// How they compute it is native.  Two execution plans (chosen per matrix at
// upload, engine.hip):
//   A  "stream": x small enough for the per-XCD L2 (or a matrix with local
//      columns): the matrix is streamed once, in CSR order, with
//      16-byte-per-lane coalesced loads and x is gathered from global memory;
//      no padded ELLPACK, no global temp buffer, no re-reads.  Described next.
//   B  "x-tiled two-phase": big x with scattered columns: x tiles are staged in
//      LDS and the products re-binned through HBM.  Described further down.
//   (C, for SH_OR_AND_I32 launches only: the bit-blocked layout of bits.hip.h -- x as a bitmap, 4 bytes per entry.)
//
// Plan A schedule (built on upload): the row range is cut into
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
#include <cstddef>
#include <type_traits>
#include "semiring.hip.h"

namespace sh {

// SH_STATS builds (tools/, never the product): timelines of the tiled plan's launches
#ifdef SH_STATS
#define SH_STAT(...) __VA_ARGS__
#else
#define SH_STAT(...)
#endif

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
constexpr int MAX_PIECES = 8;
struct StepDev {
  int32_t *changed;         // nullptr: no test
  const uint32_t *prev;     // previous vector; row r compares prev[prev_off + r]
  int64_t prev_off;
  double delta;
  // Launches of an iteration loop are enqueued several iterations ahead of the host (sh_iterate): a launch
  // whose gate word is 0 -- the previous iteration changed nothing, the loop is over -- returns at once.
  const int32_t *gate = nullptr;
  // Row pieces (multi-GPU iteration driver, sh_spmv_step_pieces): the matrix' rows live in n_pieces runs of piece_rows
  // rows; row r of piece c = r / piece_rows is element r + piece_delta[c] of out, y and prev (the vectors interleave
  // the pieces of all ranks so that one piece of every rank is one contiguous region to all-gather).  The geometry
  // lives in device memory (PieceDev, one per matrix, rewritten only when it changes), not in the kernel arguments:
  // as 30 more preloaded SGPRs it made spmv_tiled_phase2s spill scalar registers.  When a
  // workgroup has written its last row of piece c it adds 1 to done[c] (after a release at system scope): the
  // host, polling, then starts the exchange of that piece while the later ones are still being computed.
  const struct PieceDev *pcs = nullptr;   // nullptr: rows are elements 0..rows-1, nothing is reported
  uint32_t expected = 0;          // arrivals per piece and launch (= workgroups of the reporting launch); 0: no reporting
  uint32_t round = 0;             // this launch's number (sh_spmv_step_pieces counts them per matrix): what a completed piece reports
};
struct PieceDev {
  int32_t n_pieces;
  int32_t piece_rows;
  int64_t piece_delta[MAX_PIECES];
  int32_t piece_bin_end[MAX_PIECES];   // tiled plan: piece c is complete once every row bin below this index is
  uint32_t *done;                      // [n_pieces] arrival counters of the launch in flight (device memory; the arrival that completes a piece resets its counter)
  uint32_t *done_host;                 // [n_pieces] words in host memory: the `expected`-th arrival at done[c] writes the launch's round there
};
__device__ __forceinline__ bool reports(const StepDev &st) { return st.expected != 0u; }
__device__ __forceinline__ int64_t row_element(const StepDev &st, int32_t row) {
  const PieceDev *pc = st.pcs;
  return pc ? (int64_t)row + pc->piece_delta[min(row / pc->piece_rows, pc->n_pieces - 1)] : (int64_t)row;
}
__device__ __forceinline__ bool gate_closed(const StepDev &st) { return st.gate != nullptr && *st.gate == 0; }

struct LongSeg { int32_t row, s, e, slot; };
struct LongRow { int32_t row, slot0, nslots, pad; };

typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));
typedef uint32_t v2u32 __attribute__((ext_vector_type(2)));
// (tools builds: cache policy bits of phase 2's P / slot loads and of phase 1's P stores, -DSH_P2_P_POL / SH_P2_S_POL / SH_P1_ST_POL =
// 1 nt, 2 sc1, 3 sc0 sc1, 4 sc0)
#define SH_POL_STR_(n) SH_POL_STR_##n
#define SH_POL_STR(n) SH_POL_STR_(n)
#define SH_POL_STR_0 ""
#define SH_POL_STR_1 " nt"
#define SH_POL_STR_2 " sc1"
#define SH_POL_STR_3 " sc0 sc1"
#define SH_POL_STR_4 " sc0"
#ifndef SH_P2_P_POL
#define SH_P2_P_POL 0
#endif
#ifndef SH_P2_S_POL
#define SH_P2_S_POL 0
#endif
#ifndef SH_P1_ST_POL
#define SH_P1_ST_POL 0
#endif
// read-once stream words (tools builds, -DSH_P1_NT: a non-temporal load, so that the stream does not push the x tiles out of L2)
template <class W>
__device__ __forceinline__ W stream_load(const W *p) {
#ifdef SH_P1_NT
  W w;
  if constexpr (sizeof(W) == 16) { const v4u32 t = __builtin_nontemporal_load(reinterpret_cast<const v4u32 *>(p)); __builtin_memcpy(&w, &t, 16); }
  else if constexpr (sizeof(W) == 8) { const v2u32 t = __builtin_nontemporal_load(reinterpret_cast<const v2u32 *>(p)); __builtin_memcpy(&w, &t, 8); }
  else if constexpr (sizeof(W) == 4) { const uint32_t t = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(p)); __builtin_memcpy(&w, &t, 4); }
  else { const uint16_t t = __builtin_nontemporal_load(reinterpret_cast<const uint16_t *>(p)); __builtin_memcpy(&w, &t, 2); }
  return w;
#else
  return *p;
#endif
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains every
// outstanding global load (s_waitcnt vmcnt(0)), which would serialise the register prefetch
// of the next bin behind the current bin's LDS work; the kernels below exchange data between
// waves through LDS exclusively, so waiting on lgkmcnt is sufficient.
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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
  const int64_t at = row_element(st, row);
  T yv = use_y ? from_bits<T>(y[at]) : SR::identity();
  T o = SR::epilogue(dot, alpha, yv, beta, use_y);
  // A launch that reports its pieces (StepDev::done) writes its rows THROUGH to memory: once the storing wave has
  // drained (s_waitcnt vmcnt(0)) they are visible system-wide and a report needs no write-back of the XCD's L2.
  // (Measured, R-MAT-23 SSSP / BFS iteration on one GPU, profiles/r03_piece_reporting_cost.log: plain stores + one
  // asynchronous buffer_wbl2 per report and workgroup +5 % at 4 pieces, +11 % at 8; write-through rows +2 % / +0 %.)
  if (reports(st)) __hip_atomic_store(out + at, to_bits<T>(o), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  else out[at] = to_bits<T>(o);
  if (st.changed) {
    T in = from_bits<T>(st.prev[st.prev_off + at]);
    if (SR::differs(in, o, st.delta))
      *st.changed = 1;   // benign race: every writer stores 1 (a reporting launch publishes the word with its last piece)
  }
}

// The same with the y / previous-vector words of the row already in registers (requested before the row was summed:
// the staged pass of the tiled plan's phase 2).
template <class SR>
__device__ inline void finish_row_loaded(int32_t row, typename SR::T dot, uint32_t y_bits, uint32_t prev_bits,
                                         typename SR::T alpha, typename SR::T beta, bool use_y,
                                         uint32_t *__restrict__ out, const StepDev &st) {
  using T = typename SR::T;
  const int64_t at = row_element(st, row);
  T yv = use_y ? from_bits<T>(y_bits) : SR::identity();
  T o = SR::epilogue(dot, alpha, yv, beta, use_y);
  if (reports(st)) __hip_atomic_store(out + at, to_bits<T>(o), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  else out[at] = to_bits<T>(o);
  if (st.changed && SR::differs(from_bits<T>(prev_bits), o, st.delta))
    *st.changed = 1;
}

// One workgroup reports "my rows of pieces [c0, c1) are written".  Called by ONE lane after every storing wave of
// the workgroup has drained its (write-through) row stores and the workgroup has met at a barrier: one arrival per
// piece; the arrival that completes a launch's round tells the host, which polls done_host[c].
// The word the host polls carries the launch's OWN round number and the counter starts every launch at zero (the
// completing arrival resets it; launches on one matrix are ordered by their stream), so launches with different
// arrival counts -- the tiled plan's workgroups, the single arrival behind the bit-blocked or CSR-stream kernels --
// can alternate on one matrix without the word running ahead of the rounds.
__device__ inline void pieces_arrive(const StepDev &st, int c0, int c1) {
  uint32_t *done = st.pcs->done, *done_host = st.pcs->done_host;
  for (int c = c0; c < c1; c++) {
    const uint32_t n = __hip_atomic_fetch_add(done + c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    if (n == st.expected) {
      __hip_atomic_store(done + c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(done_host + c, st.round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// After launches that do not report by themselves (CSR-stream plan, a matrix of heavy rows only): one more tiny
// launch on the same stream -- every row is written by then -- completes the round of every piece.
static __global__ void report_all_pieces(StepDev st) {
  // (launched behind the kernels that wrote the rows: a kernel boundary, their write-through stores are complete)
  if (threadIdx.x == 0 && blockIdx.x == 0)
    for (int c = 0; c < st.pcs->n_pieces; c++)
      __hip_atomic_store(st.pcs->done_host + c, st.round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}


// Row reduction out of LDS, shared by both plans.  rp[0..nr] are the rows'
// offsets into prod[].  The number of lanes that cooperate on a row depends
// only on that row's length, so one long row inside a block of short ones
// never serialises the workgroup and a row's summation order is the same
// under every plan:
//   len <= 40    1 lane, sequential in stored order
//   len <= 256   8 lanes  (stride-8 partial sums, then a DPP tree into the group's last lane)
//   len <= 4096  64 lanes (stride-64 partial sums, then a DPP tree into lane 63)
//   longer       the whole workgroup (stride-NT sums, wave trees, then waves in order)
// Rows are classified by one pass that finishes the short ones on the spot
// and appends the others to three LDS work lists.
#ifndef SH_RL_SHORT
#define SH_RL_SHORT 40   // (24 until the 8-lane rows were handed out on demand: 189.6 -> 186 us for phase 2 at 40, the same at 48; profiles/r04_ab_loader_addresses_rowptr_dynamic_rows.log)
#endif
constexpr int RL_SHORT = SH_RL_SHORT, RL_MID = 256, RL_WAVE = 4096;
#ifndef SH_RL_BATCH
#define SH_RL_BATCH 8
#endif
constexpr int RL_BATCH = SH_RL_BATCH;          // products of a one-lane row read per LDS round trip
constexpr int RL_BATCH8 = 4, RL_BATCH64 = 4;   // the same for the lanes of an 8-lane / a 64-lane row
// Spare words behind a product array: a batch starts inside its row and reads on past the row's end with immediate
// offsets from ONE address register (no clamp per read: three instructions per product less) -- up to 7 words for a
// one-lane row, 8 * (RL_BATCH8 - 1) + 7 for an 8-lane row, 64 * (RL_BATCH64 - 1) + 63 for a 64-lane row -- so the
// reads of the image's last row may run this far past the image.  Word 0 of them is where phase 2's loaders drop
// padding products.
constexpr int RL_PAD = 64 * RL_BATCH64;
// rp[] entries may carry RP_SKIP: the row is produced elsewhere (heavy rows of the tiled plan)
// and must be neither written nor tested here.  Offsets stay below 2^30.
constexpr int32_t RP_SKIP = 1 << 30, RP_MASK = RP_SKIP - 1;

// Cross-lane sums of the cooperative rows in DPP (no LDS round trip per step, unlike __shfl_xor = ds_bpermute): the
// partial sums of a group of 8 lanes end up in the group's LAST lane (row_shr 4, 2, 1 inside a row of 16 lanes), those
// of a wave in lane 63 (row_shr 8, 4, 2, 1, then row_bcast:15 / :31).  Lanes whose source lies outside the DPP row
// receive the identity.  The order of the additions depends on the lane positions alone.
template <class SR, int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ typename SR::T dpp_fold(typename SR::T t) {
  using T = typename SR::T;
  const T up = from_bits<T>((uint32_t)__builtin_amdgcn_update_dpp((int)to_bits<T>(SR::identity()), (int)to_bits<T>(t), CTRL, ROW_MASK, 0xF, false));
  return SR::add(t, up);
}
template <class SR>
__device__ __forceinline__ typename SR::T sum_to_last_of_8(typename SR::T t) {
  t = dpp_fold<SR, 0x114>(t);
  t = dpp_fold<SR, 0x112>(t);
  return dpp_fold<SR, 0x111>(t);
}
template <class SR>
__device__ __forceinline__ typename SR::T sum_to_lane_63(typename SR::T t) {
  t = dpp_fold<SR, 0x118>(t);
  t = dpp_fold<SR, 0x114>(t);
  t = dpp_fold<SR, 0x112>(t);
  t = dpp_fold<SR, 0x111>(t);
  t = dpp_fold<SR, 0x142, 0xA>(t);      // row_bcast:15 into rows 1 and 3
  return dpp_fold<SR, 0x143, 0xC>(t);   // row_bcast:31 into rows 2 and 3
}

// (se8 / se64: the listed row's product range, first | one-past-last << 16, noted by the thread that classified the row
// -- it had both offsets in registers -- so that a cooperative trip is two dependent LDS round trips, list entry and
// products, not three: the cooperative rows are what phase 2's reducers spend the second half of a bin on, a quarter
// longer than the loaders' half)
template <int NT, int NNZ_CAP> struct ReduceScratch {
  uint16_t lst8[NNZ_CAP / (RL_SHORT + 1) + 1];
  uint16_t lst64[NNZ_CAP / (RL_MID + 1) + 1];
  uint32_t se8[NNZ_CAP / (RL_SHORT + 1) + 1];
  uint32_t se64[NNZ_CAP / (RL_MID + 1) + 1];
  uint16_t lstB[NNZ_CAP / (RL_WAVE + 1) + 1];
  int32_t cnt[8];   // two sets of {n8, n64, nB, next 8-lane row to hand out}: the wave-specialised phase 2 alternates them
  uint32_t wred[NT / 64];
};

// `tid` in [0, NT) numbers the threads that take part (all of them hit the barriers inside);
// `cnt` are the three list counters, zeroed before the barrier that published prod[].
// RAW (phase 2 of the tiled plan): rp[] holds the light row offsets as they lie in device memory -- bit 31 = the row is
// produced elsewhere, the rest an offset that `raw_base` turns into an offset into prod[] -- because the words are put
// there by the loader waves, 16 bytes at a time, untouched (see tiled_phase2_run).
template <class SR, int NT, int NNZ_CAP, bool RAW = false>
__device__ inline void reduce_rows_from_lds(const uint32_t *prod, const int32_t *rp, int nr, int r0,
                                            ReduceScratch<NT, NNZ_CAP> &sc, int32_t *cnt, const int tid,
                                            const uint32_t *__restrict__ y,
                                            typename SR::T alpha, typename SR::T beta, bool use_y,
                                            uint32_t *__restrict__ out, const StepDev &st,
                                            uint32_t *stage = nullptr, uint64_t *prof = nullptr, uint32_t raw_base = 0u) {
  using T = typename SR::T;
  auto offset_word = [&](int row) -> uint32_t {   // the row's offset in the encoding of the comment above RP_SKIP
    const uint32_t w = (uint32_t)rp[row];
    if (!RAW) return w;
    return ((w & 0x7FFFFFFFu) - raw_base) | ((w >> 31) ? (uint32_t)RP_SKIP : 0u);
  };
  // stage != nullptr (workgroup-uniform): the row's dot goes to stage[row] in LDS and the caller
  // applies the epilogue later in a coalesced pass (used when the epilogue reads y / prev: those
  // loads would otherwise sit, one row at a time, in the middle of the reduction)
  auto emit = [&](int row, T acc) {
    if (stage)
      stage[row] = to_bits<T>(acc);
    else
      finish_row<SR>(r0 + row, acc, y, alpha, beta, use_y, out, st);
  };
  // The sums below read their products a batch at a time -- all reads of a batch are issued before the first is
  // used -- instead of one LDS round trip per product: a wave's time in a dependent read / add loop is the LDS
  // latency times its longest row (phase 2 of the tiled plan spent 78 % of its cycles in these loops:
  // profiles/r04_phase2_role_profile.log).  What a batch reads past the row's end is not added; the order of the
  // additions is the stored order, as before.  (prof: SH_STATS builds, cycles of wave 0 in {MID barrier, one-lane
  // pass, 8-lane rows, 64-lane rows}.)
  SH_STAT(const uint64_t pf_r0 = __builtin_amdgcn_s_memtime();)
  // Every row once: its offsets, its class, and the one-lane rows summed on the spot -- R1 rows per thread and trip, so
  // that the offsets of both and then the first batches of both share one LDS round trip each.
  constexpr int R1 = 1;   // (two rows per thread and trip measured slower, 200.6 vs 194.6 us same box: profiles/r04_phase2_role_profile.log)
  for (int row0 = tid; row0 < nr; row0 += R1 * NT) {
    uint32_t a[R1], b[R1];
#pragma unroll
    for (int k = 0; k < R1; k++) {
      const int row = min(row0 + k * NT, nr);   // (rp[nr] is the last offset: a row past the end comes out empty)
      a[k] = offset_word(row);
      b[k] = offset_word(min(row + 1, nr));
    }
    int s[R1], len1[R1];
    bool one[R1];
    T acc[R1];
#pragma unroll
    for (int k = 0; k < R1; k++) {
      const int row = row0 + k * NT;
      const int len = (int)(b[k] & RP_MASK) - (int)(a[k] & RP_MASK);
      const bool mine = row < nr && !(a[k] & RP_SKIP);
      s[k] = (int)(a[k] & RP_MASK);
      one[k] = mine && len <= RL_SHORT;
      len1[k] = one[k] ? len : 0;
      acc[k] = SR::identity();
      if (mine && len > RL_SHORT) {
        const uint32_t se = (uint32_t)s[k] | ((uint32_t)(s[k] + len) << 16);   // (offsets stay below 2^15 where this is read: phase 2's bins; the CSR-stream plan's blocks hold 4096 products)
        if (len <= RL_MID) { const int at = atomicAdd(&cnt[0], 1); sc.lst8[at] = (uint16_t)row; sc.se8[at] = se; }
        else if (len <= RL_WAVE) { const int at = atomicAdd(&cnt[1], 1); sc.lst64[at] = (uint16_t)row; sc.se64[at] = se; }
        else sc.lstB[atomicAdd(&cnt[2], 1)] = (uint16_t)row;
      }
    }
    for (int j0 = 0; j0 < RL_SHORT; j0 += RL_BATCH) {
      bool more = false;
#pragma unroll
      for (int k = 0; k < R1; k++) more = more || j0 < len1[k];
      if (!more) break;
      uint32_t v[R1][RL_BATCH];
#pragma unroll
      for (int k = 0; k < R1; k++)
        if (j0 < len1[k]) {
#pragma unroll
          for (int i = 0; i < RL_BATCH; i++)
            v[k][i] = prod[s[k] + j0 + i];   // (unclamped, immediate offsets from one address: see RL_PAD)
        }
#pragma unroll
      for (int k = 0; k < R1; k++)
        if (j0 < len1[k]) {
#pragma unroll
          for (int i = 0; i < RL_BATCH; i++)
            if (j0 + i < len1[k])
              acc[k] = SR::add(acc[k], from_bits<T>(v[k][i]));
        }
    }
#pragma unroll
    for (int k = 0; k < R1; k++)
      if (one[k])
        emit(row0 + k * NT, acc[k]);
  }
  SH_STAT(const uint64_t pf_m0 = __builtin_amdgcn_s_memtime(); if (prof) prof[1] += pf_m0 - pf_r0;)
  lds_barrier();
  SH_STAT(const uint64_t pf_m1 = __builtin_amdgcn_s_memtime(); if (prof) prof[0] += pf_m1 - pf_m0;)
  const int n8 = cnt[0], n64 = cnt[1], nB = cnt[2];
  // The 64-lane rows first, one per wave in list order; then the 8-lane rows, eight at a time to whichever wave asks next
  // (cnt[3]: an LDS counter, zero at the barrier above) -- a wave that has just summed a row of a thousand products takes
  // fewer of them.  Dealt out by position the waves with a long row were what a bin waited for: wave 0 spent 3.9 K cycles
  // of a bin on its 8-lane rows and 2.7 K on its 64-lane row, seven of the twelve waves had no long row at all.  Which
  // wave sums a row does not change the sum.
  for (int idx = tid >> 6; idx < n64; idx += NT / 64) {
    const int row = sc.lst64[idx], l = tid & 63;
    const uint32_t se = sc.se64[idx];
    const int e = (int)(se >> 16);
    T acc = SR::identity();
    for (int j = (int)(se & 0xFFFFu) + l; j < e; j += 64 * RL_BATCH64) {
      uint32_t v[RL_BATCH64];
#pragma unroll
      for (int k = 0; k < RL_BATCH64; k++)
        v[k] = prod[j + 64 * k];
#pragma unroll
      for (int k = 0; k < RL_BATCH64; k++)
        if (j + 64 * k < e)
          acc = SR::add(acc, from_bits<T>(v[k]));
    }
    acc = sum_to_lane_63<SR>(acc);
    if (l == 63)
      emit(row, acc);
  }
  SH_STAT(const uint64_t pf_m2 = __builtin_amdgcn_s_memtime(); if (prof) prof[3] += pf_m2 - pf_m1;)
  for (;;) {
    int first = 0;
    if ((tid & 63) == 0)
      first = atomicAdd(&cnt[3], 8);
    first = __builtin_amdgcn_readfirstlane(first);   // (lane 0 of the wave is always active here)
    if (first >= n8)
      break;
    const int idx = first + ((tid & 63) >> 3), l = tid & 7;
    if (idx < n8) {
      const int row = sc.lst8[idx];
      const uint32_t se = sc.se8[idx];
      const int e = (int)(se >> 16);
      T acc = SR::identity();
      for (int j = (int)(se & 0xFFFFu) + l; j < e; j += 8 * RL_BATCH8) {
        uint32_t v[RL_BATCH8];
#pragma unroll
        for (int k = 0; k < RL_BATCH8; k++)
          v[k] = prod[j + 8 * k];   // (unclamped: see RL_PAD)
#pragma unroll
        for (int k = 0; k < RL_BATCH8; k++)
          if (j + 8 * k < e)
            acc = SR::add(acc, from_bits<T>(v[k]));
      }
      acc = sum_to_last_of_8<SR>(acc);
      if (l == 7)
        emit(row, acc);
    }
  }
  SH_STAT(if (prof) prof[2] += __builtin_amdgcn_s_memtime() - pf_m2;)
  for (int idx = 0; idx < nB; idx++) {   // nB is workgroup-uniform
    const int row = sc.lstB[idx];
    const int e = (int)(offset_word(row + 1) & RP_MASK);
    T acc = SR::identity();
    for (int j = (int)(offset_word(row) & RP_MASK) + tid; j < e; j += NT)
      acc = SR::add(acc, from_bits<T>(prod[j]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
      acc = SR::add(acc, from_bits<T>(__shfl_xor(to_bits<T>(acc), o, 64)));
    if ((tid & 63) == 0)
      sc.wred[tid >> 6] = to_bits<T>(acc);
    lds_barrier();
    if (tid == 0) {
      T t = from_bits<T>(sc.wred[0]);
      for (int w = 1; w < NT / 64; w++)
        t = SR::add(t, from_bits<T>(sc.wred[w]));
      emit(row, t);
    }
    lds_barrier();
  }
}

template <class SR>
__global__ __launch_bounds__(BS) void spmv_csr_kernel(
    CsrDev A, const uint32_t *__restrict__ x, const uint32_t *__restrict__ y,
    typename SR::T alpha, typename SR::T beta, int use_y_i, uint32_t *__restrict__ out,
    const int32_t *__restrict__ blk_row, int32_t n_stream,
    const LongSeg *__restrict__ segs, uint32_t *__restrict__ partial, StepDev st) {
  using T = typename SR::T;
  if (gate_closed(st))
    return;
  __shared__ uint32_t prod[NNZ_BLK + RL_PAD];
  __shared__ int32_t rp[ROWS_BLK + 1];
  __shared__ ReduceScratch<BS, NNZ_BLK> sc;
  uint32_t *wred = sc.wred;
  const int tid = threadIdx.x;
  const bool use_y = use_y_i != 0;
  const int b = blockIdx.x;
  if (tid < 4)
    sc.cnt[tid] = 0;

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
    // Phase 2: per-row reduction out of LDS (rp[] made relative to prod[]).
    for (int i = tid; i <= nr; i += BS)
      rp[i] -= base;
    __syncthreads();
    reduce_rows_from_lds<SR, BS, NNZ_BLK>(prod, rp, nr, r0, sc, sc.cnt, tid, y, alpha, beta, use_y, out, st);
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
  if (i >= n_long || gate_closed(st))
    return;
  const LongRow lr = rows[i];
  T acc = SR::identity();
  for (int k = 0; k < lr.nslots; k++)
    acc = SR::add(acc, from_bits<T>(partial[lr.slot0 + k]));
  finish_row<SR>(lr.row, acc, y, alpha, beta, use_y_i != 0, out, st);
}



// Heavy rows of the tiled plan: phase 1 leaves one partial per (row, tile, wave-part); one
// 256-thread block per row adds them (stride-256 sums in slot order, wave xor-trees, then the
// four waves in order) and applies the epilogue.
constexpr int HFIX_BS = 256;
template <class SR>
__global__ __launch_bounds__(HFIX_BS) void spmv_heavy_fixup(
    const LongRow *__restrict__ rows, const uint32_t *__restrict__ partial,
    const uint32_t *__restrict__ y, typename SR::T alpha, typename SR::T beta, int use_y_i,
    uint32_t *__restrict__ out, StepDev st) {
  using T = typename SR::T;
  if (gate_closed(st))
    return;
  __shared__ uint32_t wred[HFIX_BS / 64];
  const LongRow lr = rows[blockIdx.x];
  const int tid = threadIdx.x;
  T acc = SR::identity();
  for (int k = tid; k < lr.nslots; k += HFIX_BS)
    acc = SR::add(acc, from_bits<T>(partial[lr.slot0 + k]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    acc = SR::add(acc, from_bits<T>(__shfl_xor(to_bits<T>(acc), o, 64)));
  if ((tid & 63) == 0)
    wred[tid >> 6] = to_bits<T>(acc);
  __syncthreads();
  if (tid == 0) {
    T t = from_bits<T>(wred[0]);
#pragma unroll
    for (int w = 1; w < HFIX_BS / 64; w++)
      t = SR::add(t, from_bits<T>(wred[w]));
    finish_row<SR>(lr.row, t, y, alpha, beta, use_y_i != 0, out, st);
  }
}

// The same sum by ONE wave (bit-identical: lane l plays threads l, l+64, l+128, l+192 of the block
// above).  Used by the wave-specialised phase 2, whose reducer waves are idle while the loaders
// fill the first bin.
template <class SR>
__device__ inline void heavy_row_by_wave(const LongRow lr, const uint32_t *__restrict__ partial, int lane,
                                         const uint32_t *__restrict__ y, typename SR::T alpha, typename SR::T beta,
                                         bool use_y, uint32_t *__restrict__ out, const StepDev &st) {
  using T = typename SR::T;
  T acc[HFIX_BS / 64];
#pragma unroll
  for (int w = 0; w < HFIX_BS / 64; w++)
    acc[w] = SR::identity();
  constexpr int HU = 4;   // chunks of HFIX_BS partials in flight per lane (a row can have thousands)
  for (int k0 = 0; k0 < lr.nslots; k0 += HFIX_BS * HU) {
    uint32_t v[HU][HFIX_BS / 64];
#pragma unroll
    for (int u = 0; u < HU; u++)
#pragma unroll
      for (int w = 0; w < HFIX_BS / 64; w++) {
        const int k = k0 + u * HFIX_BS + w * 64 + lane;
        // clamped: branch-free, all loads fly together
        v[u][w] = partial[lr.slot0 + min(k, lr.nslots - 1)];
      }
#pragma unroll
    for (int u = 0; u < HU; u++)
#pragma unroll
      for (int w = 0; w < HFIX_BS / 64; w++) {
        const int k = k0 + u * HFIX_BS + w * 64 + lane;
        if (k < lr.nslots)
          acc[w] = SR::add(acc[w], from_bits<T>(v[u][w]));
      }
  }
#pragma unroll
  for (int w = 0; w < HFIX_BS / 64; w++) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
      acc[w] = SR::add(acc[w], from_bits<T>(__shfl_xor(to_bits<T>(acc[w]), o, 64)));
  }
  if (lane == 0) {
    T t = acc[0];
#pragma unroll
    for (int w = 1; w < HFIX_BS / 64; w++)
      t = SR::add(t, acc[w]);
    finish_row<SR>(lr.row, t, y, alpha, beta, use_y, out, st);
  }
}

// ===========================================================================
// x-tiled two-phase plan (for matrices whose x does not fit the per-XCD L2).
//
// Measured on MI355X (profiles/r01_lab_gather_microbench.log): a random 4-byte
// gather costs the same whether the 40 MB x sits in Infinity Cache or HBM
// (~55-60 G gathers/s: every L2 miss moves a whole line) and ~230 G/s when x
// is L2-resident, while a gather out of LDS runs at the HBM streaming rate of
// its index stream (~1350 G/s).  So for big x no global gather is issued:
//
//   phase 1  (one workgroup per <= TCHUNK entries of one 32768-column tile)
//     the x tile (128 KiB) is staged in LDS; the tile's entries, stored
//     tile-major as {value word or one-byte value code, col u16}, are streamed,
//     multiplied against LDS, and
//       light rows: each aligned group of 4 products is stored with one 16-byte
//         store into the product array P at the entry's own stream position;
//       heavy rows (>= 8 entries per tile on average): summed per (row, tile)
//         inside the wave (segmented scan); one partial per run, no P traffic.
//   phase 2  (persistent, wave-specialised: spmv_tiled_phase2s)
//     per row bin of <= TBIN light products: loader waves gather the bin's
//     (bin, tile) pieces from P (piece tables: where each group of 4 lies) and scatter
//     them into an LDS image at their CSR slot (u16 per product); reducer waves
//     sum the rows out of the previous image with the same code as the stream
//     kernel above (deterministic order, same epilogue) and add up the heavy
//     rows' partials.
//
// HBM bytes per light entry: 3 (coded) or 6 read + 4 written in phase 1,
// 4 + 2 + ~0.14 read in phase 2; per heavy entry ~2.8 or ~6.3 read -- against the
// 8 B algorithmic, but all of it streamed.
// ===========================================================================
#ifndef SH_TCOLS
#define SH_TCOLS 32760
#endif
constexpr int TCOLS = SH_TCOLS;         // columns per x tile (4 B each in LDS); a multiple of 8, < 32768: a column code has 15 bits
constexpr int TBS = 1024;               // threads per phase-1 workgroup
#ifndef SH_TBIN
#define SH_TBIN 16384   // 32768: one 1024-thread WG per CU; 16384: two 512-thread WGs (measured 5 % faster)
#endif
constexpr int TBIN = SH_TBIN;           // products per row bin (LDS image: 4 B each)
constexpr int TBIN_ROWS = TBIN / 8;     // rows per bin (row_ptr slice in LDS)
#ifndef SH_TCHUNK
#define SH_TCHUNK 32768
#endif
#ifndef SH_P1_UNROLL
#define SH_P1_UNROLL 2
#endif
constexpr int TCHUNK = SH_TCHUNK;       // entries per phase-1 workgroup
constexpr int P1U = SH_P1_UNROLL;       // 16-byte groups in flight per thread in phase 1
constexpr uint16_t TCOL_IDENTITY = (uint16_t)TCOLS; // col16 code of "x reads as the identity": the LDS slot behind the tile holds it
// TCOL_FOLD: on the FIRST column code of a light group: the group's 4 products are folded into those of the group (lane) in
// front.  (Experimental builds with tiles wider than 15 bits have no room for the flag: they do not fold.)
constexpr uint16_t TCOL_FOLD = SH_TCOLS < 32768 ? 0x8000 : 0;
constexpr uint32_t TCOL_MASK = SH_TCOLS < 32768 ? 0x7FFFu : 0xFFFFu;
constexpr uint16_t TSLOT_PAD = 0xFFFF;     // slot16 marker: padding product
static_assert(TCOLS % 8 == 0 && TCOLS <= 65528 && TCOL_IDENTITY == TCOLS, "the identity column code indexes the slot behind the x tile; bit 15 is the fold flag when the tile leaves it free");
// Heavy rows: every (row, tile) piece is padded to whole STRIPS of HSTRIP consecutive stream entries.  One
// lane of phase 1 sums a strip (wide loads, 16 products in stream order); consecutive lanes whose strips
// belong to the same piece are then combined by a segmented wave scan in DPP, and the last lane of each run
// stores ONE partial per (row, tile, wave-part of 64 strips).  gdest[strip] = partial slot | (strips of the
// same partial to its left in the wave) << 25 | (last strip of its partial) << 31: the plan builder knows
// every split in advance, no key travels between lanes.  (History: the scan once ran per group of 4 entries
// through ds_bpermute and made the heavy chunks -- 31 % of the entries of the power-law matrix, a third of
// the bytes -- cost more than the light ones; one partial per strip without any scan left a 2.4 M-entry row
// with 150 K partials for a single wave of phase 2 to add up: 0.60 instead of 0.49 ms.)
constexpr int HSTRIP = 16;
constexpr int GD_DIST_SHIFT = 25;
constexpr uint32_t GD_SLOT_MASK = (1u << GD_DIST_SHIFT) - 1, GD_LAST = 0x80000000u;

// entries [s,e) of the stream (the tiles' light runs, then the tiles' heavy runs); positions >= hs
// belong to heavy rows.  Light chunks: ob0 = index in obase[] of the chunk's first block of 64 groups
// (obase[b] = P position of the block's first product).  Heavy chunks: pdelta = stream position of the
// first heavy entry (strip k of the heavy part covers entries [pdelta + 16 k, pdelta + 16 k + 16)).
struct TileChunk { int32_t tile, s, e, hs, pdelta, ob0, pad1, pad2; };
// r0/nr: rows of the bin; csr0: CSR position of its first entry; n: products incl. padding;
// pstart: where the bin's slots start in pslot[] (bin-major); gb0 / pt0: its first 64-group block in
// gblk[] / its first piece in ptab[].
// (field order: the second 16 bytes are all that phase 2's loader waves need of a bin -- they fetch three bins ahead,
// one s_load_dwordx4 each instead of the whole record: 12 scalar registers less in a kernel that ran out of them)
struct RowBin { int32_t r0, nr, csr0, pad0, gb0, n, pstart, pt0; };
struct RowBinL { int32_t gb0, n, pstart, pt0; };   // a bin as the loaders see it
static_assert(sizeof(RowBin) == 32 && offsetof(RowBin, gb0) == 16 && sizeof(RowBinL) == 16, "the loaders read the second half of a RowBin");

// Value coding (VC): when the matrix holds at most 256 distinct 4-byte values (always true for
// pattern files, and for every file once the reference's int narrowing -- quirk A-3 -- has been
// applied to small weights) the tile-major stream carries one-byte dictionary codes instead of the
// values: 3 instead of 6 bytes per entry read by phase 1.  Lossless: dict[code] is the original
// bit pattern; code 0 is always the all-zero word (padding).
#ifndef SH_P1_UNROLL_VC
#define SH_P1_UNROLL_VC 4
#endif
constexpr int P1U_VC = SH_P1_UNROLL_VC;
constexpr int VDICT = 256;      // one-byte (and four-bit) codes
constexpr int VDICT16 = 4096;   // two-byte codes: the dictionary (16 KiB) sits in LDS beside the x tile (VC = 3)

// Segmented inclusive scan over a wave: lane i ends up with (+) of t over lanes [i - dist, i]
// (dist = lanes of the same segment to the left).  Hillis-Steele inside each row of 16 lanes
// (row_shr DPP), then the row tails travel down by row_bcast:15 (rows 1, 3) and row_bcast:31
// (rows 2, 3).  The order of the additions is fixed by the lane position alone.
template <class SR>
__device__ __forceinline__ typename SR::T seg_scan_wave(typename SR::T t, const int dist, const int lane) {
  using T = typename SR::T;
  const int idb = (int)to_bits<T>(SR::identity());
  const int lr = lane & 15, dr = min(dist, lr);
#define SH_SEG_STEP(O)                                                                                              \
  {                                                                                                                 \
    const T up = from_bits<T>((uint32_t)__builtin_amdgcn_update_dpp(idb, (int)to_bits<T>(t), 0x110 + O, 0xF, 0xF, false)); \
    if (dr >= O) t = SR::add(t, up);                                                                                \
  }
  SH_SEG_STEP(1) SH_SEG_STEP(2) SH_SEG_STEP(4) SH_SEG_STEP(8)
#undef SH_SEG_STEP
  {
    const T up = from_bits<T>((uint32_t)__builtin_amdgcn_update_dpp(idb, (int)to_bits<T>(t), 0x142, 0xA, 0xF, false));   // row_bcast:15
    if ((lane & 16) && dist > lr) t = SR::add(t, up);
  }
  {
    const T up = from_bits<T>((uint32_t)__builtin_amdgcn_update_dpp(idb, (int)to_bits<T>(t), 0x143, 0xC, 0xF, false));   // row_bcast:31
    if (lane >= 32 && dist > (lane & 31)) t = SR::add(t, up);
  }
  return t;
}

// VC: 0 = raw 4-byte values, 1 = one-byte dictionary codes, 2 = four-bit codes (<= 16 values), 3 = two-byte codes
// (<= 4096 values: 4 instead of 6 bytes per entry for a matrix with a few hundred or thousand distinct weights, e.g.
// small integers after the reference's int narrowing, src/sparse_matrix.cpp:107, or 1 / degree weights)
// xs: [TCOLS + 4] words of LDS, ds: [VDICT] ([VDICT16] for VC = 3).
SH_STAT(__device__ uint64_t *g_p1_stats;)
// per workgroup of phase 2 (wave 0 of each role, shader cycles): [0] loader total, [1] loader in vmcnt waits, [2] loader in
// barriers, [3] bins, [4] reducer total, [5] reducer in barriers, [6] reducer in the reduction proper, [7] bins,
// [8..10] reducer in the one-lane pass / the 8-lane rows / the 64-lane rows
SH_STAT(__device__ uint64_t g_p2_prof[256 * 16];)
SH_STAT(__device__ uint64_t g_p2_wave[256 * 12 * 4];)   // per reducer wave: cycles at barriers, in the classifying pass, in the cooperative rows, in all
#ifdef SH_STATS
#define SH_TIMED(acc, ...) { const uint64_t pf_a = __builtin_amdgcn_s_memtime(); __VA_ARGS__; acc += __builtin_amdgcn_s_memtime() - pf_a; }
#else
#define SH_TIMED(acc, ...) { __VA_ARGS__; }
#endif
struct NoHook { __device__ void operator()() const {} };
// staged(): called by every thread right after the barrier that publishes the x tile (SH_STATS builds stamp it).
template <class SR, int VC, class Hook = NoHook>
__device__ __forceinline__ void tiled_phase1_chunk(
    const TileChunk ch, uint32_t *xs, uint32_t *ds, const void *__restrict__ tval_or_code,
    const uint32_t *__restrict__ vdict, const uint16_t *__restrict__ tcol,
    const uint32_t *__restrict__ gdest, const uint32_t *__restrict__ obase, const uint32_t *__restrict__ x, int32_t cols,
    uint32_t *__restrict__ P, uint32_t *__restrict__ partial, const bool skip_dead_tiles, uint32_t *__restrict__ tile_live,
    Hook staged = NoHook()) {
  using T = typename SR::T;
  constexpr int U = VC ? P1U_VC : P1U;
  // skip_dead_tiles (workgroup-uniform): when every x word of the tile is absorbing (SR::absorbing: an unreached vertex
  // of SSSP, a vertex outside the BFS frontier) all products of the chunk are the identity and are written without
  // reading the chunk's entries -- the first iterations of a search from one vertex touch a handful of tiles.
  const bool may_skip = SR::has_absorbing && skip_dead_tiles;
  // the value words of one group of 4 entries: 4 values, 4 one-byte codes, or 4 nibbles
  using VWord = typename std::conditional<VC == 0, uint4, typename std::conditional<VC == 1, uint32_t, typename std::conditional<VC == 2, uint16_t, uint2>::type>::type>::type;
  // xs[TCOLS] holds the identity: a column code of TCOL_IDENTITY (== TCOLS) reads it with no test
  const VWord *__restrict__ tval = reinterpret_cast<const VWord *>(tval_or_code);
  const uint2 *__restrict__ tcol2 = reinterpret_cast<const uint2 *>(tcol);
  const int tid = threadIdx.x;
  const int c0 = ch.tile * TCOLS;
  const uint32_t ident = to_bits<T>(SR::identity());
  if constexpr (VC == 3)
    reinterpret_cast<uint4 *>(ds)[tid] = reinterpret_cast<const uint4 *>(vdict)[tid];   // TBS * 4 == VDICT16 words
  else if (VC && tid < VDICT)
    ds[tid] = vdict[tid];
  if (tid == 0)
    xs[TCOLS] = ident;
  // Stage the x tile (cols is arbitrary, x is only guaranteed 4-byte aligned), then request the chunk's
  // first stream batch.  (LDS-DMA staging -- global_load_lds_dwordx4, no VGPR round trip -- was measured
  // neutral on the same box: 494-512 vs 494-497 us per SpMV.)  (Requesting it right behind the staging loads, so that its HBM latency runs
  // under the LDS writes, was measured SLOWER on the same box: 533 vs 508 us per SpMV.)
  auto stage = [&](auto request_first) -> bool {
    bool live_word = false;   // this thread staged a word that is not absorbing
    if (c0 + TCOLS <= cols && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
      // full tile, 16-byte aligned: 1 KiB per wave-instruction; every load is issued before the first
      // LDS write (a rolled loop would pay one memory latency per iteration)
      constexpr int NI = (TCOLS / 4 + TBS - 1) / TBS;
      uint4 t[NI];
#pragma unroll
      for (int k = 0; k < NI; k++)
        t[k] = reinterpret_cast<const uint4 *>(x + c0)[min(tid + k * TBS, TCOLS / 4 - 1)];
#pragma unroll
      for (int k = 0; k < NI; k++) {
        if (tid + k * TBS < TCOLS / 4)
          reinterpret_cast<uint4 *>(xs)[tid + k * TBS] = t[k];
        if (SR::has_absorbing)
          live_word = live_word || !SR::absorbing(t[k].x) || !SR::absorbing(t[k].y) || !SR::absorbing(t[k].z) || !SR::absorbing(t[k].w);
      }
    } else {
      for (int i = tid; i < TCOLS; i += TBS) {
        const uint32_t w = (c0 + i < cols) ? x[c0 + i] : ident;
        xs[i] = w;
        if (SR::has_absorbing) live_word = live_word || !SR::absorbing(w);
      }
    }
    bool live = true;
    if (may_skip) live = __syncthreads_or(live_word ? 1 : 0) != 0;
    else __syncthreads();
    // tile_live (launches whose phase 2 reads the pieces through tiled_mark_dead's table): every work item of the tile
    // says the same; a dead tile's products are then neither written here nor read there
    int t_here = tid;
    asm volatile("" : "+v"(t_here));   // (compared here: kept from the top of the kernel, the lane mask of tid == 0 spilled scalar registers)
    if (may_skip && tile_live != nullptr && t_here == 0) tile_live[ch.tile] = live ? 1u : 0u;
    staged();
    if (live) request_first();
    return live;
  };
  // the four products of one 16-byte group: x gathered from LDS, values decoded when coded
  auto products = [&](const VWord &w, const uint2 &c, T (&pr)[4]) {
    uint4 v;
    if constexpr (VC == 1)
      v = make_uint4(ds[w & 0xFFu], ds[(w >> 8) & 0xFFu], ds[(w >> 16) & 0xFFu], ds[w >> 24]);
    else if constexpr (VC == 2)
      v = make_uint4(ds[w & 0xFu], ds[(w >> 4) & 0xFu], ds[(w >> 8) & 0xFu], ds[(w >> 12) & 0xFu]);
    else if constexpr (VC == 3)
      v = make_uint4(ds[w.x & 0xFFFFu], ds[w.x >> 16], ds[w.y & 0xFFFFu], ds[w.y >> 16]);
    else
      v = w;
    // (bit 15 of a group's first column code is the fold flag)
    pr[0] = SR::mul(from_bits<T>(xs[c.x & TCOL_MASK]), from_bits<T>(v.x));
    pr[1] = SR::mul(from_bits<T>(xs[c.x >> 16]), from_bits<T>(v.y));
    pr[2] = SR::mul(from_bits<T>(xs[c.y & 0xFFFFu]), from_bits<T>(v.z));
    pr[3] = SR::mul(from_bits<T>(xs[c.y >> 16]), from_bits<T>(v.w));
  };
  // Both loops are software-pipelined: the loads of batch i+1 are issued before batch i is
  // consumed, so a wave always has one batch of loads in flight behind the stores it issues
  // (stores count in vmcnt on gfx9: without this every batch would wait out the previous
  // batch's write latency).  Loads are unconditional on clamped indices -- one basic block.
  // ---- light entries.  A lane owns a group of 4 consecutive stream entries and stores their 4 products with one
  // 16-byte store.  Two entries of one row that fall into this tile are a PAIR, laid out column-wise over the two
  // lanes of a lane pair (entry k of the odd lane B pairs with entry k of the even lane A; B's first column code
  // carries the FOLD flag): A adds B's four products to its own -- one DPP move each, no LDS -- and B stores nothing,
  // so a seventh of the light products of a power-law matrix never leave the register file.  Every lane thus stores
  // 0 or 4 products and the wave's products stay one contiguous, 16-byte aligned run of P: a storing lane's place is
  // the number of storing lanes below it (one ballot), the wave's place comes from obase[] (one scalar load per 64
  // groups; the plan builder counted the flags).
  // (Measured on the way here, same box, profiles/r03_ab_phase1_store_variants.log and r03_kernel_times_store_variants*:
  // folding runs of up to 4 entries INSIDE a lane leaves 1..4 products per lane; compacting them cost more than the
  // bytes saved -- four predicated dword stores per lane +93 us per SpMV, one 1..4-dword store per lane +55 us of
  // phase 1, through a wave-private LDS strip as 16-byte stores +32..36 us of phase 1 -- while phase 2 gained 37 us.)
  // Software-pipelined: the loads of batch i+1 are issued before batch i is consumed (ping-pong register sets A/B
  // instead of a copy at the loop end: a copy would wait for the loads it copies).
  // A chunk is either all light (hs == e) or all heavy (hs <= s): the plan builder cuts them apart.
  const int le = min(ch.e, max(ch.s, ch.hs)) / 4;
  if (ch.hs > ch.s) {
    constexpr int S = TBS * U;
    const int gs = ch.s / 4, last_blk = (le - gs - 1) >> 6, lane = tid & 63;
    auto load = [&](int gbase, VWord (&vw)[U], uint2 (&c)[U], uint32_t (&ob)[U]) {
#pragma unroll
      for (int k = 0; k < U; k++) {
        const int g = min(gbase + k * TBS, le - 1);
        vw[k] = stream_load(&tval[g]);
        c[k] = stream_load(&tcol2[g]);
        // the wave's 64 groups are one block of obase[] (chunks start on block boundaries): a scalar load
        ob[k] = obase[ch.ob0 + __builtin_amdgcn_readfirstlane(min((gbase + k * TBS - gs) >> 6, last_blk))];
      }
    };
    // (the loop runs per WAVE -- while the wave's first group is inside the chunk -- so that every lane takes part in
    // the DPP moves and ballots; lanes past the end of the chunk neither fold nor store)
    auto consume = [&](int gbase, const VWord (&vw)[U], const uint2 (&c)[U], const uint32_t (&ob)[U]) {
#pragma unroll
      for (int k = 0; k < U; k++) {
        const int g = gbase + k * TBS;
        if (g - lane < le) {   // wave-uniform
          const bool valid = g < le;
          T pr[4];
          products(vw[k], c[k], pr);
          const bool folds = valid && (c[k].x & TCOL_FOLD) != 0;          // this lane is the B of a pair
          const uint64_t fm = __ballot(folds), sm = __ballot(valid && !folds);
          const bool takes = ((fm >> 1) >> lane) & 1u;                   // the lane behind folds into this one
#pragma unroll
          for (int i = 0; i < 4; i++) {
            // quad_perm [1,1,3,3]: an even lane reads its odd neighbour
            const T up = from_bits<T>((uint32_t)__builtin_amdgcn_update_dpp(0, (int)to_bits<T>(pr[i]), 0xF5, 0xF, 0xF, false));
            if (takes) pr[i] = SR::add(pr[i], up);
          }
          if (valid && !folds) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(sm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sm, 0u));
#if SH_P1_ST_POL != 0
            {
              v4u32 pw = {to_bits<T>(pr[0]), to_bits<T>(pr[1]), to_bits<T>(pr[2]), to_bits<T>(pr[3])};
              asm volatile("global_store_dwordx4 %0, %1, off" SH_POL_STR(SH_P1_ST_POL) : : "v"(P + ob[k] + 4u * rank), "v"(pw) : "memory");
            }
#else
            *reinterpret_cast<uint4 *>(P + ob[k] + 4u * rank) =
                make_uint4(to_bits<T>(pr[0]), to_bits<T>(pr[1]), to_bits<T>(pr[2]), to_bits<T>(pr[3]));
#endif
          }
        }
      }
    };
    VWord va[U], vb[U];
    uint2 ca[U], cb[U];
    uint32_t oa[U], obb[U];
    int g0 = gs + tid;
    const bool live = stage([&]() { load(g0, va, ca, oa); });
    if (!live && tile_live != nullptr) {
      // a dead tile whose pieces phase 2 will not read: nothing to write
    } else if (!live) {
      // a dead tile: every block of 64 groups stores identity products where its own would have gone
      // (obase[b + 1] - obase[b] products: the table carries one entry behind the last block)
      const uint4 id4 = make_uint4(ident, ident, ident, ident);
      for (int blk = tid >> 6; blk <= last_blk; blk += TBS / 64) {
        const uint32_t p0 = obase[ch.ob0 + blk], p1 = obase[ch.ob0 + blk + 1];
        if (p0 + 4u * (uint32_t)lane < p1) *reinterpret_cast<uint4 *>(P + p0 + 4u * lane) = id4;
      }
    } else if (g0 - lane < le) {
      for (;;) {
        load(g0 + S, vb, cb, obb);
        consume(g0, va, ca, oa);
        g0 += S;
        if (g0 - lane >= le) break;
        load(g0 + S, va, ca, oa);
        consume(g0, vb, cb, obb);
        g0 += S;
        if (g0 - lane >= le) break;
      }
    }
  }
  // ---- heavy entries (rows averaging >= 8 entries per tile): their products never travel through P.
  // One lane per strip of HSTRIP = 16 consecutive entries of one (row, tile) piece: wide loads (32 B of
  // columns, 8 / 16 / 64 B of value codes / values, the strip's gdest word), 16 products summed in stream
  // order, then the segmented wave scan over the strips of a piece; the last lane of a run stores the
  // partial.  Wave boundaries are fixed by the stream position (chunks are cut at multiples of 64 strips
  // from the heavy run's start), so the builder knows every split.  Deterministic, no atomics.
  if (ch.hs <= ch.s) {
    using SWord = typename std::conditional<VC == 2, uint2, uint4>::type;
    constexpr int NV = VC == 0 ? 4 : (VC == 3 ? 2 : 1);   // SWords of values per strip: 4 x 16 B raw, 16 B of byte codes, 8 B of nibbles, 2 x 16 B of two-byte codes
    const SWord *__restrict__ sval = reinterpret_cast<const SWord *>(tval_or_code);
    const uint4 *__restrict__ scol = reinterpret_cast<const uint4 *>(tcol);
    const int s0 = ch.s / HSTRIP, s1 = ch.e / HSTRIP, sbase = ch.pdelta / HSTRIP;   // strips of the chunk; first heavy strip
    auto load = [&](int sid, SWord (&vw)[NV], uint4 (&c)[2], uint32_t &d) {
      const int q = min(sid, s1 - 1);
#pragma unroll
      for (int k = 0; k < NV; k++) vw[k] = stream_load(&sval[(size_t)q * NV + k]);
      c[0] = stream_load(&scol[(size_t)q * 2]);
      c[1] = stream_load(&scol[(size_t)q * 2 + 1]);
      d = gdest[q - sbase];
    };
    const int lane = tid & 63;
    auto consume = [&](int sid, const SWord (&vw)[NV], const uint4 (&c)[2], uint32_t d) {
      const bool valid = sid < s1;
      if (!__ballot(valid)) return;   // wave-uniform
      const uint32_t cw[8] = {c[0].x, c[0].y, c[0].z, c[0].w, c[1].x, c[1].y, c[1].z, c[1].w};
      T t = SR::identity();
#pragma unroll
      for (int i = 0; i < HSTRIP; i++) {
        uint32_t v;
        if constexpr (VC == 0) {
          const uint4 w = vw[i / 4];
          v = (i % 4 == 0) ? w.x : (i % 4 == 1) ? w.y : (i % 4 == 2) ? w.z : w.w;
        } else if constexpr (VC == 1) {
          const uint32_t w = (i / 4 == 0) ? vw[0].x : (i / 4 == 1) ? vw[0].y : (i / 4 == 2) ? vw[0].z : vw[0].w;
          v = ds[(w >> (8 * (i % 4))) & 0xFFu];
        } else if constexpr (VC == 3) {
          const uint4 q = vw[i / 8];
          const uint32_t w = (i % 8 / 2 == 0) ? q.x : (i % 8 / 2 == 1) ? q.y : (i % 8 / 2 == 2) ? q.z : q.w;
          v = ds[(w >> (16 * (i % 2))) & 0xFFFFu];
        } else {
          const uint32_t w = (i / 8 == 0) ? vw[0].x : vw[0].y;
          v = ds[(w >> (4 * (i % 8))) & 0xFu];
        }
        const uint32_t col = (cw[i / 2] >> (16 * (i % 2))) & 0xFFFFu;
        t = SR::add(t, SR::mul(from_bits<T>(xs[col]), from_bits<T>(v)));
      }
      if (!valid) t = SR::identity();   // (clamped reload of the chunk's last strip: must not join a run)
      t = seg_scan_wave<SR>(t, valid ? (int)((d >> GD_DIST_SHIFT) & 63u) : 0, lane);
      if (valid && (d & GD_LAST))
        partial[d & GD_SLOT_MASK] = to_bits<T>(t);
    };
    SWord va[NV], vb[NV];
    uint4 ca[2], cb[2];
    uint32_t da = 0, db = 0;
    int sid = s0 + tid;
    const bool live = stage([&]() { load(sid, va, ca, da); });
    if (!live) {
      for (int q = sid; q < s1; q += TBS) {   // a dead tile: identity partials, found through the strips' gdest words alone
        const uint32_t d = gdest[q - sbase];
        if (d & GD_LAST) partial[d & GD_SLOT_MASK] = ident;
      }
    } else if (sid < s1) {
      for (;;) {
        load(sid + TBS, vb, cb, db);
        consume(sid, va, ca, da);
        sid += TBS;
        if (sid >= s1) break;
        load(sid + TBS, va, ca, da);
        consume(sid, vb, cb, db);
        sid += TBS;
        if (sid >= s1) break;
      }
    }
  }
}


template <class SR, int VC>
__global__ __launch_bounds__(TBS) void spmv_tiled_phase1(
    const TileChunk *__restrict__ chunks, const void *__restrict__ tval_or_code,
    const uint32_t *__restrict__ vdict, const uint16_t *__restrict__ tcol,
    const uint32_t *__restrict__ gdest, const uint32_t *__restrict__ obase, const uint32_t *__restrict__ x, int32_t cols,
    uint32_t *__restrict__ P, uint32_t *__restrict__ partial, const int32_t *gate, int32_t skip_dead_tiles,
    uint32_t *__restrict__ tile_live) {
  __shared__ uint32_t xs[TCOLS + 4];
  __shared__ uint32_t ds[VC == 3 ? VDICT16 : (VC ? VDICT : 1)];
  static_assert(TBS * 4 == VDICT16, "one 16-byte load per thread stages the two-byte dictionary");
  const TileChunk ch = chunks[blockIdx.x];
  if (ch.s >= ch.e || (gate != nullptr && *gate == 0))
    return;   // filler that keeps the XCD-aligned chunk order / the iteration loop is over (StepDev::gate)
  SH_STAT(const uint64_t st_t0 = __builtin_amdgcn_s_memrealtime(); __shared__ uint64_t st_staged;)
  SH_STAT(auto stamp = [&]() { if (threadIdx.x == 0) st_staged = __builtin_amdgcn_s_memrealtime(); };)
#ifdef SH_STATS
  tiled_phase1_chunk<SR, VC>(ch, xs, ds, tval_or_code, vdict, tcol, gdest, obase, x, cols, P, partial, skip_dead_tiles != 0, tile_live, stamp);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && g_p1_stats) {   // per chunk: kind, entries, start, staged, end (100 MHz ticks)
    uint64_t *S = g_p1_stats + (size_t)blockIdx.x * 5;
    S[0] = ch.hs <= ch.s; S[1] = (uint64_t)(ch.e - ch.s); S[2] = st_t0; S[3] = st_staged; S[4] = __builtin_amdgcn_s_memrealtime();
  }
#else
  tiled_phase1_chunk<SR, VC>(ch, xs, ds, tval_or_code, vdict, tcol, gdest, obase, x, cols, P, partial, skip_dead_tiles != 0, tile_live);
#endif
}

// Dead pieces (semirings with absorbing words: the first iterations of a search touch a handful of column tiles).
// Between the phases of such a launch: the piece table phase 2 reads, with the pieces of the tiles phase 1 found dead
// (tile_live) replaced by PIECE_DEAD -- phase 2's loaders then fetch ONE group of identity words (written here, behind
// the last product of P) instead of the piece: a cache hit per wave-instruction instead of ~215 bytes of HBM per
// piece, and phase 1 did not write those products either.  ~2 us for 2.2 M pieces.
constexpr int32_t PIECE_DEAD = INT32_MIN;
static __global__ void tiled_mark_dead(const int32_t *__restrict__ ptab, const uint16_t *__restrict__ ptile, const uint32_t *__restrict__ tile_live,
                                       int32_t *__restrict__ ptab_live, int64_t n_pieces, uint32_t *__restrict__ ident_group, uint32_t ident,
                                       const int32_t *gate) {
  if (gate != nullptr && *gate == 0) return;
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < 4) ident_group[k] = ident;
  if (k < n_pieces) ptab_live[k] = tile_live[ptile[k]] ? ptab[k] : PIECE_DEAD;
}

// The prefetch of phase 2 goes through inline-asm loads.  hipcc tracks vmcnt only for loads it
// emitted itself and is conservative across the loop back-edge: with compiler-visible loads it
// drained the whole prefetch in front of the reduction, waited between the refills, and the P
// addresses of the next bin depended on a source-index load issued in the same iteration (~2 us per bin
// with nothing else in flight).  With asm loads the compiler inserts no waits for them at all;
// the single hand-placed wait is phase2_wait_all() at the top of the loop, whose operand list
// ties every prefetch register to it so that no use can be scheduled above it.
__device__ __forceinline__ void async_load(v4u32 &dst, const void *addr) {
  asm volatile("global_load_dwordx4 %0, %1, off" SH_POL_STR(SH_P2_P_POL) : "=v"(dst) : "v"(addr) : "memory");
}
__device__ __forceinline__ void async_load(v2u32 &dst, const void *addr) {
  asm volatile("global_load_dwordx2 %0, %1, off" SH_POL_STR(SH_P2_S_POL) : "=v"(dst) : "v"(addr) : "memory");
}
__device__ __forceinline__ void async_load(uint32_t &dst, const void *addr) {
  asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(addr) : "memory");
}
// The same from a wave-uniform base (a scalar register pair) plus a 32-bit byte offset per lane: no 64-bit address per lane to
// compute -- phase 2 is bound by the instructions its waves issue, and a loader step spent a quarter of its on addresses.
__device__ __forceinline__ void async_load_at(v4u32 &dst, const void *sbase, uint32_t byte_off) {
  asm volatile("global_load_dwordx4 %0, %1, %2" SH_POL_STR(SH_P2_P_POL) : "=v"(dst) : "v"(byte_off), "s"(sbase) : "memory");
}
__device__ __forceinline__ void async_load_at(v2u32 &dst, const void *sbase, uint32_t byte_off) {
  asm volatile("global_load_dwordx2 %0, %1, %2" SH_POL_STR(SH_P2_S_POL) : "=v"(dst) : "v"(byte_off), "s"(sbase) : "memory");
}
__device__ __forceinline__ void async_load_at(uint32_t &dst, const void *sbase, uint32_t byte_off) {
  asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(byte_off), "s"(sbase) : "memory");
}
struct UniformAt { const void *base; uint32_t off; };   // a wave-uniform base and a lane's byte offset

// ---------------------------------------------------------------------------------------------
// Phase 2, wave-specialised (default).  One 1024-thread workgroup per CU; waves 0-3 are LOADERS,
// waves 4-15 REDUCERS, and the LDS holds two product images:
//   loaders   stream bin j+1 (P pieces via the piece tables, slots) and scatter it into image (j+1)&1;
//   reducers  reduce bin j out of image j&1 and write the rows.
// (Round 4 measured where this kernel's time goes -- profiles/r04_phase2_role_profile.log, r04_phase2_decoupled_roles_probe.log:
// while the reducers were the slower role the loader waves waited at barriers, not for loads, and the kernel looked
// compute-bound; with the reducers out of the way -- batched LDS reads, DPP trees, cooperative rows handed out on demand --
// it runs at what the reasoning below says: the read requests a CU keeps in flight per memory latency, 8.1 M of them per launch.)
// Why: a CU sustains only what its miss queue holds per memory latency, so HBM time is lost
// whenever no wave of the CU has a load to issue.  With every wave alternating between "stream
// a bin" and "reduce a bin" (spmv_tiled_phase2 above, two workgroups per CU) the two workgroups
// fall into step and the CU's load queue runs dry during the reductions: 277 us where the loads
// alone take 200 us.  Here four waves do nothing but keep loads in flight -- a rolling window of
// two quarter-bin steps per thread, hand-scheduled: asm loads, one s_waitcnt vmcnt(8) per step --
// and the reduction runs beside them on the other twelve.
// Per-thread order of VMEM issue in the loaders (in-order return makes the wait a constant):
//   step q:  wait vmcnt(8)  -> P/S(q) and the piece words of step q+2 have landed, P/S(q+1) (8 loads) still fly
//            scatter P/S(q) into the image; request the piece words of q+3 (4 loads); issue P/S(q+2) (8 loads)
// Barriers (s_barrier is workgroup-wide, so both roles execute the same two per bin): MID is the
// one inside the reduction (list hand-over), END swaps the images.  While the loaders fill the
// very first image the reducers have nothing to reduce: they add up the heavy rows' partials
// (the job of spmv_heavy_fixup, same summation order) instead of idling.
#ifndef SH_P2S_LD
#define SH_P2S_LD 256
#endif
constexpr int P2S_BS = 1024, P2S_LD = SH_P2S_LD, P2S_RD = P2S_BS - P2S_LD;   // threads: all, loaders, reducers
constexpr int P2S_K = 4;                       // groups (of 4 products) per loader thread per step
constexpr int P2S_Q = P2S_K * P2S_LD;          // groups per step: 1024 (a quarter bin) with 256 loaders
constexpr int P2S_NS = TBIN / 4 / P2S_Q;       // steps per bin: 4
constexpr int P2S_RPU = (TBIN_ROWS + P2S_RD) / P2S_RD;   // row_ptr entries per reducer thread: 3
static_assert(P2S_NS * P2S_Q * 4 == TBIN && P2S_NS % 2 == 0 && TBIN_ROWS + 1 <= P2S_RPU * P2S_RD, "phase-2 geometry");
// a light row is shorter than TBIN / 4: never the workgroup-wide reduction path, whose extra barriers only the reducers would execute
static_assert(TBIN / 4 <= RL_WAVE, "light rows must not reach the workgroup-wide reduction of reduce_rows_from_lds");

// LDS of the phase-2 role
struct P2Lds {
  uint32_t prod[2][TBIN + RL_PAD];   // [TBIN]: where the loaders drop padding products; the rest: see RL_PAD
  // the light row offsets of a bin, raw, from the 16-byte boundary at or below the bin's first row (the loaders copy them)
  alignas(16) int32_t rp[2][TBIN_ROWS + 12];   // (the last four words: where a loader thread without a part of the bin's offsets drops its words)
  uint32_t dots[TBIN_ROWS];
  ReduceScratch<P2S_RD, TBIN> sc;
};

// One workgroup's share of phase 2: bins b0, b0 + stride, ... (nb of them) of bins[].
template <class SR>
__device__ __forceinline__ void tiled_phase2_run(
    P2Lds &L, const RowBin *__restrict__ bins, const int b0, const int nb, const int stride,
    const int32_t *__restrict__ row_ptr, const uint32_t *__restrict__ P, int32_t last_group,
    const uint16_t *__restrict__ pslot, const uint4 *__restrict__ gblk, const int32_t *__restrict__ ptab,
    const LongRow *__restrict__ heavy_rows, int32_t n_heavy, int32_t heavy_first, int32_t heavy_stride,
    const uint32_t *__restrict__ heavy_partial, const uint32_t *__restrict__ y, typename SR::T alpha,
    typename SR::T beta, const bool use_y, uint32_t *__restrict__ out, const StepDev &st) {
  auto &prod = L.prod;
  auto &rp = L.rp;
  auto &dots = L.dots;
  auto &sc = L.sc;
  const int tid = threadIdx.x;
  // epilogues that read y or the previous vector run as a coalesced pass after one more barrier (MID2)
  const bool staged = use_y || st.changed != nullptr;
  auto bin_at = [&](int j) -> RowBin { return bins[b0 + min(j, nb - 1) * stride]; };   // clamped: scalar loads
  // Piece reporting (StepDev::done): once this workgroup has reduced j of its bins, every piece whose bins all lie
  // below its next bin is complete as far as this workgroup is concerned (its heavy rows were written before the
  // first bin).  Wave-uniform and identical in both roles: the reducers drain their row stores, all waves meet, one
  // lane arrives.  (The loaders store nothing: their prefetch window stays in flight.)
  int reported = 0;   // pieces [0, reported) reported
  auto report = [&](int j_done) {
    if (!reports(st)) return;
    const int n_pieces = st.pcs->n_pieces;
    if (reported >= n_pieces) return;
    const int next_bin = j_done < nb ? b0 + j_done * stride : 0x7FFFFFFF;
    int upto = reported;
    while (upto < n_pieces && st.pcs->piece_bin_end[upto] <= next_bin) upto++;
    if (upto == reported) return;
    if (__builtin_amdgcn_readfirstlane(tid) >= P2S_LD) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (a reducer wave; wave-uniform test)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    int t_here = tid;
    asm volatile("" : "+v"(t_here));   // (compared here, not once in front of the bin loop: see the note at `chg` below)
    if (t_here == P2S_LD) {
      // the changed word travels with the last piece: it was raised by plain stores (millions of rows may raise it: they
      // stay in the XCD's L2); this workgroup's share of it is written through once, ahead of its last arrival
      int32_t *chg = st.changed;
      asm volatile("" : "+s"(chg));   // (the test stays here: hoisted out of the bin loop its lane mask was one scalar pair too many -- a spill)
      if (upto == n_pieces && chg != nullptr &&
          __hip_atomic_load(chg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
        __hip_atomic_store(chg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      pieces_arrive(st, reported, upto);
    }
    reported = upto;
  };
  if (tid < 8)
    sc.cnt[tid] = 0;
  lds_barrier();

  static_assert(P2S_LD % 64 == 0, "a wave is all loader or all reducer");
  if (__builtin_amdgcn_readfirstlane(tid) < P2S_LD) {   // (wave-uniform: a scalar branch, no exec mask saved across a role's body)
    // ------------------------------------------------------------------ loaders
    const v4u32 *P4 = reinterpret_cast<const v4u32 *>(P);
    // Where the products of step q = 4*j + quarter lie in P.  The bin's groups are laid out piece by piece (one piece
    // per column tile); instead of a source address per group (1 B per product of HBM traffic) the plan keeps, per
    // 64 groups, a record {mask of the groups that start a piece, pieces started before the block} -- one scalar
    // load per wave-instruction -- and per piece one word ptab = (P group index of the piece) - (its group index in
    // the bin): a lane counts the piece starts up to its own group (mbcnt of the mask), gathers that word (a table of ~1.2 KB per
    // bin: cache hits) and adds its group index.
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto lbin_at = [&](int j) -> RowBinL {   // clamped: one scalar load of 16 bytes
      return *reinterpret_cast<const RowBinL *>(reinterpret_cast<const char *>(bins + (b0 + min(j, nb - 1) * stride)) + offsetof(RowBin, gb0));
    };
    auto record_of = [&](const RowBinL &bn, int quarter, int k) -> uint4 {
      const int last_blk = max((int)((((uint32_t)bn.n >> 2) + 63u) >> 6), 1) - 1;
      const uint32_t blk = (uint32_t)min(quarter * (P2S_Q / 64) + k * (P2S_LD / 64) + wave, last_blk);   // wave-uniform: a scalar load
      // (a 32-bit unsigned byte offset from the table's base: one s_load with a scalar offset, no 64-bit address to form)
      return *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(gblk) + (((uint32_t)bn.gb0 + blk) << 4));
    };
    auto ptab_index = [&](const RowBinL &bn, const uint4 rec) -> UniformAt {
      // pieces started at or before this lane's group = (started before the block) + (bits of the mask up to the lane's).
      // The mask is shifted down by one on the SCALAR unit so that mbcnt -- bits below the lane -- counts the lane's own
      // bit too; what falls off (bit 0: the block's first group starts a piece) joins the base.  Two vector
      // instructions per group instead of seven.
      const uint64_t m = ((uint64_t)rec.y << 32 | rec.x) >> 1;
      const uint32_t base1 = max(rec.z + (rec.x & 1u), 1u);   // (one more than the index of the piece the block's first group lies in)
      const uint32_t upto = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
      return UniformAt{ptab + bn.pt0 - 1, (upto + base1) * 4u};
    };
    // The records of a step are fetched one step before its piece words are requested (a scalar load takes a few
    // hundred cycles; asked for and used in the same step it cost 2 us per bin): rec[] always holds the records
    // of the step whose words issue_gs() requests next.
    uint4 rec[P2S_K];
    auto fetch_rec = [&](const RowBinL &bn, int quarter) {
#pragma unroll
      for (int k = 0; k < P2S_K; k++)
        rec[k] = record_of(bn, quarter, k);
    };
    auto issue_gs = [&](const RowBinL &bn, uint32_t (&g)[P2S_K]) {
#pragma unroll
      for (int k = 0; k < P2S_K; k++)
      {
        const UniformAt at = ptab_index(bn, rec[k]);
        async_load_at(g[k], at.base, at.off);
      }
    };
    auto issue_ps = [&](const RowBinL &bn, int quarter, const uint32_t (&g)[P2S_K], v4u32 (&p)[P2S_K], v2u32 (&s)[P2S_K]) {
      const int n4 = max(bn.n / 4, 1);
      const v2u32 *S4 = reinterpret_cast<const v2u32 *>(pslot + bn.pstart);
#pragma unroll
      for (int k = 0; k < P2S_K; k++) {
#if defined(SH_DBG_P2) && (SH_DBG_P2 & 2)   // tools builds (wrong results): the slot words of a bin come out of one line
        async_load(s[k], S4 + (tid & 15));
#else
        async_load_at(s[k], S4, (uint32_t)min(quarter * P2S_Q + k * P2S_LD + tid, n4 - 1) * 8u);
#endif
        // (clamped: a stale register must not fault)
#if defined(SH_DBG_P2) && (SH_DBG_P2 & 1)   // tools builds (wrong results): P read bin-major, i.e. sequentially
        const int32_t pg = max(0, min(bn.pstart / 4 + min(quarter * P2S_Q + k * P2S_LD + tid, n4 - 1), last_group));
#else
        // (clamped: a stale register must not fault -- one unsigned min: a negative sum wraps to the far end)
        int32_t pg = (int32_t)min((uint32_t)((int32_t)g[k] + min(quarter * P2S_Q + k * P2S_LD + tid, n4 - 1)), (uint32_t)last_group);
        // a piece of a dead tile (tiled_mark_dead): the group of identity words behind the last product
        if (SR::has_absorbing && (int32_t)g[k] == PIECE_DEAD) pg = last_group + 1;
#endif
        async_load_at(p[k], P4, (uint32_t)pg * 16u);   // (P < 4 GB: checked at upload)
      }
    };
    auto wait8 = [&](v4u32 (&p)[P2S_K], v2u32 (&s)[P2S_K], uint32_t (&g)[P2S_K]) {
      asm volatile("s_waitcnt vmcnt(8)"
                   : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]),
                     "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3])
                   :
                   : "memory");
    };
    auto scatter = [&](uint32_t *img, const RowBinL &bn, int quarter, const v4u32 (&p)[P2S_K], const v2u32 (&s)[P2S_K]) {
      const int n4 = bn.n / 4;
#pragma unroll
      for (int k = 0; k < P2S_K; k++) {
        // Branch-free (6 -> 3 instructions per product: v_min_u32_sdwa, v_lshl_add_u32, ds_write_b32): a padding product --
        // slot TSLOT_PAD -- lands in the spare word behind the image, and a lane past the bin's last group (its loads
        // were clamped) drops what it loaded the same way.
#if defined(SH_DBG_P2) && (SH_DBG_P2 & 2)
        const uint32_t gq = (uint32_t)(quarter * P2S_Q + k * P2S_LD + tid) * 4u;
        const uint32_t s0 = (s[k].x & 3u) + gq, s1 = s0 ^ 1u, s2 = s0 ^ 2u, s3 = s0 ^ 3u;
        (void)n4;
#else
        const bool in_bin = quarter * P2S_Q + k * P2S_LD + tid < n4;
        const uint32_t sx = in_bin ? s[k].x : 0xFFFFFFFFu, sy = in_bin ? s[k].y : 0xFFFFFFFFu;
        const uint32_t s0 = sx & 0xFFFFu, s1 = sx >> 16, s2 = sy & 0xFFFFu, s3 = sy >> 16;
#endif
        img[min(s0, (uint32_t)TBIN)] = p[k].x;
        img[min(s1, (uint32_t)TBIN)] = p[k].y;
        img[min(s2, (uint32_t)TBIN)] = p[k].z;
        img[min(s3, (uint32_t)TBIN)] = p[k].w;
      }
    };
    // The light row offsets of the bin being streamed (needed by its reduction, one bin later): words [r0 & ~3, r0 + nr]
    // of row_ptr[], one to three 16-byte loads per loader thread in step 0 -- issued in FRONT of the step's other loads, so
    // that the unchanged s_waitcnt vmcnt(8) of step 1 covers them (loads return in order) -- and stored untouched in step 1.
    constexpr int P2S_RO = (TBIN_ROWS + 8 + 4 * P2S_LD - 1) / (4 * P2S_LD);   // 16-byte loads per loader thread: 3
    v4u32 ro[P2S_RO];
    auto rows_of = [&](int j) -> int2 {   // (r0, nr) of the workgroup's j-th bin: one scalar load
      return *reinterpret_cast<const int2 *>(bins + (b0 + min(j, nb - 1) * stride));
    };
    auto issue_ro = [&](const int2 rn) {
      const int last4 = ((rn.x & 3) + rn.y) & ~3;   // the 16-byte group of the bin's last offset (row nr)
#pragma unroll
      for (int k = 0; k < P2S_RO; k++)
        async_load_at(ro[k], row_ptr + (rn.x & ~3), (uint32_t)min(4 * tid + 4 * P2S_LD * k, last4) * 4u);
    };
    auto store_ro = [&](int32_t *dst, const int2 rn) {
      const int last = (rn.x & 3) + rn.y;
#pragma unroll
      for (int k = 0; k < P2S_RO; k++) {
        const int i4 = 4 * tid + 4 * P2S_LD * k;
        *reinterpret_cast<v4u32 *>(dst + (i4 <= last ? i4 : TBIN_ROWS + 8)) = ro[k];   // (branch-free, as the scatter)
      }
    };
    RowBinL cur = lbin_at(0), nxt = lbin_at(1), nxt2 = lbin_at(2);
    // step q + d (d = 2, 3) seen from step s of the current bin: which bin, which step in it
    auto bin_of = [&](int s_plus_d) -> const RowBinL & {
      return s_plus_d < P2S_NS ? cur : (s_plus_d < 2 * P2S_NS ? nxt : nxt2);
    };
    v4u32 p[2][P2S_K];
    v2u32 sl[2][P2S_K];
    uint32_t g[2][P2S_K];
    // prologue: the piece words of steps 0 and 1 by ordinary loads; then P/S(0), words(2), P/S(1) in steady-state order
    {
#pragma unroll
      for (int k = 0; k < P2S_K; k++) {
        const RowBinL &b1 = bin_of(1);
        const UniformAt a0 = ptab_index(cur, record_of(cur, 0, k)), a1 = ptab_index(b1, record_of(b1, 1 % P2S_NS, k));
        g[0][k] = *reinterpret_cast<const uint32_t *>(static_cast<const char *>(a0.base) + a0.off);
        g[1][k] = *reinterpret_cast<const uint32_t *>(static_cast<const char *>(a1.base) + a1.off);
      }
      issue_ps(cur, 0, g[0], p[0], sl[0]);
      fetch_rec(bin_of(2), 2 % P2S_NS);
      issue_gs(bin_of(2), g[0]);
      fetch_rec(bin_of(3), 3 % P2S_NS);                    // (for the first step of the loop)
      issue_ps(bin_of(1), 1 % P2S_NS, g[1], p[1], sl[1]);
    }
    SH_STAT(uint64_t pf_wait = 0, pf_bar = 0; const uint64_t pf_t0 = __builtin_amdgcn_s_memtime();)
    for (int j = 0; j <= nb; j++) {
      if (j < nb) {
        uint32_t *img = prod[j & 1];
#pragma unroll
        for (int s = 0; s < P2S_NS; s++) {
          const int a = s & 1;
          SH_TIMED(pf_wait, wait8(p[a], sl[a], g[a]))      // P/S(q) and the piece words of q+2 have landed
          scatter(img, cur, s, p[a], sl[a]);
          if (s == 0) issue_ro(rows_of(j));
          if (s == 1) {
            asm volatile("" : "+v"(ro[0]), "+v"(ro[1]), "+v"(ro[2]));   // (landed: older than what this step's wait let through)
            store_ro(rp[j & 1], rows_of(j));
          }
          issue_gs(bin_of(s + 3), g[a ^ 1]);                          // piece words of step q+3 (records fetched a step ago)
          fetch_rec(bin_of(s + 4), (s + 4) % P2S_NS);                 // records of step q+4
          issue_ps(bin_of(s + 2), (s + 2) % P2S_NS, g[a], p[a], sl[a]);   // P/S(q+2)
          if (s == P2S_NS / 2 - 1)
            SH_TIMED(pf_bar, lds_barrier())   // MID
        }
        cur = nxt;
        nxt = nxt2;
        nxt2 = lbin_at(j + 3);
      } else {
        SH_TIMED(pf_bar, lds_barrier())   // MID of the last reduction
      }
      if (staged)
        SH_TIMED(pf_bar, lds_barrier())   // MID2
      SH_TIMED(pf_bar, lds_barrier())     // END
      report(j);         // (bins 0 .. j-1 of this workgroup are reduced)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the look-ahead loads must land before the wave moves on
    SH_STAT(if (tid == 0) { uint64_t *S = g_p2_prof + (blockIdx.x & 255) * 16; S[0] = __builtin_amdgcn_s_memtime() - pf_t0; S[1] = pf_wait; S[2] = pf_bar; S[3] = (uint64_t)nb; })
  } else {
    // ------------------------------------------------------------------ reducers
    const int rt = tid - P2S_LD;
    RowBin prev = bin_at(0);
    SH_STAT(uint64_t pf_bar = 0, pf_red = 0, pf_in[4] = {0, 0, 0, 0}; const uint64_t pf_t0 = __builtin_amdgcn_s_memtime();)
    for (int j = 0; j <= nb; j++) {
      const RowBin cur = bin_at(j);
      // (the row offsets of the bin being streamed now reach rp[j & 1] through the LOADER waves: loaded by a reducer they had
      // to be waited for behind the reduction, and that wait -- s_waitcnt vmcnt(0): the counter cannot tell the rows' stores
      // from the loads in front of them -- drained every store of the bin's rows, ~2.5 K cycles per bin of the waves that set
      // the kernel's pace: profiles/r04_phase2_role_profile.log)
      int32_t *cnt = sc.cnt + 4 * (j & 1);
      if (j >= 1) {
        // the y / previous-vector words of the rows this lane finishes below are requested now, so that their latency
        // runs under the reduction (R-MAT-23 (min,+) step: phase 2 +50..70 us against a launch without y when they were
        // requested behind MID2; +20 us now: profiles/r03_epilogue_probe_{before,after}.json)
        uint32_t yw[P2S_RPU], pw[P2S_RPU];
        const uint32_t *yq = y;
        asm volatile("" : "+s"(yq));   // (not hoisted out of the bin loop: as a loop-invariant lane mask it was the scalar pair that spilled)
        const int32_t *chq = st.changed;
        asm volatile("" : "+s"(chq));
        const bool same_words = use_y && chq != nullptr && st.prev + st.prev_off == yq;
        if (staged) {
#pragma unroll
          for (int k = 0; k < P2S_RPU; k++) {
            const int i = rt + k * P2S_RD;
            yw[k] = 0u; pw[k] = 0u;
            if (i < prev.nr) {
              const int64_t at = row_element(st, prev.r0 + i);
              if (use_y) yw[k] = y[at];
              if (chq != nullptr && !same_words) pw[k] = st.prev[st.prev_off + at];
            }
          }
        }
        const int32_t *rpp = rp[(j - 1) & 1] + (prev.r0 & 3);   // (row 0 of the bin inside the aligned copy)
#ifdef SH_STATS
        uint64_t *const pf_arg = pf_in;
#else
        uint64_t *const pf_arg = nullptr;
#endif
        SH_TIMED(pf_red, (reduce_rows_from_lds<SR, P2S_RD, TBIN, true>(prod[(j - 1) & 1], rpp, prev.nr, prev.r0, sc, cnt, rt, y,
                                               alpha, beta, use_y, out, st, staged ? dots : nullptr, pf_arg, (uint32_t)prev.csr0)))   // contains MID
        if (staged) {
          lds_barrier();   // MID2
#pragma unroll
          for (int k = 0; k < P2S_RPU; k++) {
            const int i = rt + k * P2S_RD;
            if (i < prev.nr && rpp[i] >= 0)   // (bit 31: a heavy row, written before the first bin)
              finish_row_loaded<SR>(prev.r0 + i, from_bits<typename SR::T>(dots[i]), yw[k], same_words ? yw[k] : pw[k], alpha, beta,
                                    use_y, out, st);
          }
        }
      } else {
        // nothing to reduce yet (the loaders are filling the first image): add up the heavy rows'
        // phase-1 partials meanwhile, one row per wave
        for (int h = heavy_first + (rt >> 6); h < n_heavy; h += heavy_stride)
          heavy_row_by_wave<SR>(heavy_rows[h], heavy_partial, rt & 63, y, alpha, beta, use_y, out, st);
        lds_barrier();   // MID
        if (staged)
          lds_barrier(); // MID2
      }
      if (rt < 4)
        sc.cnt[4 * ((j + 1) & 1) + rt] = 0;   // the other set: last read before the previous END
      SH_TIMED(pf_bar, lds_barrier())     // END
      report(j);
      prev = cur;
    }
    SH_STAT(if (rt == 0) { uint64_t *S = g_p2_prof + (blockIdx.x & 255) * 16; S[4] = __builtin_amdgcn_s_memtime() - pf_t0; S[5] = pf_bar + pf_in[0]; S[6] = pf_red - pf_in[0]; S[7] = (uint64_t)nb; S[8] = pf_in[1]; S[9] = pf_in[2]; S[10] = pf_in[3]; })
    SH_STAT(if ((rt & 63) == 0) { uint64_t *W = g_p2_wave + ((blockIdx.x & 255) * 12 + (rt >> 6)) * 4; W[0] = pf_bar + pf_in[0]; W[1] = pf_in[1]; W[2] = pf_in[2] + pf_in[3]; W[3] = __builtin_amdgcn_s_memtime() - pf_t0; })
  }
}

template <class SR>
__global__ __launch_bounds__(P2S_BS) void spmv_tiled_phase2s(
    const RowBin *__restrict__ bins, int32_t n_bins, const int32_t *__restrict__ row_ptr,
    const uint32_t *__restrict__ P, int32_t last_group, const uint16_t *__restrict__ pslot,
    const uint4 *__restrict__ gblk, const int32_t *__restrict__ ptab, const LongRow *__restrict__ heavy_rows, int32_t n_heavy,
    const uint32_t *__restrict__ heavy_partial, const uint32_t *__restrict__ y, typename SR::T alpha,
    typename SR::T beta, int use_y_i, uint32_t *__restrict__ out, StepDev st) {
  __shared__ P2Lds L;
  if (gate_closed(st))
    return;
  const int G = gridDim.x;
  // consecutive bins on one XCD: workgroups are dealt round-robin over the 8 XCDs (blocks w and
  // w+8 share one) and neighbouring bins own neighbouring pieces of every tile in P, so the 128-B
  // lines at piece boundaries are wanted by both (speed only; any placement is correct)
  const int b0 = (G % 8 == 0) ? (int)(blockIdx.x % 8) * (G / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  if (b0 >= n_bins)
    return;
  const int nb = (n_bins - b0 + G - 1) / G;    // bins of this workgroup: b0, b0+G, ...
  tiled_phase2_run<SR>(L, bins, b0, nb, G, row_ptr, P, last_group, pslot, gblk, ptab, heavy_rows, n_heavy,
                       (int)blockIdx.x * (P2S_RD / 64), G * (P2S_RD / 64), heavy_partial, y, alpha, beta,
                       use_y_i != 0, out, st);
}


} // namespace sh
