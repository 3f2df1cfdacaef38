/* sparseharness_hip.h -- C ABI of the MI355X-native CSR SpMV engine.
 *
 * This is the drop-in boundary for sparseharness's hot path.  The reference
 * has no FFI: its apps subclass Harness<TimingType,SemiRingType>
 * (inc/harness.h:11) and the "operator" is an OpenCL source string bound
 * positionally.  The entry points below are what that class's protected
 * helpers bind to once OpenCL is replaced by HIP; each one names the
 * reference interface it replaces (paths relative to the reference root).
 * The source-compatible C++ mirror that calls them lives in
 * sparseharness_amd/host/inc/harness.h; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - every call returns int: SH_OK (0) or a negative SH_E* code; nothing
 *     exits the process (the reference's checkCLError -> exit(1),
 *     inc/opencl_utils.h:15-23, is re-created by the C++ mirror on top);
 *   - sh_last_error() gives the message of the last failing call on an engine
 *     (or of the last failing sh_engine_create when passed NULL);
 *   - the caller owns host memory, the engine owns device memory until the
 *     matching *_free / sh_engine_destroy;
 *   - one engine per host thread; an engine owns (or borrows) ONE hipStream_t
 *     and all its work is ordered on it (the reference has one in-order
 *     queue, inc/harness.h:79-80);
 *   - all vector/matrix elements are 4 bytes: float for SH_PLUS_TIMES_F32 and
 *     SH_MIN_PLUS_F32, int32 for SH_OR_AND_I32 and SH_MAX_MIN_I32;
 *   - there is NO CPU fallback: without a HIP device sh_engine_create fails
 *     with SH_ENODEVICE.
 */
#ifndef SPARSEHARNESS_HIP_H_
#define SPARSEHARNESS_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SH_ABI_VERSION 3   /* 2: sh_plan_options lost the fused-launch fields and gained `fold`; sh_csr_footprint, sh_plan_row_work;
                              3: sh_row_pieces gained `gate`; sh_csr_piece_state */

enum {
  SH_OK = 0,
  SH_EINVAL = -1,    /* bad argument (null pointer, negative size, bad enum) */
  SH_ENODEVICE = -2, /* no usable HIP device / ordinal out of range */
  SH_EHIP = -3,      /* a HIP runtime call failed; see sh_last_error */
  SH_ENOMEM = -4,    /* host or device allocation failed */
  SH_ESHAPE = -5     /* operand sizes do not match the matrix */
};

/* Semirings = the user functions of example/{spmv,sssp,bfs}/kernel5.json:3
 *   PLUS_TIMES: mult l*r, add x+y, identity 0,       out = dot*alpha + y*beta
 *   MIN_PLUS:   mult |a|+|b|, add min(|a|,|b|), identity FLT_MAX,
 *               out = min(|dot|+|alpha|, |y|+|beta|)
 *   OR_AND:     mult (a!=0)&&(b!=0), add ||, identity 0,
 *               out = (dot&&alpha) || (y&&beta)
 *   MAX_MIN:    (example/scc/kernel5.json:3) mult min, add max, identity INT_MIN,
 *               out = max(min(dot,alpha), min(y,beta))
 * PageRank (example/pr/kernel5.json:3) is PLUS_TIMES with beta = (1-d)/N.      */
typedef enum {
  SH_PLUS_TIMES_F32 = 0,
  SH_MIN_PLUS_F32 = 1,
  SH_OR_AND_I32 = 2,
  SH_MAX_MIN_I32 = 3
} sh_semiring;

/* Launch geometry handed down from the run-file: replaces class Run
 * (inc/run.h:9-32) as consumed by Harness::executeKernel
 * (inc/harness.h:153-158).  The native kernels derive grid AND workgroup size
 * from the matrix schedule built at upload; the run's numbers are recorded by
 * the caller (they appear in the SQL row) and do not shape the launch.
 * May be NULL. */
typedef struct sh_launch {
  uint64_t global[3];
  uint64_t local[3];
} sh_launch;

typedef struct sh_engine sh_engine;
typedef struct sh_csr sh_csr;
typedef struct sh_vec sh_vec;

/* ---- engine: replaces Harness::Harness (inc/harness.h:13-82) ------------ */
int sh_abi_version(void);
/* Number of HIP devices (0 when none / no driver).  Never fails. */
int sh_device_count(void);
/* Create an engine on HIP device `device_ordinal` with its own stream. */
int sh_engine_create(int device_ordinal, sh_engine **out);
/* Same, but all work is enqueued on the caller's hipStream_t (e.g. the
 * current PyTorch stream) instead of an engine-owned one. */
int sh_engine_create_on_stream(int device_ordinal, void *hip_stream, sh_engine **out);
int sh_engine_destroy(sh_engine *e);
/* Replaces Harness::getDeviceName (inc/harness.h:100-107). */
int sh_engine_device_name(sh_engine *e, char *buf, size_t buflen);
/* Replaces deviceGetMaxAllocSize (inc/opencl_utils.h:216-226): free device bytes. */
int sh_engine_max_alloc(sh_engine *e, uint64_t *bytes);
/* Block until everything enqueued on the engine's stream has finished
 * (the reference waits after every enqueue, inc/harness.h:159). */
int sh_engine_synchronize(sh_engine *e);
const char *sh_last_error(const sh_engine *e);

/* ---- matrix: replaces SparseMatrix::cl_encode (src/sparse_matrix.cpp:122-399)
 *      + the two createAndUploadGlobalArg calls of Harness::allocateBuffers
 *      (inc/harness.h:201-205).  Takes the CSR view of the reference's row
 *      structure (row r = ellpackMatrix[r], stored order kept).  `val` is
 *      float[nnz] or int32[nnz] bit patterns.  Builds the device-side launch
 *      schedule (row blocks, long-row segments).  col_idx entries outside
 *      [0, cols) are legal and read as the semiring identity
 *      (bounds ladder of kernel5.json:3). */
int sh_csr_upload(sh_engine *e, int64_t rows, int64_t cols, int64_t nnz,
                  const int32_t *row_ptr, const int32_t *col_idx, const void *val,
                  sh_csr **out);
/* The same with explicit plan options instead of the environment.  sh_csr_upload
 * is sh_csr_upload_ex with sh_plan_options_from_env(): the SH_* variables are
 * read once per upload, never at launch time.  Zero-initialise, then call
 * sh_plan_options_default() and change what you need. */
typedef struct sh_plan_options {
  int32_t plan;            /* 0 auto (size rule, then timing when the columns are local), 1 stream, 2 tiled   [SH_PLAN]     */
  int32_t autotune;        /* 1: time both plans at upload when the rule says tiled and the columns are local [SH_AUTOTUNE] */
  int32_t value_coding;    /* 0 auto (4-bit / 8-bit dictionary codes when the data allows), 8: one-byte codes at most,
                              -1 off (raw 4-byte values)                                                      [SH_VALCODE]  */
  int32_t build_threads;   /* host threads of the layout build, 0 = min(hardware, 16)                         [SH_BUILD_THREADS] */
  int32_t heavy_per_tile;  /* rows averaging >= this many entries per column tile are pre-reduced in phase 1  [SH_HEAVY_PER_TILE] */
  int32_t chunk;           /* entries per phase-1 work item, 0 = by the number of items per CU (48 K or 64 K)  [SH_CHUNK]    */
  int32_t xcd_order;       /* 1: phase-1 work items ordered so that an XCD stages only its eighth of x        [SH_XCD_ORDER] */
  int32_t fold;            /* 1: phase 1 folds the entries of one row inside one column tile into ONE product before
                              it travels through P (a fifth of the light products of a power-law matrix)       [SH_FOLD]     */
  int32_t or_and_bits;     /* 1: also build the bit-blocked layout that SH_OR_AND_I32 launches then run on (x as a bitmap,
                              4 B per entry, no product array: a BFS iteration moves a third of the bytes); 2: ONLY
                              that layout (the matrix then serves SH_OR_AND_I32 alone)                          [SH_OR_AND_BITS] */
  int32_t build;           /* where the tiled layout is built: 0 default, 1 on the host (threads above), 2 on the device from
                              the CSR arrays (sorts and scans; a failed device step falls back to the host builder).
                              Both produce the same arrays, byte for byte                                    [SH_BUILD=host|device] */
  int32_t placement_tries; /* where hipMalloc puts the big arrays moves the time of one and the same layout by +-2 %: the
                              upload times this many placements of them and keeps the fastest (about 3 ms each for
                              a 200 M-entry matrix). 0 default (6 for matrices with >= 2^22 products, else 1), 1 = take the first         [SH_PLACEMENT_TRIES] */
} sh_plan_options;
void sh_plan_options_default(sh_plan_options *o);
void sh_plan_options_from_env(sh_plan_options *o);
/* Prefix sum (rows + 1 values, work_prefix[0] = 0) of the HBM bytes the engine expects to move per row under the
 * plan it would choose for a matrix of `cols` columns and `nnz` entries -- nnz is the size the PLAN is chosen for
 * (what one rank uploads, i.e. its share of the matrix), not necessarily row_ptr[rows]: what row-range sharding
 * across GPUs balances on (the reference has one device, inc/harness.h:419).  The weights describe the x-tiled and
 * the CSR-stream plan; they say nothing about the bit-blocked (or,and) layout (4 B per live entry: balance on the
 * entries).  Host-only: needs no device.  opt == NULL: the defaults. */
int sh_plan_row_work(int64_t rows, int64_t cols, int64_t nnz, const int32_t *row_ptr, const sh_plan_options *opt,
                     uint64_t *work_prefix);
int sh_csr_upload_ex(sh_engine *e, int64_t rows, int64_t cols, int64_t nnz,
                     const int32_t *row_ptr, const int32_t *col_idx, const void *val,
                     const sh_plan_options *opt, sh_csr **out);
int sh_csr_free(sh_engine *e, sh_csr *m);
/* Who built the matrix's tiled layout: *where = 0 host, 1 device; note (optional, cap bytes) = why the device builder
 * was not used although asked for, or empty. */
int sh_csr_builder(const sh_csr *m, int32_t *where, char *note, int64_t cap);
/* Placement trials of the upload (sh_plan_options::placement_tries): how many placements of the big arrays were timed,
 * the (+,x) launch time of the first one and of the one kept, in ms (0 when only one was tried). */
int sh_csr_placement(const sh_csr *m, int32_t *tries, float *first_ms, float *kept_ms);
int sh_csr_dims(const sh_csr *m, int64_t *rows, int64_t *cols, int64_t *nnz);
/* Algorithmic bytes of one SpMV over this matrix:
 * 8*nnz + 4*(rows+1) + 4*cols + 4*rows [+ 4*rows if y is read]. */
int sh_csr_algorithmic_bytes(const sh_csr *m, int reads_y, uint64_t *bytes);
/* Which execution plan sh_csr_upload chose (SH_PLAN=stream|tiled|auto overrides):
 * 0 = CSR-stream (x gathered from global memory, for L2-resident x),
 * 1 = x-tiled two-phase (x tiles staged in LDS, products re-binned through HBM),
 * 2 = the bit-blocked (or,and) layout alone (sh_plan_options::or_and_bits = 2).
 * streamed_bytes = HBM bytes one SpMV moves by construction under that plan (tiled: 2.5, 3 or 6 B per
 * stream entry + 4 B written and ~6.2 B re-read per product that travels through P + the vectors; the
 * measured figure of a layout is in profiles/measured_traffic.json). */
int sh_csr_plan(const sh_csr *m, int32_t *plan, uint64_t *streamed_bytes);
/* One-line description of the layout built at upload, for logs and bench records, e.g.
 * "tiled values=dict4(16) tiles=306 chunks=3420 bins=6750 heavy_rows=5276 stream=216M light=133.9M products=109M folded"
 * (stream: entries phase 1 reads, padding included; light: entries of light rows; products: what travels through P).
 * values=dict8(k) / dict4(k): the matrix has k <= 256 / <= 16 distinct 4-byte values and the tiled stream
 * carries one-byte / four-bit codes (lossless; SH_VALCODE=8 stops at one-byte codes, SH_VALCODE=off keeps
 * raw values); values=raw otherwise.
 * " tuned(stream=..ms,tiled=..ms)" is appended when the plan was confirmed by timing both at upload:
 * large matrices get the tiled plan by size; when their columns are local (row bins touch less than
 * half of the column tiles) both plans are timed and the CSR-stream plan is kept if > 10 % faster; SH_PLAN=stream|tiled or SH_AUTOTUNE=0 skip the timing. */
int sh_csr_describe(const sh_csr *m, char *buf, size_t buflen);
/* Device memory the matrix holds (the arrays of the plan that runs; the CSR arrays themselves -- 8 B per entry --
 * are uploaded only for the CSR-stream plan or while both plans are timed at upload).  The reference keeps its
 * padded ELLPACK buffers for the life of the process (inc/harness.h:197-250, never released). */
int sh_csr_footprint(const sh_csr *m, uint64_t *device_bytes);

/* ---- vectors: replace createAndUploadGlobalArg / createGlobalArg /
 *      writeToGlobalArg / fillGlobalArg / readFromGlobalArg
 *      (inc/harness.h:266-391) for x, y, output ------------------------- */
int sh_vec_alloc(sh_engine *e, int64_t n, sh_vec **out);
/* Wrap device memory owned by someone else (e.g. a torch tensor); never freed
 * by the engine. */
int sh_vec_wrap(sh_engine *e, void *device_ptr, int64_t n, sh_vec **out);
int sh_vec_free(sh_engine *e, sh_vec *v);
int sh_vec_upload(sh_engine *e, sh_vec *v, const void *host, int64_t n);   /* blocking */
int sh_vec_download(sh_engine *e, const sh_vec *v, void *host, int64_t n); /* blocking */
int sh_vec_fill(sh_engine *e, sh_vec *v, uint32_t pattern32);              /* async   */
int sh_vec_copy(sh_engine *e, sh_vec *dst, const sh_vec *src);             /* async   */
int64_t sh_vec_len(const sh_vec *v);
void *sh_vec_device_ptr(const sh_vec *v);

/* ---- the hot path: replaces Harness::executeKernel (inc/harness.h:149-195)
 *      running a Lift kernel (example/<algo>/kernel*.json:3):
 *        out[r] = epilogue( (+)_j ( x[col_j] (x) val_j ), alpha, y[r], beta )
 *      alpha/beta point to one element of the semiring's type.  y may be NULL
 *      when the epilogue does not read it (PLUS_TIMES or OR_AND with beta==0,
 *      MAX_MIN with beta==INT_MIN).
 *      out must not alias x.  If kernel_ns != NULL the call waits and returns
 *      the device time of the launch(es) in ns (hipEvent START->STOP, as the
 *      reference's CL_PROFILING_COMMAND_START/END, inc/harness.h:183-194);
 *      if NULL the call only enqueues. */
int sh_spmv(sh_engine *e, sh_semiring sr, const sh_csr *A, const sh_vec *x,
            const sh_vec *y, const void *alpha, const void *beta, sh_vec *out,
            const sh_launch *launch, uint64_t *kernel_ns);

/* ---- iterative apps: replaces the do/while of HarnessSSSP::executeRun +
 *      should_terminate_iteration (app/sssp.cpp:97-176) and the BFS twin
 *      (app/bfs.cpp:94-174) with an on-device loop: the convergence test is
 *      fused into the kernel epilogue (float: |in-out| < delta, int: ==) and
 *      only one flag word per iteration crosses PCIe.
 *        launch k: out = kernel(in, y); y aliases in after launch 0.
 *      x holds x0 on entry and the final vector on return (the buffer the
 *      reference's `input` pointer designates after its last swap); y0 is
 *      read by launch 0 only; scratch is clobbered.  *iters counts launches
 *      including the confirming one; max_iters bounds non-terminating graphs
 *      (TODO.md:7-8).  ns_per_iter (may be NULL, capacity max_iters) receives
 *      each launch's device time; total_ns their sum (MULTI_ITERATION_SUM,
 *      app/sssp.cpp:77-84). */
/* The same step for a matrix whose rows live in PIECES of the vectors (multi-GPU iteration driver: the vectors
 * interleave the pieces of all ranks so that piece c of every rank is one contiguous region to all-gather; the
 * reference has one device and no counterpart, app/sssp.cpp:112-153 is the loop this serves).  Row r belongs to
 * piece c = r / piece_rows and is element element_of_piece[c] + (r - c * piece_rows) of out, of y and -- for the
 * convergence test -- of x.  With report != 0 the launch tells when each piece is complete: *done_words then points
 * to n_pieces words in host memory (owned by the matrix) and word c reaches *round once every row of piece c is
 * written and visible system-wide, pieces completing in ascending order while the launch is still running, so the
 * caller can start exchanging piece c while later pieces are being computed.  ONE launch of the ordinary plan: no
 * per-piece matrices.  At most 8 pieces. */
typedef struct sh_row_pieces {
  int32_t n_pieces;
  int32_t piece_rows;
  int64_t element_of_piece[8];
  int32_t report;
  int32_t reserved;
  const int32_t *gate;   /* device word or NULL: a launch whose gate word is 0 when it starts returns at once and writes nothing
                            (what sh_iterate does internally: a caller that enqueues iteration k + 1 before it has read the
                            flags of iteration k passes the device-side OR of those flags here) */
} sh_row_pieces;
int sh_spmv_step_pieces(sh_engine *e, sh_semiring sr, sh_csr *A, const sh_vec *x, const sh_vec *y,
                        const void *alpha, const void *beta, sh_vec *out, const sh_row_pieces *pieces, double delta,
                        int32_t *changed_flag_device, uint32_t *round, const volatile uint32_t **done_words);

/* Diagnosis for a caller whose wait on *done_words timed out (no counterpart in the reference): host_words[8] = the
 * words the caller polls, *round = the latest reporting launch, *expected = arrivals per piece it waits for,
 * arrivals[8] = the device-side arrival counters of that launch, read through a stream of the call's own so that a
 * launch that never ends cannot block the question (0xFFFFFFFF each when even that copy did not finish in 2 s). */
int sh_csr_piece_state(sh_engine *e, sh_csr *A, uint32_t *arrivals, uint32_t *host_words, uint32_t *expected, uint32_t *round);

int sh_iterate(sh_engine *e, sh_semiring sr, const sh_csr *A, sh_vec *x,
               const sh_vec *y0, sh_vec *scratch, const void *alpha,
               const void *beta, double delta, int32_t max_iters,
               const sh_launch *launch, int32_t *iters, int32_t *converged,
               uint64_t *ns_per_iter, uint64_t *total_ns);

/* One iteration step for callers that drive the loop themselves (the
 * multi-GPU driver): out = kernel(x, y) and *changed_flag (device int32,
 * may be NULL) is set to 1 if any row fails the convergence test against
 * `x`.  x/out may address a longer (replicated) vector: row r of this matrix
 * compares x[out_row_offset + r] with out[r].  Enqueue only. */
int sh_spmv_step(sh_engine *e, sh_semiring sr, const sh_csr *A, const sh_vec *x,
                 const sh_vec *y, const void *alpha, const void *beta, sh_vec *out,
                 int64_t x_row_offset, double delta, int32_t *changed_flag_device);

#ifdef __cplusplus
}
#endif
#endif /* SPARSEHARNESS_HIP_H_ */
