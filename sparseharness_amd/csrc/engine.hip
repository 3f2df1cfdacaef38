// engine.hip -- implementation of the C ABI declared in
// include/sparseharness_hip.h: a thin hipMalloc / hipMemcpyAsync / hipStream
// layer (replacing the reference's OpenCL context + CLMemoryManager,
// inc/harness.h:13-82, inc/cl_memory_manager.h:6-29) plus the launch logic of
// the native CSR SpMV kernels (kernels.hip.h).
//
// No CPU fallback: every compute entry point needs a live HIP device.
#include "../../include/sparseharness_hip.h"
#include "kernels.hip.h"
#include "bits.hip.h"
#include "plan_common.h"
#include "plan_host.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>
#include <thread>
#include <atomic>
#include <chrono>

using namespace sh;

struct sh_engine {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipEvent_t ev_iter[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // sh_iterate: one per launch of a batch
  int32_t *d_flags = nullptr;   // per-iteration convergence flags
  int32_t n_flags = 0;
  int32_t *h_flag = nullptr;    // pinned, 64 B: convergence flags of a batch of iterations, read back
  char name[256] = {0};
  int n_cus = 256;
  std::string err;
};

struct sh_csr {
  int64_t rows = 0, cols = 0, nnz = 0;
  int32_t *d_row_ptr = nullptr, *d_col = nullptr;
  uint32_t *d_val = nullptr;
  int32_t *d_blk_row = nullptr;
  int32_t n_stream = 0;
  LongSeg *d_segs = nullptr;
  int32_t n_segs = 0;
  LongRow *d_long = nullptr;
  int32_t n_long = 0;
  uint32_t *d_partial = nullptr;
  // x-tiled two-phase plan (kernels.hip.h); built when plan == PLAN_TILED
  int plan = 0;
  bool tuned = false;           // plan confirmed by timing both at upload (autotune_plan)
  float tuned_ms[2] = {0, 0};   // [stream, tiled]
  RowBin *d_bins = nullptr;
  int32_t n_bins = 0;
  TileChunk *d_chunks = nullptr;
  int32_t n_chunks = 0;
  uint32_t *d_tval = nullptr, *d_gdest = nullptr, *d_gblk = nullptr, *d_P = nullptr, *d_obase = nullptr;
  int32_t *d_ptab = nullptr;
  uint16_t *d_ptile = nullptr;     // per piece: its column tile
  int32_t *d_ptab_live = nullptr;  // per launch of a semiring with absorbing words: ptab with the pieces of dead tiles marked (tiled_mark_dead)
  uint32_t *d_tile_live = nullptr; // per column tile: phase 1 found a word of x that is not absorbing
  int64_t n_pieces = 0;
  uint8_t *d_tcode = nullptr;    // value coding: one-byte dictionary codes instead of d_tval
  uint32_t *d_vdict = nullptr;   // [VDICT] original bit patterns
  int n_vdict = 0;               // 0 = values stored raw
  int code_bits = 0;             // 8 or 4 when n_vdict != 0
  int n_vdict_used = 0;          // distinct values found (<= VDICT)
  uint16_t *d_tcol = nullptr, *d_pslot = nullptr;
  LongRow *d_tlong = nullptr;   // heavy rows (pre-reduced in phase 1)
  int32_t n_tlong = 0;
  uint32_t *d_tpartial = nullptr;
  int32_t *d_lrp = nullptr;     // light row offsets, bit 31 = heavy row
  int64_t light_len = 0, light_entries = 0;   // light part of the stream (padding included) / light entries of the matrix
  int64_t stream_len = 0, p_len = 0;          // stream entries in all / products in P
  int fold = 0;                               // phase 1 folds a row's entries inside a tile into one product
  size_t stream_bytes = 0, tiled_bytes = 0;   // device memory held by the arrays of plan A / plan B
  std::vector<int32_t> bin_r0;                // first row of every row bin (host copy: piece reporting)
  // the (or,and) semiring on bits (bits.hip.h); built when sh_plan_options::or_and_bits asks for it
  BitsItem *d_bits_items = nullptr;
  uint32_t *d_bits_ent = nullptr, *d_bits_partial = nullptr;
  int32_t *d_bits_sub = nullptr, *d_bits_rr0 = nullptr;
  uint64_t *d_xbits = nullptr;
  int32_t n_bits_items = 0, bits_ct = 0;
  int64_t bits_entries = 0;
  size_t bits_bytes = 0;
  bool bits_only = false;                     // no other plan was built: only SH_OR_AND_I32 launches are served
  uint32_t *d_done = nullptr, *h_done = nullptr;   // piece reporting (sh_spmv_step_pieces): arrival counters / host-visible round words
  PieceDev *d_pcs = nullptr;                  // the pieces' geometry as the kernels read it (device copy of pcs_host)
  PieceDev pcs_host{};                        // what d_pcs holds (rewritten only when a call brings another geometry)
  bool pcs_valid = false;
  uint32_t round = 0;                         // reporting launches so far
  uint32_t last_expected = 0;                 // arrivals per piece the latest reporting launch waits for (sh_csr_piece_state)
  bool built_on_device = false;               // the tiled layout was built by plan_gpu.hip
  int placement_tries = 1;                    // placements of the big arrays timed at upload (tune_placement)
  float placement_ms[2] = {0, 0};             // [first placement, the one kept]
  bool skip_minplus = false;                  // every |value| < 2^103: FLT_MAX + |a| == FLT_MAX, so tiles of unreached x words may be skipped
  std::string build_note;                     // why the device builder was not used / fell back (empty: nothing to say)
};
enum { PLAN_STREAM = 0, PLAN_TILED = 1 };

struct sh_vec {
  void *d = nullptr;
  int64_t n = 0;
  bool owned = false;
};

static thread_local std::string g_create_err;

static int fail(sh_engine *e, int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (e)
    e->err = buf;
  else
    g_create_err = buf;
  return code;
}

#define HIP_TRY(e, call)                                                        \
  do {                                                                          \
    hipError_t _r = (call);                                                     \
    if (_r != hipSuccess)                                                       \
      return fail((e), _r == hipErrorOutOfMemory ? SH_ENOMEM : SH_EHIP,         \
                  "%s failed: %s (%s:%d)", #call, hipGetErrorString(_r),        \
                  __FILE__, __LINE__);                                          \
  } while (0)

extern "C" {

int sh_abi_version(void) { return SH_ABI_VERSION; }

int sh_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess)
    return 0;
  return n;
}

const char *sh_last_error(const sh_engine *e) {
  return e ? e->err.c_str() : g_create_err.c_str();
}

static int engine_create(int device, void *stream, bool borrow, sh_engine **out) {
  if (!out)
    return fail(nullptr, SH_EINVAL, "sh_engine_create: out is NULL");
  *out = nullptr;
  int n = sh_device_count();
  if (n <= 0)
    return fail(nullptr, SH_ENODEVICE,
                "no HIP device available: the sparseharness HIP engine has no CPU fallback");
  if (device < 0 || device >= n)
    return fail(nullptr, SH_ENODEVICE, "device ordinal %d out of range [0,%d)", device, n);
  sh_engine *e = new (std::nothrow) sh_engine();
  if (!e)
    return fail(nullptr, SH_ENOMEM, "out of host memory");
  e->device = device;
  hipError_t r = hipSetDevice(device);
  if (r == hipSuccess) {
    if (borrow) {
      e->stream = (hipStream_t)stream;
    } else {
      r = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
      e->own_stream = true;
    }
  }
  if (r == hipSuccess) r = hipEventCreate(&e->ev0);
  if (r == hipSuccess) r = hipEventCreate(&e->ev1);
  if (r == hipSuccess) r = hipHostMalloc((void **)&e->h_flag, 64, hipHostMallocDefault);
  if (r == hipSuccess) memset(e->h_flag, 0, 64);
  hipDeviceProp_t prop;
  if (r == hipSuccess) r = hipGetDeviceProperties(&prop, device);
  if (r != hipSuccess) {
    int rc = fail(nullptr, SH_EHIP, "engine init failed: %s", hipGetErrorString(r));
    delete e;
    return rc;
  }
  snprintf(e->name, sizeof e->name, "%s (%s)", prop.name[0] ? prop.name : "AMD GPU", prop.gcnArchName);
  e->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  *out = e;
  return SH_OK;
}

int sh_engine_create(int device_ordinal, sh_engine **out) {
  return engine_create(device_ordinal, nullptr, false, out);
}
int sh_engine_create_on_stream(int device_ordinal, void *hip_stream, sh_engine **out) {
  return engine_create(device_ordinal, hip_stream, true, out);
}

int sh_engine_destroy(sh_engine *e) {
  if (!e)
    return SH_OK;
  (void)hipSetDevice(e->device);
  (void)hipStreamSynchronize(e->stream);
  if (e->d_flags) (void)hipFree(e->d_flags);
  if (e->h_flag) (void)hipHostFree(e->h_flag);
  if (e->ev0) (void)hipEventDestroy(e->ev0);
  if (e->ev1) (void)hipEventDestroy(e->ev1);
  for (auto ev : e->ev_iter) if (ev) (void)hipEventDestroy(ev);
  if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
  return SH_OK;
}

int sh_engine_device_name(sh_engine *e, char *buf, size_t buflen) {
  if (!e || !buf || buflen == 0)
    return fail(e, SH_EINVAL, "sh_engine_device_name: bad argument");
  snprintf(buf, buflen, "%s", e->name);
  return SH_OK;
}

int sh_engine_max_alloc(sh_engine *e, uint64_t *bytes) {
  if (!e || !bytes)
    return fail(e, SH_EINVAL, "sh_engine_max_alloc: bad argument");
  size_t fr = 0, tot = 0;
  HIP_TRY(e, hipSetDevice(e->device));
  HIP_TRY(e, hipMemGetInfo(&fr, &tot));
  *bytes = fr;
  return SH_OK;
}

int sh_engine_synchronize(sh_engine *e) {
  if (!e)
    return SH_EINVAL;
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return SH_OK;
}


static int dispatch(sh_engine *e, sh_semiring sr, const sh_csr *A, const sh_vec *x, const sh_vec *y,
                    const void *alpha, const void *beta, sh_vec *out, StepDev st);

// The size rule picks the tiled plan for every large matrix, but a large matrix whose columns are
// local (banded, FEM-like) keeps its x window in L2 and streams 8 B/entry under plan A, which the
// tiled plan cannot match.  So when the rule says "tiled" and nobody forced a plan, both are
// timed once on the device (x = 0: the memory behaviour of a launch does not depend on the values)
// and plan A is kept only if it is clearly faster -- 10 % -- so that near-ties, where the two
// plans' float rounding of heavy rows could differ, always resolve the same way.
static void autotune_plan(sh_engine *e, sh_csr *m) {
  sh_vec xv, ov;
  if (hipMalloc(&xv.d, (size_t)std::max<int64_t>(m->cols, 1) * 4) != hipSuccess) return;
  if (hipMalloc(&ov.d, (size_t)std::max<int64_t>(m->rows, 1) * 4) != hipSuccess) { (void)hipFree(xv.d); return; }
  xv.n = m->cols; ov.n = m->rows;
  (void)hipMemsetAsync(xv.d, 0, (size_t)std::max<int64_t>(m->cols, 1) * 4, e->stream);
  const float one = 1.0f, zero = 0.0f;
  float ms[2] = {0, 0};
  bool ok = true;
  for (int plan : {PLAN_TILED, PLAN_STREAM}) {
    m->plan = plan;
    for (int rep = 0; rep < 3 && ok; rep++) {   // one warm-up, then the faster of two
      ok = hipEventRecord(e->ev0, e->stream) == hipSuccess &&
           dispatch(e, SH_PLUS_TIMES_F32, m, &xv, nullptr, &one, &zero, &ov, StepDev{nullptr, nullptr, 0, 0.0}) == SH_OK &&
           hipEventRecord(e->ev1, e->stream) == hipSuccess && hipEventSynchronize(e->ev1) == hipSuccess;
      float t = 0;
      if (ok) ok = hipEventElapsedTime(&t, e->ev0, e->ev1) == hipSuccess;
      if (ok && rep > 0) ms[plan] = (rep == 1) ? t : std::min(ms[plan], t);
    }
  }
  (void)hipStreamSynchronize(e->stream);
  (void)hipFree(xv.d);
  (void)hipFree(ov.d);
  m->plan = (ok && ms[PLAN_STREAM] < 0.9f * ms[PLAN_TILED]) ? PLAN_STREAM : PLAN_TILED;
  // the layout of the plan that lost is of no further use
  if (m->plan == PLAN_STREAM) {
    m->tiled_bytes = 0;
    for (void **p : {(void **)&m->d_bins, (void **)&m->d_chunks, (void **)&m->d_tval, (void **)&m->d_tcol, (void **)&m->d_gdest,
                     (void **)&m->d_pslot, (void **)&m->d_gblk, (void **)&m->d_ptab, (void **)&m->d_P, (void **)&m->d_tlong, (void **)&m->d_tpartial,
                     (void **)&m->d_ptile, (void **)&m->d_ptab_live, (void **)&m->d_tile_live, (void **)&m->d_lrp, (void **)&m->d_tcode, (void **)&m->d_vdict, (void **)&m->d_obase}) {
      if (*p) (void)hipFree(*p);
      *p = nullptr;
    }
  } else {
    m->stream_bytes = 0;
    for (void **p : {(void **)&m->d_row_ptr, (void **)&m->d_col, (void **)&m->d_val, (void **)&m->d_blk_row, (void **)&m->d_segs,
                     (void **)&m->d_long, (void **)&m->d_partial}) {
      if (*p) (void)hipFree(*p);
      *p = nullptr;
    }
  }
  m->tuned = ok;
  m->tuned_ms[0] = ms[PLAN_STREAM];
  m->tuned_ms[1] = ms[PLAN_TILED];
}

// Where hipMalloc happens to put the big arrays moves the SpMV time of one and the same layout by +-2 % (stable for the
// life of the allocation: profiles/r03_placement_probe_*.json; no alignment or stride rule was found behind it).  So the
// upload of a large matrix tries `tries` placements of the four big streams -- the product array, the column codes, the
// value codes and the slots -- times (+,x) launch pairs on each (four back to back after a warm-up)
// and keeps the fastest; the others are freed.  All candidates stay allocated until the choice is made, or the allocator
// would hand the same place out again.  Costs about 5 ms and one copy of the arrays per try, at upload only.
static void tune_placement(sh_engine *e, sh_csr *m, int tries) {
  if (tries <= 1 || m->plan != PLAN_TILED || m->n_bins <= 0 || m->n_chunks <= 0) return;
  sh_vec xv, ov;
  if (hipMalloc(&xv.d, (size_t)std::max<int64_t>(m->cols, 1) * 4) != hipSuccess) return;
  if (hipMalloc(&ov.d, (size_t)std::max<int64_t>(m->rows, 1) * 4) != hipSuccess) { (void)hipFree(xv.d); return; }
  xv.n = m->cols; ov.n = m->rows;
  (void)hipMemsetAsync(xv.d, 0, (size_t)std::max<int64_t>(m->cols, 1) * 4, e->stream);
  const bool coded = m->n_vdict != 0;
  struct Slot { void **field; size_t bytes; bool copy; };
  const Slot slots[4] = {
      {(void **)&m->d_P, (size_t)std::max<int64_t>(m->p_len, 4) * 4 + 16, false},   // (rewritten by every launch)
      {(void **)&m->d_tcol, (size_t)m->stream_len * 2 + SLACK_WIDE, true},
      {coded ? (void **)&m->d_tcode : (void **)&m->d_tval,
       coded ? tcode_bytes(m->code_bits, m->stream_len) + SLACK_TCODE : (size_t)m->stream_len * 4 + SLACK_WIDE, true},
      {(void **)&m->d_pslot, (size_t)m->p_len * 2 + SLACK_WIDE, true}};
  const float one = 1.0f, zero = 0.0f;
  auto time_it = [&]() -> float {   // one warm-up, then four launch pairs back to back as one interval (the steady state of a loop)
    constexpr int REPS = 4;
    float t = 0;
    const StepDev none{nullptr, nullptr, 0, 0.0};
    if (dispatch(e, SH_PLUS_TIMES_F32, m, &xv, nullptr, &one, &zero, &ov, none) != SH_OK ||
        hipEventRecord(e->ev0, e->stream) != hipSuccess)
      return -1.f;
    for (int rep = 0; rep < REPS; rep++)
      if (dispatch(e, SH_PLUS_TIMES_F32, m, &xv, nullptr, &one, &zero, &ov, none) != SH_OK) return -1.f;
    if (hipEventRecord(e->ev1, e->stream) != hipSuccess || hipEventSynchronize(e->ev1) != hipSuccess ||
        hipEventElapsedTime(&t, e->ev0, e->ev1) != hipSuccess)
      return -1.f;
    return t / REPS;
  };
  struct Set { void *p[4]; float ms; };
  std::vector<Set> sets;
  Set cur{{*slots[0].field, *slots[1].field, *slots[2].field, *slots[3].field}, time_it()};
  sets.push_back(cur);
  size_t best = 0;
  size_t set_bytes = 0;
  for (const Slot &sl : slots) set_bytes += sl.bytes;
  for (int t = 1; t < tries && sets[0].ms > 0; t++) {
    size_t fr = 0, tot = 0;   // (the candidates are all held until the end: never take more than half of what is free)
    if (hipMemGetInfo(&fr, &tot) != hipSuccess || fr < 2 * set_bytes + ((size_t)1 << 30)) break;
    Set fresh{{nullptr, nullptr, nullptr, nullptr}, -1.f};
    bool ok = true;
    for (int i = 0; i < 4 && ok; i++) {
      ok = hipMalloc(&fresh.p[i], slots[i].bytes) == hipSuccess;
      if (ok && slots[i].copy)
        ok = hipMemcpyAsync(fresh.p[i], sets[0].p[i], slots[i].bytes, hipMemcpyDeviceToDevice, e->stream) == hipSuccess;
    }
    if (!ok) {
      (void)hipStreamSynchronize(e->stream);
      for (void *q : fresh.p) if (q) (void)hipFree(q);
      break;
    }
    for (int i = 0; i < 4; i++) *slots[i].field = fresh.p[i];
    fresh.ms = time_it();
    sets.push_back(fresh);
    if (fresh.ms > 0 && fresh.ms < sets[best].ms) best = sets.size() - 1;
  }
  (void)hipStreamSynchronize(e->stream);
#ifdef SH_PLAN_EMULATE
  if (getenv("SH_PLACEMENT_LOG"))
    for (size_t k = 0; k < sets.size(); k++)
      fprintf(stderr, "[placement] %2zu  P %p  tcol %p  tcode %p  pslot %p  %.4f ms%s\n", k, sets[k].p[0], sets[k].p[1], sets[k].p[2], sets[k].p[3],
              sets[k].ms, k == best ? "  <- kept" : "");
#endif
  for (int i = 0; i < 4; i++) *slots[i].field = sets[best].p[i];
  for (size_t k = 0; k < sets.size(); k++)
    if (k != best)
      for (void *q : sets[k].p) (void)hipFree(q);
  m->placement_tries = (int)sets.size();
  m->placement_ms[0] = sets[0].ms; m->placement_ms[1] = sets[best].ms;
  (void)hipFree(xv.d);
  (void)hipFree(ov.d);
}

static int choose_plan(const sh_plan_options &opt, int64_t cols, int64_t nnz) {
  if (opt.plan == 1) return PLAN_STREAM;
  if (opt.plan == 2) return PLAN_TILED;
  // auto: x beyond the per-XCD L2 (4 MiB) makes global gathers line-miss bound
  return (cols > (1 << 20) && nnz >= (1 << 22)) ? PLAN_TILED : PLAN_STREAM;
}

// Estimated HBM bytes per row under the plan sh_csr_upload would choose, as a prefix sum: what row-range sharding
// balances on (a shard full of heavy rows would otherwise finish early while the others still stream).  The weights
// follow the tiled plan's traffic: an entry of a light row ~3 B of stream + ~10.5 B for the product that travels
// through P (pair folding saves a seventh of them), an entry of a heavy row ~3 B of stream + the padding of its
// strips; a row costs ~12 B (offsets, result) either way.  CSR-stream plan: 8 B per entry + 12 B per row.
int sh_plan_row_work(int64_t rows, int64_t cols, int64_t nnz, const int32_t *row_ptr, const sh_plan_options *opt_p,
                     uint64_t *work_prefix) {
  if (rows < 0 || cols < 0 || nnz < 0 || !row_ptr || !work_prefix)
    return SH_EINVAL;
  sh_plan_options opt;
  if (opt_p) opt = *opt_p; else sh_plan_options_default(&opt);
  const bool tiled = choose_plan(opt, cols, nnz) == PLAN_TILED && nnz > 0;
  const int CT = (int)std::max<int64_t>(1, (cols + TCOLS - 1) / TCOLS);
  const int64_t heavy_thr = std::min<int64_t>(TBIN / 4, std::max<int64_t>(512, (int64_t)std::max(1, opt.heavy_per_tile) * CT));
  uint64_t acc = 0;
  work_prefix[0] = 0;
  for (int64_t r = 0; r < rows; r++) {
    const int64_t d = (int64_t)row_ptr[r + 1] - row_ptr[r];
    if (d < 0) return SH_ESHAPE;
    acc += 12u + (uint64_t)(!tiled ? 8 * d : (d >= heavy_thr ? 4 * d : 12 * d));
    work_prefix[r + 1] = acc;
  }
  return SH_OK;
}

// sh_plan_options::build == 0: the tiled layout is built on the device from this many entries on (a 200 M-entry matrix:
// 0.11 s instead of 0.72 s per upload, profiles/r03_build_probe.json); below, the host builder's microseconds beat the
// device builder's ~40 allocations and launches.
static constexpr int64_t DEVICE_BUILD_MIN_NNZ = 1 << 20;

void sh_plan_options_default(sh_plan_options *o) {
  if (!o) return;
  memset(o, 0, sizeof *o);
  o->autotune = 1;
  o->heavy_per_tile = 8;
  o->chunk = 0;   // auto (see build_tiled_plan)
  o->xcd_order = 1;
  o->fold = 1;
  o->or_and_bits = 0;
}

void sh_plan_options_from_env(sh_plan_options *o) {
  if (!o) return;
  sh_plan_options_default(o);
  auto num = [](const char *name, int32_t &dst) { if (const char *v = getenv(name)) dst = atoi(v); };
  if (const char *v = getenv("SH_PLAN")) o->plan = !strcmp(v, "stream") ? 1 : (!strcmp(v, "tiled") ? 2 : 0);
  if (const char *v = getenv("SH_AUTOTUNE")) o->autotune = v[0] != '0';
  if (const char *v = getenv("SH_VALCODE")) o->value_coding = !strcmp(v, "off") ? -1 : (!strcmp(v, "8") ? 8 : 0);
  num("SH_BUILD_THREADS", o->build_threads);
  num("SH_HEAVY_PER_TILE", o->heavy_per_tile);
  num("SH_CHUNK", o->chunk);
  if (const char *v = getenv("SH_XCD_ORDER")) o->xcd_order = v[0] != '0';
  if (const char *v = getenv("SH_FOLD")) o->fold = v[0] != '0';
  num("SH_OR_AND_BITS", o->or_and_bits);
  num("SH_PLACEMENT_TRIES", o->placement_tries);
  if (const char *v = getenv("SH_BUILD")) o->build = !strcmp(v, "host") ? 1 : ((!strcmp(v, "device") || !strcmp(v, "gpu")) ? 2 : 0);
}

int sh_csr_upload(sh_engine *e, int64_t rows, int64_t cols, int64_t nnz, const int32_t *row_ptr,
                  const int32_t *col_idx, const void *val, sh_csr **out) {
  sh_plan_options opt;
  sh_plan_options_from_env(&opt);   // the SH_* knobs: read once per upload, never at launch time
  return sh_csr_upload_ex(e, rows, cols, nnz, row_ptr, col_idx, val, &opt, out);
}

int sh_csr_upload_ex(sh_engine *e, int64_t rows, int64_t cols, int64_t nnz, const int32_t *row_ptr,
                     const int32_t *col_idx, const void *val, const sh_plan_options *opt_p, sh_csr **out) {
  sh_plan_options opt;
  if (opt_p) opt = *opt_p; else sh_plan_options_default(&opt);
  if (!e || !out || rows < 0 || cols < 0 || nnz < 0 || !row_ptr || (nnz > 0 && (!col_idx || !val)))
    return fail(e, SH_EINVAL, "sh_csr_upload: bad argument");
  if (rows > INT32_MAX - 1 || cols > INT32_MAX || nnz > INT32_MAX - 8)
    return fail(e, SH_EINVAL, "sh_csr_upload: sizes exceed int32 indexing (shard the matrix)");
  if (row_ptr[0] != 0 || row_ptr[rows] != nnz)
    return fail(e, SH_ESHAPE, "sh_csr_upload: row_ptr[0]=%d row_ptr[rows]=%d, nnz=%lld", row_ptr[0],
                row_ptr[rows], (long long)nnz);
  *out = nullptr;
  HIP_TRY(e, hipSetDevice(e->device));
  sh_csr *m = new (std::nothrow) sh_csr();
  if (!m)
    return fail(e, SH_ENOMEM, "out of host memory");
  m->rows = rows; m->cols = cols; m->nnz = nnz;

  for (int64_t r = 0; r < rows; r++)
    if (row_ptr[r + 1] < row_ptr[r]) {
      delete m;
      return fail(e, SH_ESHAPE, "sh_csr_upload: row_ptr not monotone at row %lld", (long long)r);
    }
  auto cleanup = [&](int rc) { sh_csr_free(e, m); return rc; };
#define HIP_TRY_M(call)                                                         \
  do {                                                                          \
    hipError_t _r = (call);                                                     \
    if (_r != hipSuccess)                                                       \
      return cleanup(fail(e, _r == hipErrorOutOfMemory ? SH_ENOMEM : SH_EHIP,   \
                          "%s failed: %s", #call, hipGetErrorString(_r)));      \
  } while (0)
  // device array of `bytes` (+ slack for the kernels' wide loads), filled from `host` when given; counted in the footprint
#define DEV_ARRAY(ptr, host, bytes, slack)                                                                    \
  do {                                                                                                        \
    HIP_TRY_M(hipMalloc((void **)&(ptr), (size_t)(bytes) + (slack)));                                         \
    *acct += (size_t)(bytes) + (slack);                                                                       \
    if ((host) != nullptr && (bytes) > 0)                                                                     \
      HIP_TRY_M(hipMemcpyAsync((ptr), (host), (size_t)(bytes), hipMemcpyHostToDevice, e->stream));            \
  } while (0)

  // The (or,and) semiring on bits, when asked for (or_and_bits: 1 = beside the ordinary plan, 2 = instead of it: a BFS
  // harness never launches another semiring on its matrix)
  size_t *acct = &m->bits_bytes;
  const int64_t padded = ((nnz + 3) & ~int64_t(3)) + 4;   // the tail is padded so that 16-byte loads at the end stay in bounds
  bool csr_on_device = false;
  auto upload_csr_arrays = [&]() -> int {   // row_ptr / col_idx / val as they are: plan A's arrays, and the device builder's input
    size_t *const acct_before = acct;
    acct = &m->stream_bytes;
    DEV_ARRAY(m->d_row_ptr, row_ptr, (rows + 1) * 4, 0);
    DEV_ARRAY(m->d_col, (const int32_t *)nullptr, padded * 4, 0);
    DEV_ARRAY(m->d_val, (const uint32_t *)nullptr, padded * 4, 0);
    HIP_TRY_M(hipMemsetAsync(m->d_col + (padded - 8 > 0 ? padded - 8 : 0), 0xFF, (padded >= 8 ? 8 : padded) * 4, e->stream));
    HIP_TRY_M(hipMemsetAsync(m->d_val + (padded - 8 > 0 ? padded - 8 : 0), 0, (padded >= 8 ? 8 : padded) * 4, e->stream));
    if (nnz > 0) {
      HIP_TRY_M(hipMemcpyAsync(m->d_col, col_idx, nnz * 4, hipMemcpyHostToDevice, e->stream));
      HIP_TRY_M(hipMemcpyAsync(m->d_val, val, nnz * 4, hipMemcpyHostToDevice, e->stream));
    }
    csr_on_device = true;
    acct = acct_before;
    return SH_OK;
  };
  const bool device_build = opt.build == 2 || (opt.build == 0 && nnz >= DEVICE_BUILD_MIN_NNZ);
  auto drop_csr_arrays = [&]() {
    (void)hipStreamSynchronize(e->stream);
    (void)hipFree(m->d_row_ptr); (void)hipFree(m->d_col); (void)hipFree(m->d_val);
    m->d_row_ptr = nullptr; m->d_col = nullptr; m->d_val = nullptr;
    m->stream_bytes = 0;
    csr_on_device = false;
  };
  if (opt.or_and_bits > 0 && nnz > 0) {
    BitsHost bh;
    int built = -1;   // 1 built, 0 the layout does not apply, -1 not tried / a device step failed: the host builder
    uint32_t *dev_ent = nullptr;
    if (device_build) {
      if (const int rc = upload_csr_arrays()) return rc;
      std::string why;
      built = build_bits_plan_gpu(e->stream, rows, cols, nnz, m->d_row_ptr, m->d_col, m->d_val, bh, &dev_ent, why);
      if (built != 1) { m->build_note = why; bh = BitsHost(); }
    }
    if (built < 0) built = build_bits_plan(rows, cols, nnz, row_ptr, col_idx, (const uint32_t *)val, opt, bh) ? 1 : 0;
    if (built == 1) {
      acct = &m->bits_bytes;
      m->n_bits_items = (int32_t)bh.items.size();
      m->bits_ct = bh.n_ct;
      m->bits_entries = bh.entries;
      m->bits_only = opt.or_and_bits >= 2;
      if (dev_ent) { m->d_bits_ent = dev_ent; *acct += (size_t)bh.ent_len * 4 + SLACK_WIDE; m->built_on_device = true; }
      else DEV_ARRAY(m->d_bits_ent, bh.ent.data(), bh.ent.size() * 4, SLACK_WIDE);
      DEV_ARRAY(m->d_bits_items, bh.items.data(), bh.items.size() * sizeof(BitsItem), 32);
      DEV_ARRAY(m->d_bits_sub, bh.bsub.data(), bh.bsub.size() * 4, 16);
      DEV_ARRAY(m->d_bits_rr0, bh.rr_item0.data(), bh.rr_item0.size() * 4, 0);
      DEV_ARRAY(m->d_xbits, (const uint64_t *)nullptr, (size_t)bh.n_ct * (BITS_BC / 8), 0);
      DEV_ARRAY(m->d_bits_partial, (const uint32_t *)nullptr, (size_t)std::max<size_t>(bh.items.size(), 1) * (BITS_BR / 8), 0);
      HIP_TRY_M(hipStreamSynchronize(e->stream)); // host vectors die at the end of this block
    }
  }
  if (m->bits_only && m->d_bits_items) {
    if (csr_on_device) drop_csr_arrays();
    m->plan = PLAN_STREAM;
    *out = m;
    return SH_OK;
  }
  m->bits_only = false;
  // The tiled plan first: when it is chosen and nothing asks for a timing of both plans, the CSR arrays
  // (8 B per entry) are neither uploaded nor kept -- the tiled kernels read their own layout only.
  TiledHost th;
  TiledDevArrays td;   // the big arrays when the layout was built on the device
  struct TdGuard { TiledDevArrays &t; ~TdGuard() { t.release(); } } td_guard{td};   // (whatever was not adopted below)
  lap(nullptr);
  // Where the tiled layout is built (sh_plan_options::build): on the device from the CSR arrays (plan_gpu.hip; the
  // default), or by the host builder below -- also the fallback when a device step fails.  Same bytes either way.
  bool want_tiled = choose_plan(opt, cols, nnz) == PLAN_TILED && nnz > 0;
  bool tiled = false;
  const bool bits_on_device = m->built_on_device;
  m->built_on_device = false;   // (from here on: the tiled layout)
  if (want_tiled && device_build) {
    if (!csr_on_device)
      if (const int rc = upload_csr_arrays()) return rc;
    lap("H2D of the CSR arrays");
    std::string why;
    const int g = build_tiled_plan_gpu(e->stream, rows, cols, nnz, row_ptr, m->d_row_ptr, m->d_col, m->d_val, opt, e->n_cus, th, td, why);
    lap("device build");
    if (g == 1) {
      tiled = true;
      m->built_on_device = true;
    } else {
      td.release();
      th = TiledHost();
      m->build_note = why;
      if (g == 0) want_tiled = false;   // the layout does not suit this matrix: the host builder would refuse as well
    }
  }
  if (want_tiled && !tiled)
    tiled = build_tiled_plan(rows, cols, nnz, row_ptr, col_idx, (const uint32_t *)val, opt, e->n_cus, th);
  // (only worth timing when the bins touch few of the column tiles, i.e. the columns are local: with
  // scattered columns -- every bin has a piece in nearly every tile -- plan A is several times slower)
  const bool tune = tiled && opt.plan == 0 && opt.autotune && th.tile_fill < 0.5;
  m->plan = tiled ? PLAN_TILED : PLAN_STREAM;
  // The CSR arrays (8 B per entry) stay on the device only for plan A, or while both plans are timed.
  if (!tiled || tune) {
    std::vector<int32_t> pairs;
    std::vector<LongSeg> segs;
    std::vector<LongRow> longs;
    build_schedule(rows, row_ptr, pairs, segs, longs);
    m->n_stream = (int32_t)(pairs.size() / 2);
    m->n_segs = (int32_t)segs.size();
    m->n_long = (int32_t)longs.size();
    if (!csr_on_device)
      if (const int rc = upload_csr_arrays()) return rc;
    acct = &m->stream_bytes;
    // The kernel reads blk_row[b] and blk_row[b+1]; with long rows in between the
    // blocks are not contiguous, so upload the pair list and index it as 2*b.
    DEV_ARRAY(m->d_blk_row, pairs.data(), pairs.size() * 4, 8);
    if (m->n_segs) {
      DEV_ARRAY(m->d_segs, segs.data(), segs.size() * sizeof(LongSeg), 0);
      DEV_ARRAY(m->d_long, longs.data(), longs.size() * sizeof(LongRow), 0);
      DEV_ARRAY(m->d_partial, (const uint32_t *)nullptr, segs.size() * 4, 0);
    }
    HIP_TRY_M(hipStreamSynchronize(e->stream)); // host vectors die at the end of this block
  } else if (csr_on_device) {
    drop_csr_arrays();
  }
  acct = &m->tiled_bytes;
  if (tiled) {
    m->n_bins = (int32_t)th.bins.size();
    m->n_chunks = (int32_t)th.chunks.size();
    m->n_tlong = (int32_t)th.heavy.size();
    m->light_len = th.light_len;
    m->stream_len = th.stream_len;
    m->p_len = th.p_len;
    m->light_entries = th.light_entries;
    m->fold = opt.fold != 0 && TCOL_FOLD != 0;
    m->bin_r0.reserve(th.bins.size());
    for (const RowBin &b : th.bins) m->bin_r0.push_back(b.r0);
    DEV_ARRAY(m->d_bins, th.bins.data(), th.bins.size() * sizeof(RowBin), 0);
    DEV_ARRAY(m->d_chunks, th.chunks.data(), th.chunks.size() * sizeof(TileChunk), 0);
    // the big arrays: already on the device (device builder) or uploaded from the host builder's vectors
#define PLAN_ARRAY(ptr, dev, host_vec, elem_bytes, slack)                                                       \
  do {                                                                                                        \
    if (m->built_on_device) { (ptr) = (dev); (dev) = nullptr; *acct += (size_t)(td_n) * (elem_bytes) + (slack); } \
    else DEV_ARRAY(ptr, (host_vec).data(), (host_vec).size() * (elem_bytes), slack);                          \
  } while (0)
    size_t td_n = 0;
    if (!th.vdict.empty()) {
      m->n_vdict = (int)th.vdict.size();
      m->n_vdict_used = th.vdict_used;
      m->code_bits = th.code_bits;
      m->skip_minplus = true;   // (known from the dictionary alone; raw values would need a pass over the matrix: not skipped)
      for (int k = 0; k < th.vdict_used; k++) m->skip_minplus = m->skip_minplus && (th.vdict[(size_t)k] & 0x7FFFFFFFu) < 0x73000000u;   // |a| < 2^103
      td_n = td.n_tcode; PLAN_ARRAY(m->d_tcode, td.tcode, th.tcode, 1, SLACK_TCODE);
      DEV_ARRAY(m->d_vdict, th.vdict.data(), th.vdict.size() * 4, 0);
    } else {
      td_n = td.n_tval; PLAN_ARRAY(m->d_tval, td.tval, th.tval, 4, SLACK_WIDE);
    }
    td_n = td.n_tcol; PLAN_ARRAY(m->d_tcol, td.tcol, th.tcol, 2, SLACK_WIDE);
    td_n = td.n_gdest; PLAN_ARRAY(m->d_gdest, td.gdest, th.gdest, 4, SLACK_WIDE);
    td_n = td.n_gblk; PLAN_ARRAY(m->d_gblk, td.gblk, th.gblk, 4, SLACK_WIDE);
    td_n = td.n_ptab; PLAN_ARRAY(m->d_ptab, td.ptab, th.ptab, 4, SLACK_WIDE);
    td_n = td.n_ptab; PLAN_ARRAY(m->d_ptile, td.ptile, th.ptile, 2, SLACK_WIDE);
    m->n_pieces = (int64_t)(m->built_on_device ? td.n_ptab : th.ptab.size());
    // dead pieces (launches of a semiring with absorbing words): the marked copy of ptab, the tiles' live words (all live until a launch says otherwise)
    DEV_ARRAY(m->d_ptab_live, (const int32_t *)nullptr, (size_t)m->n_pieces * 4, SLACK_WIDE);
    {
      const size_t ct = (size_t)std::max<int64_t>(1, (m->cols + TCOLS - 1) / TCOLS);
      DEV_ARRAY(m->d_tile_live, (const uint32_t *)nullptr, ct * 4, 0);
      HIP_TRY_M(hipMemsetD32Async((hipDeviceptr_t)m->d_tile_live, 1, ct, e->stream));
    }
    td_n = td.n_pslot; PLAN_ARRAY(m->d_pslot, td.pslot, th.pslot, 2, SLACK_WIDE);
    td_n = td.n_obase; PLAN_ARRAY(m->d_obase, td.obase, th.obase, 4, SLACK_WIDE);
    DEV_ARRAY(m->d_P, (const uint32_t *)nullptr, (size_t)std::max<int64_t>(m->p_len, 4) * 4, 16);
    if (m->built_on_device) { m->d_lrp = (int32_t *)td.lrp; td.lrp = nullptr; *acct += td.n_lrp * 4 + 16; }
    else DEV_ARRAY(m->d_lrp, th.lrp.data(), th.lrp.size() * 4, 16);   // (+16: see plan_gpu.hip)
    if (m->n_tlong) {
      DEV_ARRAY(m->d_tlong, th.heavy.data(), th.heavy.size() * sizeof(LongRow), 0);
      DEV_ARRAY(m->d_tpartial, (const uint32_t *)nullptr, (size_t)th.n_partials * 4, 16);
    }
    HIP_TRY_M(hipStreamSynchronize(e->stream)); // host vectors die at return
    lap("hipMalloc + H2D of the plan");
  }
#undef PLAN_ARRAY
#undef DEV_ARRAY
#undef HIP_TRY_M
  if (tune)
    autotune_plan(e, m);   // frees the arrays of the plan that lost
  {
    // placements of the big arrays to try: the option, else six for a matrix whose product array has >= 2^22 words.
    // Eight fresh processes per setting, one box each time (profiles/r03_ab_placement_tries*.log): 1 / 3 / 6 tries
    // 0.4476 / 0.4384 / 0.4331 ms; on another box 1 / 6 / 12 / 24 tries 0.4456 / 0.4413 / 0.4391 / 0.4417 (means): a trial
    // predicts the steady state of the caller's loop (other x and out vectors) only in part, and past six nothing is gained.
    const int tries = opt.placement_tries > 0 ? opt.placement_tries : (m->plan == PLAN_TILED && m->p_len >= ((int64_t)1 << 22) ? 6 : 1);
    tune_placement(e, m, tries);
  }
  if (bits_on_device && !tiled) m->built_on_device = true;   // (a matrix whose only device-built layout is the bit-blocked one)
  *out = m;
  return SH_OK;
}

int sh_csr_free(sh_engine *e, sh_csr *m) {
  if (!m)
    return SH_OK;
  if (e) {
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
  }
  if (m->d_row_ptr) (void)hipFree(m->d_row_ptr);
  if (m->d_col) (void)hipFree(m->d_col);
  if (m->d_val) (void)hipFree(m->d_val);
  if (m->d_blk_row) (void)hipFree(m->d_blk_row);
  if (m->d_segs) (void)hipFree(m->d_segs);
  if (m->d_long) (void)hipFree(m->d_long);
  if (m->d_partial) (void)hipFree(m->d_partial);
  for (void *p : {(void *)m->d_bins, (void *)m->d_chunks, (void *)m->d_tval, (void *)m->d_tcol, (void *)m->d_gdest,
                  (void *)m->d_pslot, (void *)m->d_gblk, (void *)m->d_ptab, (void *)m->d_ptile, (void *)m->d_ptab_live, (void *)m->d_tile_live, (void *)m->d_P, (void *)m->d_tlong, (void *)m->d_tpartial, (void *)m->d_lrp,
                  (void *)m->d_tcode, (void *)m->d_vdict, (void *)m->d_obase, (void *)m->d_done, (void *)m->d_pcs, (void *)m->d_bits_items,
                  (void *)m->d_bits_ent, (void *)m->d_bits_partial, (void *)m->d_bits_sub, (void *)m->d_bits_rr0, (void *)m->d_xbits})
    if (p) (void)hipFree(p);
  if (m->h_done) (void)hipHostFree(m->h_done);
  delete m;
  return SH_OK;
}

int sh_csr_builder(const sh_csr *m, int32_t *where, char *note, int64_t cap) {
  if (!m)
    return SH_EINVAL;
  if (where) *where = m->built_on_device ? 1 : 0;
  if (note && cap > 0) snprintf(note, (size_t)cap, "%s", m->build_note.c_str());
  return SH_OK;
}

int sh_csr_placement(const sh_csr *m, int32_t *tries, float *first_ms, float *kept_ms) {
  if (!m)
    return SH_EINVAL;
  if (tries) *tries = m->placement_tries;
  if (first_ms) *first_ms = m->placement_ms[0];
  if (kept_ms) *kept_ms = m->placement_ms[1];
  return SH_OK;
}

int sh_csr_dims(const sh_csr *m, int64_t *rows, int64_t *cols, int64_t *nnz) {
  if (!m)
    return SH_EINVAL;
  if (rows) *rows = m->rows;
  if (cols) *cols = m->cols;
  if (nnz) *nnz = m->nnz;
  return SH_OK;
}

int sh_csr_algorithmic_bytes(const sh_csr *m, int reads_y, uint64_t *bytes) {
  if (!m || !bytes)
    return SH_EINVAL;
  *bytes = 8ull * m->nnz + 4ull * (m->rows + 1) + 4ull * m->cols + 4ull * m->rows +
           (reads_y ? 4ull * m->rows : 0ull);
  return SH_OK;
}

int sh_csr_plan(const sh_csr *m, int32_t *plan, uint64_t *streamed_bytes) {
  if (!m)
    return SH_EINVAL;
  if (plan) *plan = m->bits_only ? 2 : m->plan;
  if (streamed_bytes && m->bits_only) {   // 4 B per entry, the x bitmap and the partial result bitmaps written and read once, the vectors
    *streamed_bytes = 4ull * (uint64_t)m->bits_entries + 2ull * (uint64_t)m->n_bits_items * (BITS_BR / 8) +
                      (uint64_t)m->n_bits_items * (BITS_BC / 8) + 4ull * m->cols + 8ull * m->rows;
    return SH_OK;
  }
  if (streamed_bytes) {
    const uint64_t vec = 4ull * (m->rows + 1) + 4ull * m->cols + 4ull * m->rows;
    *streamed_bytes = (m->plan == PLAN_TILED)
                          ? (m->n_vdict ? 2ull : 6ull) * m->stream_len + (m->n_vdict ? (uint64_t)m->stream_len * m->code_bits / 8 : 0ull) + (uint64_t)(m->stream_len - m->light_len) / 4 /* gdest: 4 B per 16-entry strip */ +
                                (uint64_t)m->light_len / 64 /* obase: 4 B per 64 groups */ +
                                4ull * m->p_len /* P written */ + 6ull * m->p_len + m->p_len / 6 /* phase 2: P, slot; piece tables ~0.14 B per product */ +
                                vec /* x once: a tile is re-staged per phase-1 workgroup, but out of its XCD's L2 */
                          : 8ull * m->nnz + vec;
  }
  return SH_OK;
}

int sh_csr_describe(const sh_csr *m, char *buf, size_t buflen) {
  if (!m || !buf || buflen == 0)
    return SH_EINVAL;
  if (m->plan == PLAN_TILED) {
    char vals[32];
    if (m->n_vdict) snprintf(vals, sizeof vals, "dict%d(%d)", m->code_bits, m->n_vdict_used);
    else snprintf(vals, sizeof vals, "raw");
    snprintf(buf, buflen, "tiled values=%s tiles=%lld chunks=%d bins=%d heavy_rows=%d stream=%.1fM light=%.1fM products=%.1fM%s", vals,
             (long long)((m->cols + TCOLS - 1) / TCOLS), m->n_chunks, m->n_bins, m->n_tlong, m->stream_len / 1e6,
             m->light_entries / 1e6, m->p_len / 1e6, m->fold ? " folded" : "");
  } else if (m->bits_only) {
    snprintf(buf, buflen, "bits-only");
  } else {
    snprintf(buf, buflen, "stream values=raw blocks=%d long_rows=%d segments=%d", m->n_stream, m->n_long, m->n_segs);
  }
  if (m->tuned) {
    const size_t len = strlen(buf);
    snprintf(buf + len, buflen - len, " tuned(stream=%.3fms,tiled=%.3fms)", m->tuned_ms[0], m->tuned_ms[1]);
  }
  {
    const size_t len = strlen(buf);
    if (m->d_bits_items)
      snprintf(buf + len, buflen - len, " or_and=bits(items=%d,entries=%.1fM%s)", m->n_bits_items, m->bits_entries / 1e6, m->bits_only ? ",only" : "");
    const size_t len2 = strlen(buf);
    snprintf(buf + len2, buflen - len2, " device=%.3fGB", (double)(m->stream_bytes + m->tiled_bytes + m->bits_bytes) / 1e9);
  }
  return SH_OK;
}

int sh_csr_footprint(const sh_csr *m, uint64_t *device_bytes) {
  if (!m || !device_bytes)
    return SH_EINVAL;
  *device_bytes = (uint64_t)(m->stream_bytes + m->tiled_bytes + m->bits_bytes);
  return SH_OK;
}

// ---------------------------------------------------------------- vectors
int sh_vec_alloc(sh_engine *e, int64_t n, sh_vec **out) {
  if (!e || !out || n < 0)
    return fail(e, SH_EINVAL, "sh_vec_alloc: bad argument");
  *out = nullptr;
  HIP_TRY(e, hipSetDevice(e->device));
  sh_vec *v = new (std::nothrow) sh_vec();
  if (!v)
    return fail(e, SH_ENOMEM, "out of host memory");
  hipError_t r = hipMalloc(&v->d, (n > 0 ? n : 1) * 4);
  if (r != hipSuccess) {
    delete v;
    return fail(e, SH_ENOMEM, "hipMalloc(%lld B) failed: %s", (long long)n * 4, hipGetErrorString(r));
  }
  v->n = n;
  v->owned = true;
  *out = v;
  return SH_OK;
}

int sh_vec_wrap(sh_engine *e, void *device_ptr, int64_t n, sh_vec **out) {
  if (!e || !out || n < 0 || (!device_ptr && n > 0))
    return fail(e, SH_EINVAL, "sh_vec_wrap: bad argument");
  sh_vec *v = new (std::nothrow) sh_vec();
  if (!v)
    return fail(e, SH_ENOMEM, "out of host memory");
  v->d = device_ptr;
  v->n = n;
  v->owned = false;
  *out = v;
  return SH_OK;
}

int sh_vec_free(sh_engine *e, sh_vec *v) {
  if (!v)
    return SH_OK;
  if (v->owned && v->d) {
    if (e) {
      (void)hipSetDevice(e->device);
      (void)hipStreamSynchronize(e->stream);
    }
    (void)hipFree(v->d);
  }
  delete v;
  return SH_OK;
}

int sh_vec_upload(sh_engine *e, sh_vec *v, const void *host, int64_t n) {
  if (!e || !v || (!host && n > 0) || n < 0)
    return fail(e, SH_EINVAL, "sh_vec_upload: bad argument");
  if (n > v->n)
    return fail(e, SH_ESHAPE, "sh_vec_upload: %lld elements into a vector of %lld", (long long)n, (long long)v->n);
  HIP_TRY(e, hipMemcpyAsync(v->d, host, n * 4, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return SH_OK;
}

int sh_vec_download(sh_engine *e, const sh_vec *v, void *host, int64_t n) {
  if (!e || !v || (!host && n > 0) || n < 0)
    return fail(e, SH_EINVAL, "sh_vec_download: bad argument");
  if (n > v->n)
    return fail(e, SH_ESHAPE, "sh_vec_download: %lld elements from a vector of %lld", (long long)n, (long long)v->n);
  HIP_TRY(e, hipMemcpyAsync(host, v->d, n * 4, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return SH_OK;
}

int sh_vec_fill(sh_engine *e, sh_vec *v, uint32_t pattern32) {
  if (!e || !v)
    return fail(e, SH_EINVAL, "sh_vec_fill: bad argument");
  if (v->n > 0)
    HIP_TRY(e, hipMemsetD32Async((hipDeviceptr_t)v->d, (int)pattern32, v->n, e->stream));
  return SH_OK;
}

int sh_vec_copy(sh_engine *e, sh_vec *dst, const sh_vec *src) {
  if (!e || !dst || !src)
    return fail(e, SH_EINVAL, "sh_vec_copy: bad argument");
  if (dst->n != src->n)
    return fail(e, SH_ESHAPE, "sh_vec_copy: length mismatch");
  if (src->n > 0)
    HIP_TRY(e, hipMemcpyAsync(dst->d, src->d, src->n * 4, hipMemcpyDeviceToDevice, e->stream));
  return SH_OK;
}

int64_t sh_vec_len(const sh_vec *v) { return v ? v->n : -1; }
void *sh_vec_device_ptr(const sh_vec *v) { return v ? v->d : nullptr; }

} // extern "C"

// ---------------------------------------------------------------- launches
template <class SR>
static int launch_spmv(sh_engine *e, const sh_csr *A, const sh_vec *x, const sh_vec *y,
                       const void *alpha_p, const void *beta_p, sh_vec *out, StepDev st) {
  using T = typename SR::T;
  T alpha, beta;
  memcpy(&alpha, alpha_p, 4);
  memcpy(&beta, beta_p, 4);
  const bool use_y = SR::reads_y(beta);
  if (use_y && !y)
    return fail(e, SH_EINVAL, "sh_spmv: y is NULL but the epilogue reads it (beta != 0 or min-plus)");
  if (use_y && y->n < A->rows)
    return fail(e, SH_ESHAPE, "sh_spmv: y has %lld elements, matrix has %lld rows", (long long)y->n, (long long)A->rows);
  if constexpr (std::is_same<SR, OrAndI32>::value) {
    if (A->d_bits_items) {   // the (or,and) semiring on bits: x -> bitmap, blocks -> partial result bitmaps, rows
      const int64_t words32 = (int64_t)A->bits_ct * (BITS_BC / 32);
      hipLaunchKernelGGL(bits_pack_x, dim3((unsigned)((words32 * 8 + 255) / 256)), dim3(256), 0, e->stream, (const uint32_t *)x->d,
                         (int32_t)A->cols, (uint32_t *)A->d_xbits, words32, st.gate);
      HIP_TRY(e, hipGetLastError());
      if (A->n_bits_items > 0) {
        hipLaunchKernelGGL(bits_blocks, dim3((unsigned)A->n_bits_items), dim3(BITS_TBS), 0, e->stream, A->d_bits_items,
                           A->d_bits_ent, A->d_bits_sub, (const uint32_t *)A->d_xbits, A->d_bits_partial, st.gate);
        HIP_TRY(e, hipGetLastError());
      }
      if (A->rows > 0) {
        hipLaunchKernelGGL(bits_finish, dim3((unsigned)((A->rows + BITS_FIN_ROWS - 1) / BITS_FIN_ROWS)), dim3(256), 0, e->stream, A->d_bits_rr0,
                           A->d_bits_partial, (int32_t)A->rows, use_y ? (const uint32_t *)y->d : nullptr, alpha, beta, use_y ? 1 : 0,
                           (uint32_t *)out->d, st);
        HIP_TRY(e, hipGetLastError());
      }
      if (st.expected) {
        hipLaunchKernelGGL(report_all_pieces, dim3(1), dim3(64), 0, e->stream, st);
        HIP_TRY(e, hipGetLastError());
      }
      return SH_OK;
    }
  }
  if (A->bits_only)
    return fail(e, SH_EINVAL, "this matrix was uploaded with or_and_bits = 2: it serves SH_OR_AND_I32 launches only");
  if (A->plan == PLAN_TILED) {
    const uint32_t *yp = use_y ? (const uint32_t *)y->d : nullptr;
#ifdef SH_STATS
    static uint64_t *p1_stats = nullptr;   // tools builds: per-chunk timeline of phase 1
    if (getenv("SH_STATS_DUMP")) {
      if (!p1_stats) (void)hipMalloc((void **)&p1_stats, (size_t)1 << 22);
      (void)hipMemsetAsync(p1_stats, 0, (size_t)1 << 22, e->stream);
      (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_p1_stats), &p1_stats, sizeof p1_stats, 0, hipMemcpyHostToDevice, e->stream);
    }
#endif
    bool mark_dead = false;
    {
      const TileChunk *ch = A->d_chunks;
      const dim3 grid((unsigned)A->n_chunks), block(TBS);
      // tiles whose x words are all absorbing are not streamed (semiring.hip.h); (min,+) needs every |value| < 2^103
      const int32_t skip_dead = SR::id == 1 ? (A->skip_minplus ? 1 : 0) : 1;
      // ... and their products are neither written by phase 1 nor read by phase 2: phase 1 leaves a live word per tile,
      // tiled_mark_dead turns it into a copy of the piece table whose dead pieces point at one group of identity
      // words, phase 2 reads the pieces through that copy (semirings with absorbing words only)
      mark_dead = SR::has_absorbing && skip_dead && A->n_bins > 0 && A->n_pieces > 0;
      uint32_t *tile_live = mark_dead ? A->d_tile_live : nullptr;
      if (A->n_chunks > 0) {
        if (A->n_vdict && A->code_bits == 16)
          hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv_tiled_phase1<SR, 3>), grid, block, 0, e->stream,
                             ch, (const void *)A->d_tcode, A->d_vdict, A->d_tcol, A->d_gdest, A->d_obase,
                             (const uint32_t *)x->d, (int32_t)A->cols, A->d_P, A->d_tpartial, st.gate, skip_dead, tile_live);
        else if (A->n_vdict && A->code_bits == 4)
          hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv_tiled_phase1<SR, 2>), grid, block, 0, e->stream,
                             ch, (const void *)A->d_tcode, A->d_vdict, A->d_tcol, A->d_gdest, A->d_obase,
                             (const uint32_t *)x->d, (int32_t)A->cols, A->d_P, A->d_tpartial, st.gate, skip_dead, tile_live);
        else if (A->n_vdict)
          hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv_tiled_phase1<SR, 1>), grid, block, 0, e->stream,
                             ch, (const void *)A->d_tcode, A->d_vdict, A->d_tcol, A->d_gdest, A->d_obase,
                             (const uint32_t *)x->d, (int32_t)A->cols, A->d_P, A->d_tpartial, st.gate, skip_dead, tile_live);
        else
          hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv_tiled_phase1<SR, 0>), grid, block, 0, e->stream,
                             ch, (const void *)A->d_tval, (const uint32_t *)nullptr, A->d_tcol, A->d_gdest, A->d_obase,
                             (const uint32_t *)x->d, (int32_t)A->cols, A->d_P, A->d_tpartial, st.gate, skip_dead, tile_live);
        HIP_TRY(e, hipGetLastError());
      }
    }
    if (mark_dead) {
      const uint32_t ident = SR::identity_bits;
      hipLaunchKernelGGL(tiled_mark_dead, dim3((unsigned)((A->n_pieces + 255) / 256)), dim3(256), 0, e->stream, A->d_ptab, A->d_ptile,
                         (const uint32_t *)A->d_tile_live, A->d_ptab_live, A->n_pieces, A->d_P + std::max<int64_t>(A->p_len, 4), ident, st.gate);
      HIP_TRY(e, hipGetLastError());
    }
    // phase 2; its reducer waves also add up the heavy rows' partials while the loaders fill the first bin
    if (A->n_bins > 0) {
      hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv_tiled_phase2s<SR>), dim3(std::min(A->n_bins, e->n_cus)), dim3(P2S_BS), 0,
                         e->stream, A->d_bins, A->n_bins, A->d_lrp, A->d_P, (int32_t)(std::max<int64_t>(A->p_len, 4) / 4 - 1), A->d_pslot,
                         (const uint4 *)A->d_gblk, mark_dead ? A->d_ptab_live : A->d_ptab, A->d_tlong, A->n_tlong, A->d_tpartial, yp, alpha, beta, use_y ? 1 : 0,
                         (uint32_t *)out->d, st);
      HIP_TRY(e, hipGetLastError());
    }
#ifdef SH_STATS
    if (getenv("SH_STATS_DUMP") && A->n_bins > 0) {   // where the two roles of phase 2 spent their cycles (wave 0 of each role, per workgroup)
      (void)hipStreamSynchronize(e->stream);
      std::vector<uint64_t> pf(256 * 16);
      (void)hipMemcpyFromSymbol(pf.data(), HIP_SYMBOL(g_p2_prof), pf.size() * 8);
      const int G = std::min(A->n_bins, 256);
      double a[16] = {0};
      for (int w = 0; w < G; w++) for (int k = 0; k < 16; k++) a[k] += (double)pf[(size_t)w * 16 + k] / G;
      fprintf(stderr, "[stats] phase 2 per workgroup (%d, %.1f bins each), shader cycles: loaders total %.0f, in vmcnt waits %.0f (%.1f %%), in barriers %.0f (%.1f %%), issuing/scattering %.0f | "
                      "reducers total %.0f, in barriers %.0f (%.1f %%), reducing %.0f (one-lane pass %.0f, 8-lane rows %.0f, 64-lane rows %.0f)\n",
              G, a[3], a[0], a[1], 100 * a[1] / a[0], a[2], 100 * a[2] / a[0], a[0] - a[1] - a[2], a[4], a[5], 100 * a[5] / a[4], a[6], a[8], a[9], a[10]);
    }
    if (getenv("SH_STATS_DUMP") && A->n_bins > 0) {   // the same per reducer wave: who a bin waits for
      std::vector<uint64_t> pw(256 * 12 * 4);
      (void)hipMemcpyFromSymbol(pw.data(), HIP_SYMBOL(g_p2_wave), pw.size() * 8);
      const int G = std::min(A->n_bins, 256);
      fprintf(stderr, "[stats] phase 2 reducer waves (K cycles: at barriers / classifying pass / cooperative rows / total):");
      for (int w = 0; w < 12; w++) {
        double a[4] = {0, 0, 0, 0};
        for (int g = 0; g < G; g++) for (int k = 0; k < 4; k++) a[k] += (double)pw[((size_t)g * 12 + w) * 4 + k] / G;
        fprintf(stderr, "  w%d %.0f/%.0f/%.0f/%.0f", w, a[0] / 1e3, a[1] / 1e3, a[2] / 1e3, a[3] / 1e3);
      }
      fprintf(stderr, "\n");
    }
    if (getenv("SH_STATS_DUMP") && p1_stats) {
      const int nch = A->n_chunks;
      std::vector<uint64_t> hs((size_t)nch * 5);
      (void)hipStreamSynchronize(e->stream);
      (void)hipMemcpy(hs.data(), p1_stats, hs.size() * 8, hipMemcpyDeviceToHost);
      for (int kind = 0; kind < 2; kind++) {
        double n = 0, ent = 0, stage = 0, total = 0;
        uint64_t t0 = ~0ull, t1 = 0;
        for (int i = 0; i < nch; i++) {
          const uint64_t *S = &hs[(size_t)i * 5];
          if (S[1] == 0 || (int)S[0] != kind) continue;
          n++; ent += S[1]; stage += (S[3] - S[2]) / 100.0; total += (S[4] - S[2]) / 100.0;
          t0 = std::min(t0, S[2]); t1 = std::max(t1, S[4]);
        }
        if (n > 0)
          fprintf(stderr, "[stats] phase 1 %s chunks: %.0f, %.0f entries each, staging %.2f us, whole chunk %.2f us (incl. store drain); span %.1f us\n",
                  kind ? "heavy" : "light", n, ent / n, stage / n, total / n, (t1 - t0) / 100.0);
      }
    }
#endif
    if (A->n_bins == 0 && A->n_tlong > 0) {   // every row is heavy: no phase 2 to host the sums
      hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv_heavy_fixup<SR>), dim3(A->n_tlong), dim3(HFIX_BS), 0, e->stream, A->d_tlong,
                         A->d_tpartial, yp, alpha, beta, use_y ? 1 : 0, (uint32_t *)out->d, st);
      HIP_TRY(e, hipGetLastError());
    }
    if (st.expected && A->n_bins == 0) {      // nobody reported: one arrival per piece behind everything
      hipLaunchKernelGGL(report_all_pieces, dim3(1), dim3(64), 0, e->stream, st);
      HIP_TRY(e, hipGetLastError());
    }
    return SH_OK;
  }
  CsrDev dev{A->d_row_ptr, A->d_col, A->d_val, (int32_t)A->rows, (int32_t)A->cols};
  const int grid = A->n_stream + A->n_segs;
  if (grid > 0) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv_csr_kernel<SR>), dim3(grid), dim3(BS), 0, e->stream, dev,
                       (const uint32_t *)x->d, use_y ? (const uint32_t *)y->d : nullptr, alpha, beta,
                       use_y ? 1 : 0, (uint32_t *)out->d, A->d_blk_row, A->n_stream, A->d_segs,
                       A->d_partial, st);
    HIP_TRY(e, hipGetLastError());
  }
  if (A->n_long > 0) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv_long_fixup<SR>), dim3((A->n_long + 63) / 64), dim3(64), 0,
                       e->stream, A->d_long, A->n_long, A->d_partial,
                       use_y ? (const uint32_t *)y->d : nullptr, alpha, beta, use_y ? 1 : 0,
                       (uint32_t *)out->d, st);
    HIP_TRY(e, hipGetLastError());
  }
  if (st.expected) {   // the CSR-stream kernels do not report pieces: one arrival per piece behind them
    hipLaunchKernelGGL(report_all_pieces, dim3(1), dim3(64), 0, e->stream, st);
    HIP_TRY(e, hipGetLastError());
  }
  return SH_OK;
}

static int check_operands(sh_engine *e, const sh_csr *A, const sh_vec *x, const void *alpha,
                          const void *beta, const sh_vec *out, const char *who) {
  if (!e || !A || !x || !alpha || !beta || !out)
    return fail(e, SH_EINVAL, "%s: NULL argument", who);
  if (x->n < A->cols)
    return fail(e, SH_ESHAPE, "%s: x has %lld elements, matrix has %lld columns", who, (long long)x->n, (long long)A->cols);
  if (out->n < A->rows)
    return fail(e, SH_ESHAPE, "%s: out has %lld elements, matrix has %lld rows", who, (long long)out->n, (long long)A->rows);
  if (out->d == x->d && A->rows > 0)
    return fail(e, SH_EINVAL, "%s: out must not alias x", who);
  return SH_OK;
}

static int dispatch(sh_engine *e, sh_semiring sr, const sh_csr *A, const sh_vec *x, const sh_vec *y,
                    const void *alpha, const void *beta, sh_vec *out, StepDev st) {
  switch (sr) {
  case SH_PLUS_TIMES_F32: return launch_spmv<PlusTimesF32>(e, A, x, y, alpha, beta, out, st);
  case SH_MIN_PLUS_F32: return launch_spmv<MinPlusF32>(e, A, x, y, alpha, beta, out, st);
  case SH_OR_AND_I32: return launch_spmv<OrAndI32>(e, A, x, y, alpha, beta, out, st);
  case SH_MAX_MIN_I32: return launch_spmv<MaxMinI32>(e, A, x, y, alpha, beta, out, st);
  default: return fail(e, SH_EINVAL, "unknown semiring %d", (int)sr);
  }
}

extern "C" {

int sh_spmv(sh_engine *e, sh_semiring sr, const sh_csr *A, const sh_vec *x, const sh_vec *y,
            const void *alpha, const void *beta, sh_vec *out, const sh_launch *launch,
            uint64_t *kernel_ns) {
  (void)launch;   // grid and workgroup size come from the matrix schedule (see header)
  int rc = check_operands(e, A, x, alpha, beta, out, "sh_spmv");
  if (rc)
    return rc;
  HIP_TRY(e, hipSetDevice(e->device));
  StepDev st{nullptr, nullptr, 0, 0.0};
  if (kernel_ns)
    HIP_TRY(e, hipEventRecord(e->ev0, e->stream));
  rc = dispatch(e, sr, A, x, y, alpha, beta, out, st);
  if (rc)
    return rc;
  if (kernel_ns) {
    HIP_TRY(e, hipEventRecord(e->ev1, e->stream));
    HIP_TRY(e, hipEventSynchronize(e->ev1));
    float ms = 0.f;
    HIP_TRY(e, hipEventElapsedTime(&ms, e->ev0, e->ev1));
    *kernel_ns = (uint64_t)((double)ms * 1e6);
  }
  return SH_OK;
}

int sh_spmv_step(sh_engine *e, sh_semiring sr, const sh_csr *A, const sh_vec *x, const sh_vec *y,
                 const void *alpha, const void *beta, sh_vec *out, int64_t x_row_offset, double delta,
                 int32_t *changed_flag_device) {
  int rc = check_operands(e, A, x, alpha, beta, out, "sh_spmv_step");
  if (rc)
    return rc;
  if (x_row_offset < 0 || x_row_offset + A->rows > x->n)
    return fail(e, SH_ESHAPE, "sh_spmv_step: x_row_offset %lld + rows %lld exceeds x length %lld",
                (long long)x_row_offset, (long long)A->rows, (long long)x->n);
  HIP_TRY(e, hipSetDevice(e->device));
  StepDev st{changed_flag_device, (const uint32_t *)x->d, x_row_offset, delta};
  return dispatch(e, sr, A, x, y, alpha, beta, out, st);
}

int sh_spmv_step_pieces(sh_engine *e, sh_semiring sr, sh_csr *A, const sh_vec *x, const sh_vec *y,
                        const void *alpha, const void *beta, sh_vec *out, const sh_row_pieces *pc, double delta,
                        int32_t *changed_flag_device, uint32_t *round, const volatile uint32_t **done_words) {
  if (!e || !A || !x || !alpha || !beta || !out || !pc)
    return fail(e, SH_EINVAL, "sh_spmv_step_pieces: NULL argument");
  if (pc->n_pieces < 1 || pc->n_pieces > MAX_PIECES || pc->piece_rows < 1 ||
      (int64_t)pc->piece_rows * pc->n_pieces < A->rows)
    return fail(e, SH_EINVAL, "sh_spmv_step_pieces: %d pieces of %d rows do not cover %lld rows (at most %d pieces)",
                pc->n_pieces, pc->piece_rows, (long long)A->rows, MAX_PIECES);
  if (x->n < A->cols)
    return fail(e, SH_ESHAPE, "sh_spmv_step_pieces: x has %lld elements, matrix has %lld columns", (long long)x->n, (long long)A->cols);
  if (out->d == x->d && A->rows > 0)
    return fail(e, SH_EINVAL, "sh_spmv_step_pieces: out must not alias x");
  HIP_TRY(e, hipSetDevice(e->device));
  StepDev st{changed_flag_device, (const uint32_t *)x->d, 0, delta, pc->gate};
  PieceDev pd{};
  pd.n_pieces = pc->n_pieces;
  pd.piece_rows = pc->piece_rows;
  for (int c = 0; c < pc->n_pieces; c++) {
    const int64_t first = (int64_t)c * pc->piece_rows, rows_c = std::max<int64_t>(0, std::min<int64_t>(A->rows - first, pc->piece_rows));
    const int64_t at = pc->element_of_piece[c];
    if (at < 0 || at + rows_c > out->n || at + rows_c > x->n || (y && at + rows_c > y->n))
      return fail(e, SH_ESHAPE, "sh_spmv_step_pieces: piece %d (%lld rows at element %lld) does not fit the vectors", c, (long long)rows_c, (long long)at);
    pd.piece_delta[c] = at - first;
    // tiled plan: the piece is complete once every bin that starts below its last row + 1 is reduced
    pd.piece_bin_end[c] = (int32_t)(std::lower_bound(A->bin_r0.begin(), A->bin_r0.end(), (int32_t)std::min<int64_t>(first + pc->piece_rows, A->rows)) - A->bin_r0.begin());
  }
  if (pc->n_pieces > 0) pd.piece_bin_end[pc->n_pieces - 1] = (int32_t)A->bin_r0.size();
  if (!A->d_done) {
    HIP_TRY(e, hipMalloc((void **)&A->d_done, MAX_PIECES * 4));
    HIP_TRY(e, hipMemsetAsync(A->d_done, 0, MAX_PIECES * 4, e->stream));
    HIP_TRY(e, hipHostMalloc((void **)&A->h_done, 64, hipHostMallocDefault));
    memset(A->h_done, 0, 64);
    HIP_TRY(e, hipMalloc((void **)&A->d_pcs, sizeof(PieceDev)));
  }
  pd.done = A->d_done;
  pd.done_host = A->h_done;
  // the geometry goes to device memory once (an iteration loop brings the same one every time); a changed one is
  // copied behind the launches already enqueued on this stream, which keep reading the old bytes until then
  if (!A->pcs_valid || memcmp(&pd, &A->pcs_host, sizeof pd) != 0) {
    HIP_TRY(e, hipStreamSynchronize(e->stream));   // (pcs_host is the copy's source: it must not change under a copy in flight)
    A->pcs_host = pd;
    HIP_TRY(e, hipMemcpyAsync(A->d_pcs, &A->pcs_host, sizeof pd, hipMemcpyHostToDevice, e->stream));
    A->pcs_valid = true;
  }
  st.pcs = A->d_pcs;
  if (pc->report) {
    // arrivals per piece and launch: one per workgroup of phase 2, or the single one of report_all_pieces
    st.expected = (A->plan == PLAN_TILED && A->n_bins > 0 && !(sr == SH_OR_AND_I32 && A->d_bits_items)) ? (uint32_t)std::min(A->n_bins, e->n_cus) : 1u;
    A->round++;
    st.round = A->round;
    A->last_expected = st.expected;
    if (round) *round = A->round;
    if (done_words) *done_words = A->h_done;
  }
  // the epilogue reads y through the same row -> element mapping; an epilogue that does not read y gets none
  sh_vec yfull;
  if (y) { yfull = *y; yfull.owned = false; yfull.n = std::max<int64_t>(y->n, A->rows); }
  return dispatch(e, sr, A, x, y ? &yfull : nullptr, alpha, beta, out, st);
}

// Diagnosis of a piece report that does not arrive (the driver's wait timed out): the host words, what the latest
// reporting launch expects, and -- read through a stream of its own, so that a launch that never ends cannot block
// the question -- the device-side arrival counters (0xFFFFFFFF each when the copy itself did not finish in 2 s).
int sh_csr_piece_state(sh_engine *e, sh_csr *A, uint32_t *arrivals, uint32_t *host_words, uint32_t *expected, uint32_t *round) {
  if (!e || !A || !arrivals || !host_words)
    return fail(e, SH_EINVAL, "sh_csr_piece_state: NULL argument");
  if (expected) *expected = A->last_expected;
  if (round) *round = A->round;
  for (int c = 0; c < MAX_PIECES; c++) { arrivals[c] = 0xFFFFFFFFu; host_words[c] = A->h_done ? ((volatile uint32_t *)A->h_done)[c] : 0u; }
  if (!A->d_done)
    return SH_OK;
  HIP_TRY(e, hipSetDevice(e->device));
  hipStream_t side;
  HIP_TRY(e, hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
  // (h_done is 16 words of pinned host memory: the upper 8 receive the counters)
  if (hipMemcpyAsync(A->h_done + MAX_PIECES, A->d_done, MAX_PIECES * 4, hipMemcpyDeviceToHost, side) == hipSuccess) {
    const auto t0 = std::chrono::steady_clock::now();
    bool ready = false;
    while (!(ready = hipStreamQuery(side) == hipSuccess) && std::chrono::steady_clock::now() - t0 < std::chrono::seconds(2))
      std::this_thread::sleep_for(std::chrono::milliseconds(1));
    if (ready)
      for (int c = 0; c < MAX_PIECES; c++) arrivals[c] = ((volatile uint32_t *)A->h_done)[MAX_PIECES + c];
  }
  (void)hipStreamDestroy(side);
  return SH_OK;
}

int sh_iterate(sh_engine *e, sh_semiring sr, const sh_csr *A, sh_vec *x, const sh_vec *y0,
               sh_vec *scratch, const void *alpha, const void *beta, double delta, int32_t max_iters,
               const sh_launch *launch, int32_t *iters, int32_t *converged, uint64_t *ns_per_iter,
               uint64_t *total_ns) {
  (void)launch;
  if (!e || !A || !x || !y0 || !scratch || !alpha || !beta || !iters || !converged || max_iters < 1)
    return fail(e, SH_EINVAL, "sh_iterate: bad argument");
  if (A->rows != A->cols)
    return fail(e, SH_ESHAPE, "sh_iterate: matrix must be square (inc/common.h:49-52)");
  if (x->n < A->rows || scratch->n < A->rows || y0->n < A->rows)
    return fail(e, SH_ESHAPE, "sh_iterate: vectors shorter than the matrix");
  if (scratch->d == x->d && A->rows > 0)
    return fail(e, SH_EINVAL, "sh_iterate: scratch must not alias x");
  HIP_TRY(e, hipSetDevice(e->device));
  if (e->n_flags < 1) {
    HIP_TRY(e, hipMalloc((void **)&e->d_flags, 64));
    e->n_flags = 16;
  }
  // Launches are enqueued ITER_BATCH iterations ahead of the host: iteration i carries the flag word of
  // iteration i - 1 as its gate and returns at once when that flag stayed 0 (nothing changed: the loop is
  // over), so the host only joins in once per batch -- one memset, one 32-byte read-back, one synchronise
  // per 8 iterations instead of per iteration.  The launch count of the reference's do/while
  // (app/sssp.cpp:112-153, confirming launch included) = index of the first flag that stayed 0, plus one.
#ifndef SH_ITER_BATCH
#define SH_ITER_BATCH 8
#endif
  constexpr int ITER_BATCH = SH_ITER_BATCH;   // (1 = one host round trip per iteration, the round-1 loop: tools A/B builds)
  static_assert(ITER_BATCH <= 8, "the flags of a batch are read back into the first 8 words of h_flag");
  if (!e->ev_iter[0])
    for (auto &ev : e->ev_iter) HIP_TRY(e, hipEventCreate(&ev));
  sh_vec *in = x, *out = scratch;
  const sh_vec *y = y0;
  int32_t it = 0;
  bool term = false;
  uint64_t total = 0;
  while (!term && it < max_iters) {
    const int nb = std::min<int32_t>(ITER_BATCH, max_iters - it);
    HIP_TRY(e, hipMemsetAsync(e->d_flags, 0, ITER_BATCH * 4, e->stream));
    HIP_TRY(e, hipEventRecord(e->ev_iter[0], e->stream));
    for (int k = 0; k < nb; k++) {
      StepDev st{e->d_flags + k, (const uint32_t *)in->d, 0, delta, k > 0 ? e->d_flags + (k - 1) : nullptr};
      int rc = dispatch(e, sr, A, in, y, alpha, beta, out, st);
      if (rc)
        return rc;
      HIP_TRY(e, hipEventRecord(e->ev_iter[k + 1], e->stream));
      sh_vec *t = in; in = out; out = t;   // std::swap(input, output), app/sssp.cpp:143
      y = in;                              // setGlobalArg(3, input_mem_ptr), :150
    }
    HIP_TRY(e, hipMemcpyAsync(e->h_flag, e->d_flags, ITER_BATCH * 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    int ran = nb;                          // launches of this batch that did run
    for (int k = 0; k < nb; k++)
      if (e->h_flag[k] == 0) { ran = k + 1; term = true; break; }
    for (int k = 0; k < ran; k++) {
      float ms = 0.f;
      HIP_TRY(e, hipEventElapsedTime(&ms, e->ev_iter[k], e->ev_iter[k + 1]));
      const uint64_t ns = (uint64_t)((double)ms * 1e6);
      if (ns_per_iter)
        ns_per_iter[it + k] = ns;
      total += ns;
    }
    // the gated launches behind the confirming one wrote nothing: the result is what launch `ran` produced
    if ((nb - ran) % 2) { sh_vec *t = in; in = out; out = t; }
    it += ran;
  }
  if (in != x) {   // the final vector lives in `scratch`: hand it back in x
    HIP_TRY(e, hipMemcpyAsync(x->d, in->d, A->rows * 4, hipMemcpyDeviceToDevice, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
  }
  *iters = it;
  *converged = term ? 1 : 0;
  if (total_ns)
    *total_ns = total;
  return SH_OK;
}

} // extern "C"

#ifdef SH_PLAN_EMULATE
#include "debug_tools.h"
#endif

