/* oracle/sh_oracle.c -- CPU restatement of the reference's SpMV path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / reported CPU baseline.  The
 * shipped path (sparseharness_amd/csrc, the C-ABI of include/) never links,
 * imports or falls back to it.
 *
 * PARITY PINNED: every function here is checked bit-for-bit against outputs
 * of the real reference code (the .npz fixtures in tests/golden, produced by
 * oracle/ref/ref_driver.cpp from the sources under /root/reference) by
 * tests/test_oracle.py.
 *
 * Plain C99, single thread, no dependencies.  Each function cites the
 * reference file:line it restates (paths relative to the reference root).
 */
#include <ctype.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define SH_ORACLE_OK 0
#define SH_ORACLE_EOPEN -1     /* src/sparse_matrix.cpp:18-21  -> exit(-1) */
#define SH_ORACLE_EBANNER -2   /* src/sparse_matrix.cpp:23-26  -> exit(-1) */
#define SH_ORACLE_ETYPE -3     /* src/sparse_matrix.cpp:65-69  -> exit(-1) */
#define SH_ORACLE_ESIZE -4     /* src/sparse_matrix.cpp:36-39  -> returns   */
#define SH_ORACLE_ENOMEM -5

enum { SR_PLUS_TIMES_F32 = 0, SR_MIN_PLUS_F32 = 1, SR_OR_AND_I32 = 2, SR_MAX_MIN_I32 = 3 };
enum { NORM_NONE = 0, NORM_PAGERANK = 1, NORM_SCC = 2 };

/* ------------------------------------------------------------------------
 * MatrixMarket -> CSR with the reference's exact semantics.
 *
 * Restates mm_read_banner (src/mmio.cpp:92-165), mm_read_mtx_crd_size
 * (src/mmio.cpp:174-199), SparseMatrix<T>::load_from_file
 * (src/sparse_matrix.cpp:11-70) and calculate_ellpack (:72-119):
 *   - accepted: "matrix coordinate {real,integer,pattern} *"; the symmetry
 *     field only matters through mm_is_symmetric (:60);
 *   - tuples are (I-1, J-1, (T)val); pattern => val = 1.0 (:50-52);
 *   - symmetric off-diagonal entries are mirrored right after the original
 *     (:60-62);
 *   - the row of an entry is its file COLUMN J, its column the file ROW I
 *     (:86,105-109: quirk A-1), rows keep file order (the sort at :113-118
 *     sorts copies: quirk A-2), duplicates are kept;
 *   - the value is narrowed through `int` (:107: quirk A-3).  elem_is_int=0
 *     restates SparseMatrix<float> (double -> float -> int -> float),
 *     elem_is_int=1 restates SparseMatrix<int> (double -> int -> int).
 * Output arrays are malloc'ed; free with oracle_free.  val holds 4-byte
 * elements (float or int32 bit patterns).
 * ---------------------------------------------------------------------- */
static void lower(char *p) {
  for (; *p; ++p)
    *p = (char)tolower((unsigned char)*p);
}

/* normalise: NORM_PAGERANK restates SparseMatrix<float>::pagerank_normalise
 * (src/sparse_matrix.cpp:409-431): column_sums[I] accumulated over the tuples in file order in T
 * arithmetic, then val = (fabs(val) / column_sums[I]) * damping, the expression evaluated in
 * double (::fabs(double) from <cmath>) and rounded to T once; NORM_SCC restates scc_normalise
 * (:433-456): val = (I == J) ? numeric_limits<T>::min() : J.  Both act on the tuple list, i.e.
 * BEFORE the int narrowing of calculate_ellpack (:107). */
int oracle_mm_load_ex(const char *path, int elem_is_int, int normalise, double damping,
                      int32_t *rows_out, int32_t *cols_out, int32_t *hdr_nnz_out, int64_t *nnz_out,
                      int32_t **row_ptr_out, int32_t **col_idx_out, void **val_out) {
  FILE *f = fopen(path, "r");
  if (!f)
    return SH_ORACLE_EOPEN;
  char line[1025], banner[65], mtx[65], crd[65], dtype[65], sym[65];
  if (!fgets(line, sizeof line, f) ||
      sscanf(line, "%64s %64s %64s %64s %64s", banner, mtx, crd, dtype, sym) !=
          5) {
    fclose(f);
    return SH_ORACLE_EBANNER;
  }
  lower(mtx); lower(crd); lower(dtype); lower(sym);
  if (strncmp(banner, "%%MatrixMarket", 14) != 0 || strcmp(mtx, "matrix") != 0) {
    fclose(f);
    return SH_ORACLE_EBANNER;
  }
  int coordinate = strcmp(crd, "coordinate") == 0;
  if (!coordinate && strcmp(crd, "array") != 0) {
    fclose(f);
    return SH_ORACLE_EBANNER;
  }
  int is_real = strcmp(dtype, "real") == 0, is_int = strcmp(dtype, "integer") == 0,
      is_pat = strcmp(dtype, "pattern") == 0, is_cplx = strcmp(dtype, "complex") == 0;
  if (!is_real && !is_int && !is_pat && !is_cplx) {
    fclose(f);
    return SH_ORACLE_EBANNER;
  }
  int symmetric = strcmp(sym, "symmetric") == 0;
  if (!symmetric && strcmp(sym, "general") != 0 && strcmp(sym, "hermitian") != 0 &&
      strcmp(sym, "skew-symmetric") != 0) {
    fclose(f);
    return SH_ORACLE_EBANNER;
  }
  if (!(coordinate && (is_real || is_int || is_pat))) {
    fclose(f);
    return SH_ORACLE_ETYPE;
  }
  /* size line: skip comment lines (src/mmio.cpp:182-186) */
  int M = 0, N = 0, nz = 0;
  do {
    if (!fgets(line, sizeof line, f)) {
      fclose(f);
      return SH_ORACLE_ESIZE;
    }
  } while (line[0] == '%');
  if (sscanf(line, "%d %d %d", &M, &N, &nz) != 3) {
    int got;
    do {
      got = fscanf(f, "%d %d %d", &M, &N, &nz);
      if (got == EOF) {
        fclose(f);
        return SH_ORACLE_ESIZE;
      }
    } while (got != 3);
  }
  size_t cap = (size_t)nz * (symmetric ? 2 : 1) + 1;
  int32_t *ti = malloc(cap * 4), *tj = malloc(cap * 4);
  double *tv = malloc(cap * 8);
  if (!ti || !tj || !tv) {
    fclose(f);
    return SH_ORACLE_ENOMEM;
  }
  size_t n = 0;
  for (int k = 0; k < nz; k++) {
    int I = 0, J = 0;
    double v = 1.0;
    if (is_pat) {
      if (fscanf(f, "%d %d\n", &I, &J) != 2) { /* reference ignores the count */ }
    } else {
      if (fscanf(f, "%d %d %lg\n", &I, &J, &v) != 3) { }
    }
    I--; J--;
    ti[n] = I; tj[n] = J; tv[n] = v; n++;
    if (symmetric && I != J) {
      ti[n] = J; tj[n] = I; tv[n] = v; n++;
    }
  }
  fclose(f);
  /* tuples hold static_cast<T>(val) (:59) */
  for (size_t k = 0; k < n; k++)
    tv[k] = elem_is_int ? (double)(int32_t)tv[k] : (double)(float)tv[k];
  if (normalise == NORM_PAGERANK && !elem_is_int) {
    float *sums = calloc((size_t)N + 1, sizeof(float));
    if (!sums) return SH_ORACLE_ENOMEM;
    for (size_t k = 0; k < n; k++)
      sums[ti[k]] = sums[ti[k]] + (float)tv[k];
    for (size_t k = 0; k < n; k++)
      tv[k] = (double)(float)((fabs((double)(float)tv[k]) / sums[ti[k]]) * (float)damping);
    free(sums);
  } else if (normalise == NORM_SCC) {
    for (size_t k = 0; k < n; k++) {
      if (elem_is_int)
        tv[k] = (ti[k] == tj[k]) ? (double)INT32_MIN : (double)tj[k];
      else
        tv[k] = (ti[k] == tj[k]) ? (double)1.17549435e-38f : (double)(float)tj[k];
    }
  }
  /* calculate_ellpack: histogram over get<1> (= J), rows sized by height() */
  int32_t *rp = calloc((size_t)M + 1, 4);
  int32_t *ci = malloc((n ? n : 1) * 4);
  uint32_t *va = malloc((n ? n : 1) * 4);
  int32_t *fill = calloc((size_t)M + 1, 4);
  if (!rp || !ci || !va || !fill)
    return SH_ORACLE_ENOMEM;
  for (size_t k = 0; k < n; k++)
    rp[tj[k] + 1]++;
  for (int r = 0; r < M; r++)
    rp[r + 1] += rp[r];
  for (size_t k = 0; k < n; k++) {
    int r = tj[k];
    int32_t pos = rp[r] + fill[r]++;
    ci[pos] = ti[k];
    if (elem_is_int) {
      int32_t t = (int32_t)tv[k];      /* static_cast<int>(double), :59 */
      int32_t iv = t;                  /* int val = get<2>, :107 */
      memcpy(&va[pos], &iv, 4);
    } else {
      float t = (float)tv[k];          /* static_cast<float>(double), :59 */
      int32_t iv = (int32_t)t;         /* int val = get<2>, :107 */
      float fv = (float)iv;            /* pair<int,T>(x, val), :108 */
      memcpy(&va[pos], &fv, 4);
    }
  }
  free(ti); free(tj); free(tv); free(fill);
  *rows_out = M; *cols_out = N; *hdr_nnz_out = nz; *nnz_out = (int64_t)n;
  *row_ptr_out = rp; *col_idx_out = ci; *val_out = va;
  return SH_ORACLE_OK;
}

int oracle_mm_load(const char *path, int elem_is_int, int32_t *rows_out,
                   int32_t *cols_out, int32_t *hdr_nnz_out, int64_t *nnz_out,
                   int32_t **row_ptr_out, int32_t **col_idx_out,
                   void **val_out) {
  return oracle_mm_load_ex(path, elem_is_int, NORM_NONE, 0.0, rows_out, cols_out, hdr_nnz_out, nnz_out,
                           row_ptr_out, col_idx_out, val_out);
}

void oracle_free(void *p) { free(p); }

/* ------------------------------------------------------------------------
 * Gold<float>::spmv (inc/spmv_gold.h:9-28), restated literally, including
 * quirk A-4: beta*y.get(value) is added PER NON-ZERO and y is a generator
 * indexed by the entry's value.  app/spmv.cpp:118 only ever passes a
 * ConstYVectorGenerator, so y is restated as that constant.  x.get(col) is
 * x[col].  acc starts at `zero`; the sum runs in stored (file) order.
 * ---------------------------------------------------------------------- */
void oracle_gold_spmv_f32(int32_t rows, const int32_t *row_ptr,
                          const int32_t *col_idx, const float *val,
                          const float *x, float y_const, float alpha,
                          float beta, float zero, float *result) {
  for (int32_t i = 0; i < rows; i++) {
    float acc = zero;
    for (int32_t j = row_ptr[i]; j < row_ptr[i + 1]; j++) {
      float first = alpha * (x[col_idx[j]] * val[j]);
      float second = beta * y_const;
      acc += first + second;
    }
    result[i] = acc;
  }
}

/* The dot loop alone (inc/spmv_gold.h:17-26 with beta = 0): what bench.py
 * times as cpu_baseline ("port", 1 thread).  Same arithmetic as above. */
void oracle_gold_dot_f32(int64_t rows, const int32_t *row_ptr,
                         const int32_t *col_idx, const float *val,
                         const float *x, float alpha, float *result) {
  for (int64_t i = 0; i < rows; i++) {
    float acc = 0.0f;
    for (int32_t j = row_ptr[i]; j < row_ptr[i + 1]; j++)
      acc += alpha * (x[col_idx[j]] * val[j]);
    result[i] = acc;
  }
}

/* The same loop, rows handed out to `threads` OpenMP threads (each row is still summed sequentially in
 * stored order, so the result is bit-identical to the single-thread loop): bench.py's all-cores
 * cpu_baseline ("plus an OpenMP row-parallel variant on all available cores").  Returns
 * the number of threads that actually ran. */
int oracle_gold_dot_f32_omp(int64_t rows, const int32_t *row_ptr,
                            const int32_t *col_idx, const float *val,
                            const float *x, float alpha, float *result, int threads) {
  int used = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
  {
#pragma omp single
    used = omp_get_num_threads();
    /* dynamic chunks: power-law rows make equal row counts unequal work */
#pragma omp for schedule(dynamic, 2048)
    for (int64_t i = 0; i < rows; i++) {
      float acc = 0.0f;
      for (int32_t j = row_ptr[i]; j < row_ptr[i + 1]; j++)
        acc += alpha * (x[col_idx[j]] * val[j]);
      result[i] = acc;
    }
  }
#else
  (void)threads;
  oracle_gold_dot_f32(rows, row_ptr, col_idx, val, x, alpha, result);
#endif
  return used;
}

/* ------------------------------------------------------------------------
 * Semiring user functions, restated from the "source" strings of
 * example/{spmv,sssp,bfs}/kernel5.json:3.
 * ---------------------------------------------------------------------- */
static float pt_mult(float l, float r) { return l * r; }
static float pt_add(float x, float y) { return x + y; }
static float pt_epi(float dp, float alpha, float y, float beta) {
  return (dp * alpha) + (y * beta);
}
static float mp_mult(float a, float b) { return fabsf(a) + fabsf(b); }
static float mp_add(float a, float b) {
  return fabsf(a) < fabsf(b) ? fabsf(a) : fabsf(b);
}
static float mp_epi(float dp, float alpha, float y, float beta) {
  float a = fabsf(dp) + fabsf(alpha);
  float b = fabsf(y) + fabsf(beta);
  return fabsf(a) < fabsf(b) ? fabsf(a) : fabsf(b);
}
static int32_t oa_mult(int32_t a, int32_t b) { return (a != 0) && (b != 0); }
static int32_t oa_add(int32_t a, int32_t b) { return (a != 0) || (b != 0); }
static int32_t oa_epi(int32_t dp, int32_t alpha, int32_t y, int32_t beta) {
  int32_t r1 = (dp != 0) && (alpha != 0);
  int32_t r2 = (y != 0) && (beta != 0);
  return r1 || r2;
}

/* (max,min) of example/scc/kernel5.json:3: int_min, int_max, doubleMinMax */
static int32_t mm_mult(int32_t a, int32_t b) { return a < b ? a : b; }
static int32_t mm_add(int32_t a, int32_t b) { return a > b ? a : b; }
static int32_t mm_epi(int32_t dp, int32_t alpha, int32_t y, int32_t beta) {
  int32_t m1 = dp < alpha ? dp : alpha, m2 = y < beta ? y : beta;
  return m1 > m2 ? m1 : m2;
}

/* ------------------------------------------------------------------------
 * One launch of the Lift `glb-sdp` kernel (example/<algo>/kernel5.json:3)
 * restated over CSR instead of the padded ELLPACK buffers:
 *   tmp[W-1-i] = mult(x_or_identity(idx[i]), val[i])      (map_seq)
 *   acc = identity; for j in 0..W-1: acc = add(acc, tmp[j]) (reduce_seq)
 *   out[row] = epilogue(acc, alpha, y[row], beta)
 * i.e. the row is reduced in REVERSE stored order, the W-len padded slots
 * (idx = -1 -> identity, val = identity) coming first; those contribute
 * add(identity, mult(identity, identity)) which leaves acc unchanged for all
 * three semirings (0+0*0; min(FLT_MAX, inf); 0||0), so they are skipped.
 * An index < 0 or >= vlength substitutes the identity for x[idx].
 * Elements are 4 bytes: float for semirings 0/1, int32 for semiring 2.
 * ---------------------------------------------------------------------- */
int oracle_kernel(int semiring, int32_t rows, const int32_t *row_ptr,
                  const int32_t *col_idx, const void *val_, const void *x_,
                  const void *y_, const void *alpha_, const void *beta_,
                  int32_t vlength, void *out_) {
  if (semiring == SR_MAX_MIN_I32) {   /* identity INT_MIN (app/scc.cpp:206); pads: max(acc, min(MIN,MIN)) = acc */
    const int32_t *val = val_, *x = x_, *y = y_;
    int32_t alpha = *(const int32_t *)alpha_, beta = *(const int32_t *)beta_;
    int32_t *out = out_;
    for (int32_t r = 0; r < rows; r++) {
      int32_t acc = INT32_MIN;
      for (int32_t j = row_ptr[r + 1] - 1; j >= row_ptr[r]; j--) {
        int32_t c = col_idx[j];
        int32_t xv = (c < 0 || c >= vlength) ? INT32_MIN : x[c];
        acc = mm_add(acc, mm_mult(xv, val[j]));
      }
      out[r] = mm_epi(acc, alpha, y[r], beta);
    }
    return 0;
  }
  if (semiring == SR_OR_AND_I32) {
    const int32_t *val = val_, *x = x_, *y = y_;
    int32_t alpha = *(const int32_t *)alpha_, beta = *(const int32_t *)beta_;
    int32_t *out = out_;
    for (int32_t r = 0; r < rows; r++) {
      int32_t acc = 0;
      for (int32_t j = row_ptr[r + 1] - 1; j >= row_ptr[r]; j--) {
        int32_t c = col_idx[j];
        int32_t xv = (c < 0 || c >= vlength) ? 0 : x[c];
        acc = oa_add(acc, oa_mult(xv, val[j]));
      }
      out[r] = oa_epi(acc, alpha, y[r], beta);
    }
    return 0;
  }
  const float *val = val_, *x = x_, *y = y_;
  float alpha = *(const float *)alpha_, beta = *(const float *)beta_;
  float *out = out_;
  if (semiring == SR_PLUS_TIMES_F32) {
    for (int32_t r = 0; r < rows; r++) {
      float acc = 0.0f;
      for (int32_t j = row_ptr[r + 1] - 1; j >= row_ptr[r]; j--) {
        int32_t c = col_idx[j];
        float xv = (c < 0 || c >= vlength) ? 0.0f : x[c];
        acc = pt_add(acc, pt_mult(xv, val[j]));
      }
      out[r] = pt_epi(acc, alpha, y[r], beta);
    }
    return 0;
  }
  if (semiring == SR_MIN_PLUS_F32) {
    const float ident = 3.4028235E38f;
    for (int32_t r = 0; r < rows; r++) {
      float acc = ident;
      for (int32_t j = row_ptr[r + 1] - 1; j >= row_ptr[r]; j--) {
        int32_t c = col_idx[j];
        float xv = (c < 0 || c >= vlength) ? ident : x[c];
        acc = mp_add(acc, mp_mult(xv, val[j]));
      }
      out[r] = mp_epi(acc, alpha, y[r], beta);
    }
    return 0;
  }
  return -1;
}

/* ------------------------------------------------------------------------
 * Iteration driver of the iterative apps, restating
 * HarnessSSSP::executeRun + should_terminate_iteration
 * (app/sssp.cpp:97-155,157-176) and the identical BFS pair
 * (app/bfs.cpp:94-152,154-174):
 *   do { launch(in, y, out); terminate = compare(in, out);
 *        swap(in, out); y = in; } while (!terminate)
 * First launch uses the caller's y0; afterwards y aliases the input
 * (setGlobalArg(3, input_mem_ptr), app/sssp.cpp:150).
 * Termination: float semirings |in[i]-out[i]| < delta for all i
 * (app/sssp.cpp:170, app/pr.cpp:170), int semirings in[i] == out[i]
 * (app/bfs.cpp:167, app/scc.cpp:166).
 * `iters` counts launches including the confirming one.  max_iters bounds
 * graphs on which the reference would spin forever (TODO.md:7-8).
 * x0 is overwritten with the final vector (the buffer the reference's
 * `input` pointer designates after the last swap).  scratch: rows elements.
 * ---------------------------------------------------------------------- */
int oracle_iterate(int semiring, int32_t rows, const int32_t *row_ptr,
                   const int32_t *col_idx, const void *val, void *x0,
                   const void *y0, void *scratch, const void *alpha,
                   const void *beta, double delta, int32_t max_iters,
                   int32_t *iters_out, int32_t *converged_out) {
  void *in = x0, *out = scratch;
  const void *y = y0;
  int32_t it = 0;
  int term = 0;
  do {
    int rc = oracle_kernel(semiring, rows, row_ptr, col_idx, val, in, y, alpha,
                           beta, rows, out);
    if (rc)
      return rc;
    int equal = 1;
    if (semiring == SR_OR_AND_I32 || semiring == SR_MAX_MIN_I32) {
      const int32_t *a = in, *b = out;
      for (int32_t i = 0; equal && i < rows; i++)
        equal = a[i] == b[i];
    } else {
      const float *a = in, *b = out;
      for (int32_t i = 0; equal && i < rows; i++)
        equal = fabs(a[i] - b[i]) < delta;
    }
    term = equal;
    void *t = in; in = out; out = t;
    y = in;
    it++;
  } while (!term && it < max_iters);
  if (in != x0)
    memcpy(x0, in, (size_t)rows * 4);
  *iters_out = it;
  *converged_out = term;
  return 0;
}

/* Harness::check_result (inc/harness.h:113-147): exact != compare of the
 * first gold_len elements; returns 0 CORRECT, 1 NOT_CHECKED, 3 BAD_LENGTH,
 * 4 BAD_VALUES (enum Correctness, inc/sql_stat.h:7-15). */
int oracle_check_result_f32(const float *gold, int64_t gold_len,
                            const float *res, int64_t res_len) {
  if (gold_len == 0)
    return 1;
  if (res_len < gold_len)
    return 3;
  int errors = 0;
  for (int64_t i = 0; i < gold_len; i++)
    if (gold[i] != res[i] && ++errors == 20)
      break;
  return errors ? 4 : 0;
}
