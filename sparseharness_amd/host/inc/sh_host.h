/* sh_host.h -- C ABI of the host-side helpers (no HIP): synthetic matrix
 * generators and the MatrixMarket -> CSR loader used by the apps, the Python
 * binding and bench.py.  Built into sparseharness_amd/libsparseharness_host.so. */
#ifndef SH_HOST_H_
#define SH_HOST_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Seeded generators.  Caller allocates row_ptr[rows+1],
 * col_idx[nnz], val[nnz].  Return 0, -1 bad argument, -2 int32 overflow. */
int sh_synth_powerlaw(int64_t rows, int64_t cols, int64_t nnz, double exponent, int64_t dmax,
                      uint64_t seed, int32_t *row_ptr, int32_t *col_idx, float *val);
int sh_synth_rmat(int scale, int edge_factor, double a, double b, double c, uint64_t seed,
                  int permute, int32_t *row_ptr, int32_t *col_idx, float *val);

/* MatrixMarket -> CSR with the reference's semantics (SparseMatrix<T>,
 * src/sparse_matrix.cpp:11-119): see host/inc/sparse_matrix.h.  elem_is_int
 * selects SparseMatrix<int> (BFS) instead of SparseMatrix<float>.
 * truncate_values = 1 reproduces the reference's int narrowing (quirk A-3).
 * Returns 0 or the exit code the reference would have used (negative). */
typedef struct sh_host_csr {
  int32_t rows, cols, header_nnz;
  int64_t nnz;
  int32_t *row_ptr;
  int32_t *col_idx;
  void *val; /* float[nnz] or int32[nnz] */
} sh_host_csr;
int sh_mm_load(const char *path, int elem_is_int, int truncate_values, sh_host_csr *out);
/* As sh_mm_load, then the app's matrix normaliser before the narrowing: normalise = 0 none,
 * 1 SparseMatrix::pagerank_normalise(damping, 0) (app/pr.cpp:199), 2 scc_normalise()
 * (app/scc.cpp:217). */
int sh_mm_load_ex(const char *path, int elem_is_int, int truncate_values, int normalise, double damping,
                  sh_host_csr *out);
void sh_host_csr_release(sh_host_csr *m);

#ifdef __cplusplus
}
#endif
#endif
