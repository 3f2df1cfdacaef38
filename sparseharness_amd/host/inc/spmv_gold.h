// spmv_gold.h -- the in-harness CPU gold that check_result compares against
// (reference: inc/spmv_gold.h:9-28).  Same expression, same order:
//   acc = zero; for each stored (col, val) of row i, in stored order:
//     acc += (alpha * (x.get(col) * val)) + (beta * y.get(val));
// including the reference's quirk A-4 (beta*y added per non-zero, y indexed
// by the VALUE) so that results are interchangeable.  Runs over the CSR
// arrays directly (no by-value matrix copy, no row-of-pairs rebuild).
// This is the harness's own checker, as in the reference; it is not a compute
// fallback: the apps only use it to label results correct/badvalues.
#pragma once
#include <vector>

#include "csds_timer.h"
#include "sparse_matrix.h"
#include "vector_generator.h"

template <typename T> class Gold {
public:
  static std::vector<T> spmv(SparseMatrix<T> &A, XVectorGenerator<T> &x, YVectorGenerator<T> &y, T alpha, T beta,
                             T zero) {
    start_timer(spmv, gold);
    const auto &rp = A.rowPtr();
    const auto &ci = A.colIdx();
    const auto &va = A.values();
    const int n = A.height();
    std::vector<T> xv = x.generate(A.width());
    std::vector<T> result((std::size_t)n, 0);
    for (int i = 0; i < n; i++) {
      T acc = zero;
      for (int32_t j = rp[(std::size_t)i]; j < rp[(std::size_t)i + 1]; j++) {
        const T v = va[(std::size_t)j];
        acc += (alpha * (xv[(std::size_t)ci[(std::size_t)j]] * v)) + (beta * y.get((int)v));
      }
      result[(std::size_t)i] = acc;
    }
    return result;
  }

private:
  Gold() {}
};
