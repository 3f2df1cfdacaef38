// pr.cpp -- `pr_harness`: PageRank as an iterated (+,x) SpMV on floats with
// y := x (reference: app/pr.cpp).  Constants as the reference: damping 0.85
// (:185), x0 = 1/N (:188), y0 = 1 (:189), alpha = 1, beta = (1-d)/N (:191-192),
// padding zero = 0, matrix.pagerank_normalise(d, 0) before the encoding (:199),
// terminate when |in[i] - out[i]| < delta for every i (:157-177).
//
// Reference behaviour worth knowing: the normalised
// weights are < 1 and the row builder narrows every value through `int`, so
// with the default (reference-faithful) truncation the matrix the kernel sees
// is all zeros for any positively weighted graph and the loop converges to 0.
// SH_NO_TRUNCATE=1 runs the PageRank the app was meant to be.
#include <cmath>
#include <iostream>
#include <sstream>

#include "common.h"
#include "csv_utils.h"
#include "iterative_app.h"
#include "kernel_config.h"
#include "options.h"
#include "sparse_matrix.h"
#include "vector_generator.h"

class HarnessPR : public HarnessIterativeApp<float> {
public:
  using HarnessIterativeApp<float>::HarnessIterativeApp;

protected:
  bool should_terminate_iteration(std::vector<char> &input, std::vector<char> &output) override {
    start_timer(should_terminate_iteration, HarnessPR);
    const float *in = reinterpret_cast<const float *>(input.data());
    const float *out = reinterpret_cast<const float *>(output.data());
    const std::size_t n = std::min(input.size(), output.size()) / sizeof(float);
    bool equal = true;
    for (std::size_t i = 0; equal && i < n; i++)
      equal = std::fabs(in[i] - out[i]) < _delta;
    return equal;
  }
};

struct PrApp {
  using SemiRingType = float;
  using HarnessType = HarnessPR;
  static constexpr float dampingFactor = 0.85f;
  static void beforeLoad() { SparseMatrix<float>::set_keep_entries(true); }
  static void normalise(SparseMatrix<float> &m) { m.pagerank_normalise(dampingFactor, 0.0f); }
  static ConstXVectorGenerator<float> initialX(SparseMatrix<float> &m) {
    return ConstXVectorGenerator<float>(1.0f / (float)m.height());
  }
  static ConstYVectorGenerator<float> initialY(SparseMatrix<float> &) { return ConstYVectorGenerator<float>(1.0f); }
  static float alpha(SparseMatrix<float> &) { return 1.0f; }
  static float beta(SparseMatrix<float> &m) { return (1.0f - dampingFactor) / (float)m.height(); }
  static float zero() { return 0.0f; }
  static std::string summarise(const std::vector<float> &r) {
    double sum = 0;
    float top = 0;
    std::size_t arg = 0;
    for (std::size_t i = 0; i < r.size(); i++) {
      sum += r[i];
      if (r[i] > top) { top = r[i]; arg = i; }
    }
    std::ostringstream o;
    o.precision(9);
    o << "rank_sum=" << sum << " top_vertex=" << arg << " top_rank=" << top;
    return o.str();
  }
};

int main(int argc, char *argv[]) { return iterative_main<PrApp>(argc, argv); }
