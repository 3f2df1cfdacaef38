// sssp.cpp -- `sssp_harness`: single-source shortest paths as an iterated
// (min,+) SpMV on floats (reference: app/sssp.cpp).  Constants as the
// reference: x0 = y0 = (0 at vertex 0, FLT_MAX elsewhere) (:186-192),
// alpha = beta = 0 (:219-220), padding zero = FLT_MAX (:231), terminate when
// |in[i] - out[i]| < delta for every i (:157-176).
#include <cmath>
#include <iostream>
#include <limits>
#include <sstream>

#include "common.h"
#include "csv_utils.h"
#include "iterative_app.h"
#include "kernel_config.h"
#include "options.h"
#include "sparse_matrix.h"
#include "vector_generator.h"

class HarnessSSSP : public HarnessIterativeApp<float> {
public:
  using HarnessIterativeApp<float>::HarnessIterativeApp;

protected:
  bool should_terminate_iteration(std::vector<char> &input, std::vector<char> &output) override {
    start_timer(should_terminate_iteration, HarnessSSSP);
    const float *in = reinterpret_cast<const float *>(input.data());
    const float *out = reinterpret_cast<const float *>(output.data());
    const std::size_t n = std::min(input.size(), output.size()) / sizeof(float);
    bool equal = true;
    for (std::size_t i = 0; equal && i < n; i++)
      equal = std::fabs(in[i] - out[i]) < _delta;
    return equal;
  }
};

struct SsspApp {
  using SemiRingType = float;
  using HarnessType = HarnessSSSP;
  static void beforeLoad() {}
  static void normalise(SparseMatrix<float> &) {}
  static InitialDistancesGeneratorX<float> initialX(SparseMatrix<float> &) { return {0.0f, std::numeric_limits<float>::max()}; }
  static InitialDistancesGeneratorY<float> initialY(SparseMatrix<float> &) { return {0.0f, std::numeric_limits<float>::max()}; }
  static float alpha(SparseMatrix<float> &) { return 0.0f; }
  static float beta(SparseMatrix<float> &) { return 0.0f; }
  static float zero() { return std::numeric_limits<float>::max(); }
  static std::string summarise(const std::vector<float> &d) {
    std::size_t reached = 0;
    double sum = 0;
    for (float v : d)
      if (v < std::numeric_limits<float>::max()) { reached++; sum += v; }
    std::ostringstream o;
    o << "reached=" << reached << " distance_sum=" << sum;
    return o.str();
  }
};

int main(int argc, char *argv[]) { return iterative_main<SsspApp>(argc, argv); }
