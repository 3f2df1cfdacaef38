// bfs.cpp -- `bfs_harness`: breadth-first frontier propagation as an iterated
// (or,and) SpMV on int32 (reference: app/bfs.cpp).  Constants as the
// reference: x0 = y0 = (1 at vertex 0, 0 elsewhere) (:184-190), alpha = 1,
// beta = 0 (:215-216), padding zero = 0, terminate on exact equality of
// consecutive vectors (:154-174).
#include <iostream>
#include <sstream>

#include "common.h"
#include "csv_utils.h"
#include "iterative_app.h"
#include "kernel_config.h"
#include "options.h"
#include "sparse_matrix.h"
#include "vector_generator.h"

class HarnessBFS : public HarnessIterativeApp<int> {
public:
  using HarnessIterativeApp<int>::HarnessIterativeApp;

protected:
  bool should_terminate_iteration(std::vector<char> &input, std::vector<char> &output) override {
    start_timer(should_terminate_iteration, HarnessBFS);
    const int *in = reinterpret_cast<const int *>(input.data());
    const int *out = reinterpret_cast<const int *>(output.data());
    const std::size_t n = std::min(input.size(), output.size()) / sizeof(int);
    bool equal = true;
    for (std::size_t i = 0; equal && i < n; i++)
      equal = in[i] == out[i];
    return equal;
  }
};

struct BfsApp {
  using SemiRingType = int;
  using HarnessType = HarnessBFS;
  static void beforeLoad() {}
  static void normalise(SparseMatrix<int> &) {}
  static InitialDistancesGeneratorX<int> initialX(SparseMatrix<int> &) { return {1, 0}; }
  static InitialDistancesGeneratorY<int> initialY(SparseMatrix<int> &) { return {1, 0}; }
  static int alpha(SparseMatrix<int> &) { return 1; }
  static int beta(SparseMatrix<int> &) { return 0; }
  static int zero() { return 0; }
  static std::string summarise(const std::vector<int> &f) {
    std::size_t set = 0;
    for (int v : f)
      set += v != 0;
    std::ostringstream o;
    o << "set=" << set;
    return o.str();
  }
};

int main(int argc, char *argv[]) { return iterative_main<BfsApp>(argc, argv); }
