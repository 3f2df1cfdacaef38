// iterative_app.h -- shared body of `sssp_harness`, `bfs_harness`, `pr_harness` and `scc_harness`.
// Restates the flow of the reference's app/sssp.cpp:44-155 (and its twin
// app/bfs.cpp:44-152): per trial, iterate  out = kernel(in, y)  until
// should_terminate_iteration(in, out); swap in/out; y := in; then append the
// per-trial MEDIAN_RESULT and MULTI_ITERATION_SUM rows and reset the inputs.
// Two loop drivers:
//   * device loop (default): IterativeHarness::iterateOnDevice -- convergence
//     test fused into the kernel, one flag word per iteration crosses PCIe;
//   * host loop (--host_loop / SH_HOST_LOOP=1): literally the reference's
//     do/while with a full read-back and the app's
//     should_terminate_iteration on the host, kept for A/B parity runs.
// Both are capped at max_iters (the reference spins forever on graphs with
// cycles and no diagonal, TODO.md:7-8).
#pragma once
#include <algorithm>
#include <cstdlib>
#include <numeric>

#include "common.h"
#include "csv_utils.h"
#include "harness.h"
#include "kernel_config.h"
#include "options.h"
#include "sparse_matrix.h"
#include "vector_generator.h"

template <typename SemiRingType>
class HarnessIterativeApp : public IterativeHarness<std::vector<SqlStat>, SemiRingType> {
  using Base = IterativeHarness<std::vector<SqlStat>, SemiRingType>;

public:
  HarnessIterativeApp(std::string &kernel_source, unsigned int platform, unsigned int device,
                      ArgContainer<SemiRingType> args, unsigned int trials, std::chrono::milliseconds timeout,
                      double delta, unsigned int max_iters, bool host_loop)
      : Base(kernel_source, platform, device, args, trials, timeout, delta), _max_iters(max_iters),
        _host_loop(host_loop) {
    this->allocateBuffers();
  }

  std::vector<std::vector<SqlStat>> benchmark(Run run, std::vector<SemiRingType> &gold) override {
    start_timer(benchmark, HarnessIterativeApp);
    std::vector<std::vector<SqlStat>> runtimes;
    for (unsigned int t = 0; t < this->_trials; t++) {
      start_timer(benchmark_iteration, HarnessIterativeApp);
      std::vector<SqlStat> run_runtimes = executeRun(run, t, gold);
      if (!run_runtimes.empty()) {
        std::sort(run_runtimes.begin(), run_runtimes.end(), SqlStat::compare);
        std::chrono::nanoseconds median_time = run_runtimes[run_runtimes.size() / 2].getTime();
        run_runtimes.push_back(SqlStat(median_time, NOT_CHECKED, run.global1, run.local1, MEDIAN_RESULT, t));
        // as the reference: the sum runs over every row pushed so far, median row included
        std::chrono::nanoseconds total_time(0);
        for (auto &s : run_runtimes)
          total_time += s.getTime();
        run_runtimes.push_back(SqlStat(total_time, NOT_CHECKED, run.global1, run.local1, MULTI_ITERATION_SUM));
        runtimes.push_back(run_runtimes);
      }
      // the final vector of this trial is kept on the host, then -- after EVERY trial, as the reference does
      // (app/sssp.cpp:88-90) -- the inputs go back to their initial state: the next trial and the next run of
      // a multi-line run-file start from x0 / y0 again
      {
        std::vector<char> bytes(this->_args.x_vect.size());
        this->readFromGlobalArg(bytes, _final_mem);
        _final_host = dechar<SemiRingType>(bytes);
      }
      this->resetInputs();
    }
    return runtimes;
  }

  int lastIterations() const { return _last_iters; }
  bool lastConverged() const { return _last_converged; }
  // final vector of the last trial (host copy taken before the inputs were reset)
  std::vector<SemiRingType> finalVector() { return _final_host; }

protected:
  std::vector<SqlStat> executeRun(Run run, unsigned int trial, std::vector<SemiRingType> &) override {
    start_timer(executeRun, HarnessIterativeApp);
    return _host_loop ? hostLoop(run, trial) : deviceLoop(run, trial);
  }

private:
  std::vector<SqlStat> deviceLoop(Run run, unsigned int trial) {
    std::vector<uint64_t> per_iter;
    this->iterateOnDevice(_max_iters, _last_iters, _last_converged, per_iter);
    _final_mem = this->_mem_manager._x_vect;
    std::vector<SqlStat> runtimes;
    for (std::size_t i = 0; i < per_iter.size(); i++)
      runtimes.push_back(SqlStat(std::chrono::nanoseconds(per_iter[i]), NOT_CHECKED, run.global1, run.local1,
                                 RAW_RESULT, trial, (unsigned int)i));
    return runtimes;
  }

  std::vector<SqlStat> hostLoop(Run run, unsigned int trial) {
    auto &mm = this->_mem_manager;
    std::vector<SqlStat> runtimes;
    device_mem *input_mem_ptr = &mm._x_vect, *output_mem_ptr = &mm._output;
    std::vector<char> *input_host_ptr = &mm._input_host_buffer, *output_host_ptr = &mm._output_host_buffer;
    bool should_terminate = false;
    unsigned int iteration = 0;
    do {
      std::copy(output_host_ptr->begin(), output_host_ptr->end(), mm._temp_out_buffer.begin());
      this->resetTempBuffers();
      auto time = this->executeKernel(run);
      runtimes.push_back(SqlStat(time, NOT_CHECKED, run.global1, run.local1, RAW_RESULT, trial, iteration));
      this->readFromGlobalArg(*output_host_ptr, *output_mem_ptr);
      assertBuffersNotEqual(*output_host_ptr, mm._temp_out_buffer);
      should_terminate = this->should_terminate_iteration(*input_host_ptr, *output_host_ptr);
      std::swap(input_mem_ptr, output_mem_ptr);
      std::swap(input_host_ptr, output_host_ptr);
      this->setGlobalArg((int)mm._input_idx, input_mem_ptr);
      this->setGlobalArg((int)mm._output_idx, output_mem_ptr);
      this->setGlobalArg(3, input_mem_ptr); // y := x
      iteration++;
    } while (!should_terminate && iteration < _max_iters);
    _last_iters = (int)iteration;
    _last_converged = should_terminate;
    _final_mem = *input_mem_ptr;
    return runtimes;
  }

  unsigned int _max_iters;
  bool _host_loop;
  int _last_iters = 0;
  bool _last_converged = false;
  device_mem _final_mem = nullptr;
  std::vector<SemiRingType> _final_host;
};

inline bool env_host_loop() {
  const char *e = std::getenv("SH_HOST_LOOP");
  return e && e[0] == '1';
}

// Shared main(): `App` supplies the element type, the harness subclass and the
// algorithm constants (initial vectors, alpha, beta, padding zero), each as a
// function of the loaded matrix (PageRank's depend on its height), plus two
// hooks: beforeLoad() runs ahead of the matrix load, normalise(matrix) right
// before the encoding, where app/pr.cpp:199 and app/scc.cpp:217 call theirs.
template <typename App> int iterative_main(int argc, char *argv[]) {
  using T = typename App::SemiRingType;
  App::beforeLoad();
  COMMON_MAIN_PREAMBLE(T)
  auto x = App::initialX(matrix);
  auto y = App::initialY(matrix);
  unsigned long max_alloc = deviceGetMaxAllocSize(opt_platform->get(), opt_device->get());
  std::cout << "Got max alloc: " << max_alloc << "\n";
  ArgContainer<T> args;
  try {
    App::normalise(matrix);
    args = executorEncodeMatrix(max_alloc, kernel, matrix, App::zero(), x, y, App::alpha(matrix), App::beta(matrix));
  } catch (unsigned long attempted_alloc_size) {
    LOG_ERROR("Attempted to allocate: ", attempted_alloc_size, " bytes, but this device's max is ", max_alloc);
    return 1;
  }
  typename App::HarnessType harness(kernel.getSource(), opt_platform->get(), opt_device->get(), args,
                                    opt_trials->get(), std::chrono::milliseconds(opt_timeout->get()),
                                    opt_float_delta->get(), opt_max_iters->get(), env_host_loop());
  std::vector<T> gold(0, 0); // the reference has no gold for the iterative apps (app/sssp.cpp:243)
  const std::string &kernel_name = kernel.getName();
  const std::string device_name = harness.getDeviceName();
  for (auto run : runs) {
    start_timer(run_iteration, main);
    std::cout << "Benchmarking run: " << run << ENDL;
    auto runtimes = harness.benchmark(run, gold);
    for (auto &statList : runtimes)
      std::cout << SqlStat::makeSqlCommand(statList, kernel_name, hostname, device_name, matrix_name, experiment)
                << "\n";
    auto fin = harness.finalVector();
    std::cout << "SH_RESULT iterations=" << harness.lastIterations() << " converged=" << harness.lastConverged()
              << " " << App::summarise(fin) << "\n";
  }
  return 0;
}
