// spmv.cpp -- `spmv_harness`: float (+,x) SpMV benchmark app.
// Same CLI, flow and output as the reference's app/spmv.cpp:43-169: x = 1,
// y = 0, alpha = 1, beta = 0; per run `trials` timed launches, each followed
// by a blocking read-back and an exact compare against the CPU gold; median
// appended; one SQL INSERT per run.  --gold_only runs the CPU gold path alone
// (BASELINE.json config 1: plumbing, no GPU).
#include <algorithm>
#include <iostream>
#include <numeric>

#include "common.h"
#include "csds_timer.h"
#include "csv_utils.h"
#include "harness.h"
#include "kernel_config.h"
#include "kernel_utils.h"
#include "options.h"
#include "run.h"
#include "sparse_matrix.h"
#include "spmv_gold.h"
#include "vector_generator.h"

class HarnessSPMV : public Harness<SqlStat, float> {
public:
  HarnessSPMV(std::string &kernel_source, unsigned int platform, unsigned int device, ArgContainer<float> args,
              unsigned int trials, std::chrono::milliseconds timeout, double delta)
      : Harness(kernel_source, platform, device, args, trials, timeout, delta) {
    allocateBuffers();
  }

  std::vector<SqlStat> benchmark(Run run, std::vector<float> &gold) override {
    start_timer(benchmark, HarnessSPMV);
    std::vector<SqlStat> runtimes;
    for (unsigned int t = 0; t < _trials; t++) {
      start_timer(benchmark_iteration, HarnessSPMV);
      resetTempBuffers();
      SqlStat stat = executeRun(run, t, gold);
      runtimes.push_back(stat);
      if (stat.getTime() > _timeout)
        break;
      lowerTimeout(stat.getTime());
      assertBuffersNotEqual(_mem_manager._output_host_buffer, _mem_manager._temp_out_buffer);
    }
    std::sort(runtimes.begin(), runtimes.end(), SqlStat::compare);
    std::chrono::nanoseconds median_time = runtimes[runtimes.size() / 2].getTime();
    runtimes.push_back(SqlStat(median_time, STATISTIC_VALUE, run.global1, run.local1, MEDIAN_RESULT));
    return runtimes;
  }

private:
  SqlStat executeRun(Run run, unsigned int, std::vector<float> &gold) override {
    std::chrono::nanoseconds time = executeKernel(run);
    readFromGlobalArg(_mem_manager._output_host_buffer, _mem_manager._output);
    return SqlStat(time, check_result(gold), run.global1, run.local1, RAW_RESULT);
  }
};

int main(int argc, char *argv[]) {
  COMMON_MAIN_PREAMBLE(float)

  ConstXVectorGenerator<float> x(1.0f);
  ConstYVectorGenerator<float> y(0);
  float alpha = 1.0f;
  float beta = 0.0f;

  if (opt_gold_only->get()) {
    auto gold = Gold<float>::spmv(matrix, x, y, alpha, beta, 0.0f);
    double sum = std::accumulate(gold.begin(), gold.end(), 0.0);
    const auto old_precision = std::cout.precision(17);
    std::cout << "GOLD rows=" << gold.size() << " sum=" << sum << " head=";
    for (std::size_t i = 0; i < gold.size() && i < 5; i++)
      std::cout << (i ? "," : "") << gold[i];
    std::cout << "\n";
    std::cout.precision(old_precision);
    return 0;
  }

  unsigned long max_alloc = deviceGetMaxAllocSize(opt_platform->get(), opt_device->get());
  std::cout << "Got max alloc: " << max_alloc << "\n";

  ArgContainer<float> args;
  try {
    args = executorEncodeMatrix(max_alloc, kernel, matrix, 0.0f, x, y, alpha, beta);
  } catch (unsigned long attempted_alloc_size) {
    // the reference logs and carries on with empty args (quirk A-8); stop instead
    LOG_ERROR("Attempted to allocate: ", attempted_alloc_size, " bytes, but this device's max is ", max_alloc);
    return 1;
  }

  HarnessSPMV harness(kernel.getSource(), opt_platform->get(), opt_device->get(), args, opt_trials->get(),
                      std::chrono::milliseconds(opt_timeout->get()), opt_float_delta->get());

  auto gold = Gold<float>::spmv(matrix, x, y, alpha, beta, 0.0f);

  const std::string &kernel_name = kernel.getName();
  const std::string device_name = harness.getDeviceName();
  for (auto run : runs) {
    start_timer(run_iteration, main);
    std::cout << "Benchmarking run: " << run << ENDL;
    std::vector<SqlStat> runtimes = harness.benchmark(run, gold);
    std::cout << "runtimes: [";
    for (auto &time : runtimes)
      std::cout << "\n\t" << time.printStat(kernel_name, hostname, device_name, matrix_name, experiment);
    std::cout << "\n]" << ENDL;
    std::cout << SqlStat::makeSqlCommand(runtimes, kernel_name, hostname, device_name, matrix_name, experiment) << "\n";
    // native extra, on its own line so the INSERT stays byte-compatible
    const double ms = (double)runtimes.back().getTime().count() / 1e6;
    const double nnz = (double)matrix.storedNonZeros();
    const double bytes = 8.0 * nnz + 4.0 * (matrix.height() + 1) + 4.0 * matrix.width() + 4.0 * matrix.height();
    std::cout << "SH_PERF median_ms=" << ms << " gflops=" << 2.0 * nnz / (ms * 1e6) << " algorithmic_GBps="
              << bytes / (ms * 1e6) << "\n";
  }
  return 0;
}
