// common.h -- shared CLI preamble of the apps (reference: inc/common.h:5-56).
// Same flag letters and defaults:
//   -p platform (ignored: HIP has none)  -d device (HIP ordinal)  -i trials (10)
//   -m matrix  -f matrix_name  -k kernel  -r runfile  -n hostname  -e experiment
//   -c delta (1e-4)  -t timeout ms (100)
// plus two additions: --max_iters (bound for non-terminating BFS graphs,
// TODO.md:7-8) and --gold_only (config 1 of BASELINE.json: CPU gold path,
// no GPU touched).  Non-square matrices exit(2) as in the reference (:49-52).
#pragma once
#define ENDL "\n"

#define COMMON_MAIN_PREAMBLE(mtype)                                                            \
  start_timer(main, global);                                                                   \
  OptParser op("Harness for SPMV sparse matrix dense vector multiplication benchmarks");       \
  auto opt_platform = op.addOption<unsigned>({'p', "platform", "ignored (kept for CLI compatibility).", 0}); \
  auto opt_device = op.addOption<unsigned>({'d', "device", "HIP device ordinal (default 0).", 0});           \
  auto opt_trials = op.addOption<unsigned>({'i', "trials", "Execute each kernel 'trials' times (default 10).", 10}); \
  auto opt_matrix_file = op.addOption<std::string>({'m', "matrix", "Input matrix"});           \
  auto opt_matrix_name = op.addOption<std::string>({'f', "matrix_name", "Input matrix name"}); \
  auto opt_kernel_file = op.addOption<std::string>({'k', "kernel", "Input kernel"});           \
  auto opt_run_file = op.addOption<std::string>({'r', "runfile", "Run configuration file"});   \
  auto opt_host_name = op.addOption<std::string>({'n', "hostname", "Host the harness is running on"}); \
  auto opt_experiment_id = op.addOption<std::string>({'e', "experiment", "An experiment ID for data reporting"}); \
  auto opt_float_delta = op.addOption<double>({'c', "delta", "Delta for floating point comparisons", 0.0001}); \
  auto opt_timeout = op.addOption<unsigned int>({'t', "timeout", "Timeout to avoid multiple executions (default 100ms).", 100}); \
  auto opt_max_iters = op.addOption<unsigned int>({'x', "max_iters", "Iteration cap for the iterative apps (default 10000).", 10000}); \
  auto opt_gold_only = op.addOption<bool>({'g', "gold_only", "Compute and report the CPU gold only; never touch a GPU.", false}); \
  op.parse(argc, argv);                                                                        \
  using namespace std;                                                                         \
  const std::string matrix_filename = opt_matrix_file->require();                              \
  const std::string matrix_name = opt_matrix_name->require();                                  \
  const std::string kernel_filename = opt_kernel_file->require();                              \
  const std::string runs_filename = opt_run_file->require();                                   \
  const std::string hostname = opt_host_name->require();                                       \
  const std::string experiment = opt_experiment_id->require();                                 \
  std::cerr << "matrix_filename " << matrix_filename << ENDL;                                  \
  std::cerr << "kernel_filename " << kernel_filename << ENDL;                                  \
  SparseMatrix<mtype> matrix(matrix_filename);                                                 \
  KernelConfig<mtype> kernel(kernel_filename);                                                 \
  auto csvlines = CSV::load_csv(runs_filename);                                                \
  std::vector<Run> runs;                                                                       \
  for (auto &l : csvlines)                                                                     \
    runs.push_back(Run(l));                                                                    \
  (void)opt_platform; (void)opt_max_iters; (void)opt_gold_only;                                \
  if (matrix.height() != matrix.width()) {                                                     \
    std::cout << "Matrix is not square. Failing computation." << ENDL;                         \
    std::cerr << "Matrix is not square. Failing computation." << ENDL;                         \
    std::exit(2);                                                                              \
  } else {                                                                                     \
    std::cout << " Matrix is square - width = " << matrix.width() << " and height = " << matrix.height() << "\n"; \
  }
