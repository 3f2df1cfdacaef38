// host_selftest.cpp -- CPU-only checks of the host mirror's small pieces against the behaviour
// the reference documents (file:line cited per block).  Run by tests/test_host.py.
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <type_traits>

#include "arithexpr_evaluator.h"
#include "csv_utils.h"
#include "kernel_config.h"
#include "kernel_utils.h"
#include "run.h"
#include "sparse_matrix.h"
#include "spmv_gold.h"
#include "sql_stat.h"
#include "vector_generator.h"

static int failures = 0;
#define CHECK(cond)                                                              \
  do {                                                                           \
    if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); failures++; } \
  } while (0)

// `host_selftest --encode <matrix.mtx> <kernel.json>...`: load each kernel config and run
// executorEncodeMatrix on the matrix (no GPU involved); one line per config.  tests/test_host.py
// points it at the reference's own example/*/kernel*.json files to show that they are accepted as they are.
template <typename T> static int encode_probe(const std::string &matrix, int n, char **jsons) {
  SparseMatrix<T> m{matrix};
  ConstXVectorGenerator<T> x((T)1);
  ConstYVectorGenerator<T> y((T)0);
  for (int i = 0; i < n; i++) {
    KernelConfig<T> kc{std::string(jsons[i])};
    auto args = executorEncodeMatrix(1ul << 30, kc, m, (T)0, x, y, (T)1, (T)0);
    const std::string &src = kc.getSource();
    const char *sr = std::is_integral<T>::value
                         ? ((src.find("int_max") != std::string::npos || src.find("doubleMinMax") != std::string::npos) ? "max-min" : "or-and")
                         : ((src.find("clmin") != std::string::npos || src.find("absadd") != std::string::npos) ? "min-plus" : "plus-times");
    std::printf("ENCODE %s name=%s semiring=%s rows=%d cols=%d size_args=", jsons[i], kc.getName().c_str(), sr, args.rows, args.cols);
    for (unsigned v : args.size_args) std::printf("%u,", v);
    std::printf(" output=%u temp_globals=%zu temp_locals=%zu x=%zu idxs=%zu\n", args.output, args.temp_globals.size(),
                args.temp_locals.size(), args.x_vect.size(), args.m_idxs.size());
  }
  return 0;
}

int main(int argc, char **argv) {
  if (argc >= 4 && (std::string(argv[1]) == "--encode" || std::string(argv[1]) == "--encode-int"))
    return std::string(argv[1]) == "--encode" ? encode_probe<float>(argv[2], argc - 3, argv + 3)
                                              : encode_probe<int>(argv[2], argc - 3, argv + 3);
  const std::string dir = argc > 1 ? argv[1] : ".";
  // ---- run-file: 6 fields, trailing comma tolerated, stop at first blank line (inc/csv_utils.h:16-49, src/run.cpp:4-16)
  {
    std::ofstream(dir + "/runs.csv") << "1280,1,1,128,1,1,\n524288,1,1,256,1,1\n\n99,9,9,9,9,9\n";
    auto lines = CSV::load_csv(dir + "/runs.csv");
    CHECK(lines.size() == 2);
    Run r0(lines[0]), r1(lines[1]);
    CHECK(r0.global1 == 1280 && r0.local1 == 128 && r0.num_work_items() == 128);
    CHECK(r1.global1 == 524288 && r1.local1 == 256 && r1.global3 == 1);
    std::ostringstream os;
    os << r0;
    CHECK(os.str() == "{1280 1 1 / 128 1 1}");
  }
  // ---- size expressions of the kernel JSON (src/arithexpr_evaluator.cpp:10-30)
  CHECK(Evaluator::evaluate("(4*v_MHeight_2*v_MWidthC_1)", 18, 1138, 1138) == 4 * 1138 * 18);
  CHECK(Evaluator::evaluate("(4*v_VLength_3)", 1, 2, 300) == 1200);
  CHECK(Evaluator::evaluate("4", 1, 2, 3) == 4);
  CHECK(Evaluator::evaluate("(8+(4*v_MWidthC_1))/2", 6, 0, 0) == 16);
  CHECK(Evaluator::evaluate("?", 1, 2, 3) == 0);          // quirk A-13: non-numeric sizes
  CHECK(Evaluator::evaluate("(4*v_Unknown)", 1, 2, 3) == 0);
  // ---- SQL row text, byte for byte (inc/sql_stat.h:28-79)
  {
    std::vector<SqlStat> st{SqlStat(std::chrono::nanoseconds(1500000), CORRECT, 1280, 128, RAW_RESULT, 2, 3),
                            SqlStat(std::chrono::nanoseconds(250), STATISTIC_VALUE, 1280, 128, MEDIAN_RESULT)};
    const std::string sql = SqlStat::makeSqlCommand(st, "k", "h", "d", "m", "e");
    CHECK(sql == "INSERT INTO table_name (time, correct, kernel, global, local, host, device, matrix, iteration, "
                 "trial,statistic, experiment_id) VALUES (1.5, \"correct\", \"k\", 1280, 128, \"h\", \"d\", \"m\",3,2,"
                 "\"RAW_RESULT\", \"e\"), (0.00025, \"statisticvalue\", \"k\", 1280, 128, \"h\", \"d\", \"m\",0,0,"
                 "\"MEDIAN_RESULT\", \"e\");");
    CHECK(SqlStat::compare(st[1], st[0]) && !SqlStat::compare(st[0], st[1]));
  }
  // ---- kernel config JSON: optional properties default to "nothing"/-1, numbers-as-strings (src/kernel_config.cpp:8-96)
  {
    std::ofstream(dir + "/k.json") << R"JSON({"name":"awrg-alcl-alcl-edp-split-8","source":"float clmin(float a,float b){}\n\"quoted\" \\ A",
      "properties":{"splitSize":"8","innerMap":"alcl","outerMap":"awrg","chunkSize":128,"dotProduct":"earlyexit","extra":true},
      "inputArgs":[{"variable":"v1","addressSpace":"global","size":"?"}],
      "tempGlobals":[{"variable":"t","addressSpace":"global","size":"4"}],
      "outputArg":{"variable":"o","addressSpace":"global","size":"(4*v_MHeight_2)"},
      "tempLocals":[{"variable":"l","addressSpace":"local","size":"(4*v_MWidthC_1)"}],
      "paramVars":["MHeight","MWidthC","VLength"],"outputSize":"(4*v_MHeight_2)","unknownKey":[1,2,{"a":null}]})JSON";
    KernelConfig<float> kc(dir + "/k.json");
    auto p = kc.getProperties();
    CHECK(kc.getName() == "awrg-alcl-alcl-edp-split-8");
    CHECK(p.splitSize == 8 && p.chunkSize == 128 && p.outerMap == "awrg" && p.innerMap2 == "nothing" && p.arrayType == "nothing");
    CHECK(kc.getSource().find("clmin") != std::string::npos && kc.getSource().find("\"quoted\" \\ A") != std::string::npos);
    CHECK(kc.getTempLocals().size() == 1 && kc.getParamVars().size() == 3 && kc.getOutputArg()->size == "(4*v_MHeight_2)");
    CHECK(kc.getArgs()[0].size == "?");
  }
  // ---- loader + encode + gold on a tiny symmetric file, incl. quirks A-1, A-3, A-6
  {
    std::ofstream(dir + "/m.mtx") << "%%MatrixMarket matrix coordinate real symmetric\n% c\n3 3 3\n1 1 2.9\n3 1 -1.5\n2 3 4\n";
    SparseMatrix<float> m(dir + "/m.mtx");
    CHECK(m.height() == 3 && m.width() == 3 && m.nonZeros() == 3 && m.storedNonZeros() == 5);
    // row = file column: row0 = {(0,2),(2,-1)}, row1 = {(2,4)}, row2 = {(0,-1),(1,4)} with values narrowed through int
    CHECK(m.rowPtr() == (std::vector<int32_t>{0, 2, 3, 5}));
    CHECK(m.colIdx() == (std::vector<int32_t>{0, 2, 2, 0, 1}));
    CHECK(m.values() == (std::vector<float>{2.f, -1.f, 4.f, -1.f, 4.f}));
    ConstXVectorGenerator<float> x(1.0f);
    ConstYVectorGenerator<float> y(0.0f);
    auto gold = Gold<float>::spmv(m, x, y, 1.0f, 0.0f, 0.0f);
    CHECK(gold == (std::vector<float>{1.f, 4.f, 3.f}));
    KernelConfig<float> kc(dir + "/k.json");   // chunkSize 128 -> height 3 + (128 - 3 % 128) = 128
    auto args = executorEncodeMatrix(1ul << 30, kc, m, 0.0f, x, y, 1.0f, 0.0f);
    CHECK(args.rows == 128 && args.cols == 128 && args.output == 4 * 128);
    CHECK(args.x_vect.size() == 4 * 128 && args.m_row_ptr.size() == 4 * 129 && args.m_idxs.size() == 4 * 5);
    CHECK(args.size_args == (std::vector<unsigned>{128, 1, 128}));   // MHeight, MWidthC = (2 + (8 - 2 % 8)) / 8, VLength
    CHECK(args.temp_globals == (std::vector<unsigned>{4}) && args.temp_locals == (std::vector<unsigned>{4}));
    bool threw = false;
    try { (void)executorEncodeMatrix(8ul, kc, m, 0.0f, x, y); } catch (unsigned long n) { threw = n == 20; }
    CHECK(threw);                                                    // oversize -> throw unsigned long (src/sparse_matrix.cpp:231-233)
  }
  std::printf(failures ? "host selftest: %d FAILED\n" : "host selftest: all passed\n", failures);
  return failures ? 1 : 0;
}
