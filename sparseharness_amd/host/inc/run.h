// run.h -- launch geometry of one benchmark run (reference: inc/run.h:9-32,
// src/run.cpp:4-16): six size_t fields global1..3 / local1..3 read from one
// run-file line.  The native kernels take their grid from the matrix schedule;
// Run is carried through to sh_launch and into the SQL row (global1/local1).
#pragma once
#include <cassert>
#include <cstddef>
#include <ostream>

#include "csv_utils.h"

class Run {
public:
  Run(std::size_t g1, std::size_t g2, std::size_t g3, std::size_t l1, std::size_t l2, std::size_t l3)
      : global1(g1), global2(g2), global3(g3), local1(l1), local2(l2), local3(l3) {}
  explicit Run(const CSV::csv_line &line) {
    assert(line.size() == 6 && "Bad CSV format");
    std::size_t *dst[6] = {&global1, &global2, &global3, &local1, &local2, &local3};
    for (int i = 0; i < 6; i++)
      *dst[i] = CSV::read_size_t(line[i]);
  }
  std::size_t num_work_items() const { return local1 * local2 * local3; }

  std::size_t global1 = 0, global2 = 0, global3 = 0;
  std::size_t local1 = 0, local2 = 0, local3 = 0;
};

inline std::ostream &operator<<(std::ostream &os, const Run &r) {
  return os << "{" << r.global1 << " " << r.global2 << " " << r.global3 << " / " << r.local1 << " "
            << r.local2 << " " << r.local3 << "}";
}
