// csv_utils.h -- run-file reader (reference: inc/csv_utils.h:16-49).
// Same behaviour: comma split (a trailing comma yields no extra token),
// reading stops at the first empty line or EOF.
#pragma once
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace CSV {
using csv_token = std::string;
using csv_line = std::vector<csv_token>;

inline csv_line tokenise_line(const std::string &line) {
  csv_line out;
  std::stringstream ss(line);
  for (std::string cell; std::getline(ss, cell, ',');)
    out.push_back(cell);
  return out;
}

inline std::size_t read_size_t(const csv_token &tok) {
  std::size_t v = 0;
  std::stringstream(tok) >> v;
  return v;
}

inline std::vector<csv_line> load_csv(const std::string &filename) {
  std::vector<csv_line> lines;
  std::ifstream in(filename);
  for (std::string line; std::getline(in, line);) {
    csv_line toks = tokenise_line(line);
    if (toks.empty())
      break;
    lines.push_back(toks);
  }
  return lines;
}
} // namespace CSV
