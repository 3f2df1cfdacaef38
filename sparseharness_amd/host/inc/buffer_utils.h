// buffer_utils.h -- typed vector <-> byte buffer helpers (reference:
// inc/buffer_utils.h:77-88 `enchar`).  The whole-matrix debug dumps of the
// reference (printc_vec / print_rsa_matrix, quirk A-9) are not reproduced.
#pragma once
#include <cstring>
#include <vector>

template <typename T> std::vector<char> enchar(const std::vector<T> &v) {
  std::vector<char> out(v.size() * sizeof(T));
  if (!v.empty())
    std::memcpy(out.data(), v.data(), out.size());
  return out;
}

template <typename T> std::vector<T> dechar(const std::vector<char> &v) {
  std::vector<T> out(v.size() / sizeof(T));
  if (!out.empty())
    std::memcpy(out.data(), v.data(), out.size() * sizeof(T));
  return out;
}
