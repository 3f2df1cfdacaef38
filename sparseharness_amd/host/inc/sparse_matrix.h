// sparse_matrix.h -- MatrixMarket file -> row structure -> device layout.
// Mirrors SparseMatrix<EType> / CL_matrix of the reference
// (inc/sparse_matrix.h:23-93, src/sparse_matrix.cpp) with the same loading
// semantics:
//   * entry (I, J, v) of the file lands in ROW J, COLUMN I (y = A^T x);
//   * rows keep file order, duplicates are kept, symmetric off-diagonal
//     entries are mirrored right behind their original;
//   * values are narrowed through `int` (src/sparse_matrix.cpp:107) unless
//     truncation is switched off (SH_NO_TRUNCATE=1 or set_truncate(false)).
// What is native: rows are held as CSR (row_ptr / col_idx / val), not as a
// vector of vectors of pairs, and cl_encode() emits that CSR as the device
// layout instead of padded ELLPACK / RSA byte buffers.
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "csds_timer.h"
#include "logger.h"

// Device-ready encoding handed to executorEncodeMatrix.  `indices`/`values`
// keep the reference's member names (inc/sparse_matrix.h:23-33); row_ptr is
// the CSR addition.  cl_height is the (possibly padded) row count, cl_width
// the longest row (informational for CSR).
class CL_matrix {
public:
  std::vector<char> indices; // int32 col_idx[nnz]
  std::vector<char> values;  // T val[nnz]
  std::vector<char> row_ptr; // int32 row_ptr[cl_height + 1]
  int cl_width = 0;
  int cl_height = 0;
};

template <typename EType> class SparseMatrix {
public:
  template <typename T> using ellpack_row = std::vector<std::pair<int, T>>;
  template <typename T> using ellpack_matrix = std::vector<ellpack_row<T>>;

  explicit SparseMatrix(std::string filename);
  // In-memory construction (synthetic benchmarks): takes a finished CSR.
  SparseMatrix(int rows, int cols, std::vector<int32_t> row_ptr, std::vector<int32_t> col_idx,
               std::vector<EType> val);

  // Native device layout.  pad_height reproduces quirk A-6: the height grows
  // to H + (m - H % m) even when already aligned; padded rows are empty.
  // pad_width / rsa / width_pad_modulo describe ELLPACK/RSA layouts and do not
  // apply to CSR (accepted, ignored).  Throws `unsigned long` (the byte count)
  // when the index array exceeds device_max_alloc_bytes, as the reference
  // does (src/sparse_matrix.cpp:231-233).
  CL_matrix cl_encode(unsigned long device_max_alloc_bytes, EType zero, bool pad_height, bool pad_width,
                      bool rsa, int height_pad_modulo, int width_pad_modulo);

  // Row-of-pairs view for code written against the reference's gold
  // (inc/spmv_gold.h:13); built lazily from the CSR, O(nnz) extra memory.
  ellpack_matrix<EType> &ellpack_encode();

  int height() const { return rows; }
  int width() const { return cols; }
  int nonZeros() const { return nonz; }           // header count, as the reference
  int64_t storedNonZeros() const { return (int64_t)col_idx_.size(); }
  unsigned int maxRowLength() const { return max_width; }

  const std::vector<int32_t> &rowPtr() const { return row_ptr_; }
  const std::vector<int32_t> &colIdx() const { return col_idx_; }
  const std::vector<EType> &values() const { return val_; }

  static void set_truncate(bool on) { truncate_flag() = on; }
  static bool truncate() { return truncate_flag(); }

  // The apps' matrix normalisers (inc/sparse_matrix.h:60-62 of the reference).  Both rewrite the
  // tuple values BEFORE the int narrowing of calculate_ellpack, as the reference does because it
  // normalises nz_entries and builds the rows afterwards (app/pr.cpp:199, app/scc.cpp:217).
  //   pagerank_normalise: val = (|val| / column_sums[I]) * damping, sums taken over the tuples in
  //     file order in T arithmetic (src/sparse_matrix.cpp:409-431).  Needs the un-narrowed values
  //     and the file order: load the matrix with set_keep_entries(true).
  //   scc_normalise: val = (I == J) ? numeric_limits<T>::min() : J   (:433-456).
  void pagerank_normalise(float dampingFactor, EType zero);
  void scc_normalise();
  // Keep what pagerank_normalise needs (8 bytes per entry); off by default.
  static void set_keep_entries(bool on) { keep_flag() = on; }

private:
  void load_from_file(const std::string &filename);
  // binary CSR cache next to the file (SH_CSR_CACHE=1 | <dir>), validated by the file's size + mtime
  bool load_from_cache(const std::string &filename);
  void store_to_cache(const std::string &filename) const;
  static bool &truncate_flag();
  static bool &keep_flag();

  int rows = 0, cols = 0, nonz = 0;
  unsigned int max_width = 0;
  std::vector<int32_t> row_ptr_, col_idx_;
  std::vector<EType> val_;
  std::vector<EType> raw_val_;      // static_cast<T>(file value) in CSR order   } only with
  std::vector<int32_t> file_order_; // CSR position of the k-th tuple            } keep_entries
  ellpack_matrix<EType> ellpack_cache_;
  bool ellpack_built_ = false;
};
