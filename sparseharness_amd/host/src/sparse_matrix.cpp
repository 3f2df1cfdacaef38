// sparse_matrix.cpp -- see inc/sparse_matrix.h.  Written from scratch: one
// read of the file into memory, a hand-rolled token scanner instead of one
// fscanf per line (the reference's wall-clock bottleneck),
// and a counting sort straight into CSR instead of per-row push_back.
#include "sparse_matrix.h"

#include <algorithm>
#include <cctype>
#include <sys/stat.h>
#include <unistd.h>
#include <type_traits>
#include <cmath>
#include <limits>
#include <stdexcept>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "sh_host.h"
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

struct Banner {
  bool ok = false, coordinate = false, real = false, integer = false, pattern = false, symmetric = false;
  std::string typecode = "    ";
};

std::string lowered(std::string s) {
  for (auto &c : s)
    c = (char)std::tolower((unsigned char)c);
  return s;
}

// First line: "%%MatrixMarket matrix <coordinate|array> <real|complex|pattern|integer>
// <general|symmetric|hermitian|skew-symmetric>" (reference: src/mmio.cpp:92-165).
Banner parse_banner(const std::string &line) {
  Banner b;
  char w[5][65];
  if (std::sscanf(line.c_str(), "%64s %64s %64s %64s %64s", w[0], w[1], w[2], w[3], w[4]) != 5)
    return b;
  if (std::strncmp(w[0], "%%MatrixMarket", 14) != 0 || lowered(w[1]) != "matrix")
    return b;
  const std::string crd = lowered(w[2]), dt = lowered(w[3]), st = lowered(w[4]);
  b.typecode[0] = 'M';
  if (crd == "coordinate") { b.coordinate = true; b.typecode[1] = 'C'; }
  else if (crd == "array") b.typecode[1] = 'A';
  else return b;
  if (dt == "real") { b.real = true; b.typecode[2] = 'R'; }
  else if (dt == "integer") { b.integer = true; b.typecode[2] = 'I'; }
  else if (dt == "pattern") { b.pattern = true; b.typecode[2] = 'P'; }
  else if (dt == "complex") b.typecode[2] = 'C';
  else return b;
  if (st == "general") b.typecode[3] = 'G';
  else if (st == "symmetric") { b.symmetric = true; b.typecode[3] = 'S'; }
  else if (st == "hermitian") b.typecode[3] = 'H';
  else if (st == "skew-symmetric") b.typecode[3] = 'K';
  else return b;
  b.ok = true;
  return b;
}

// Token scanner over the in-memory file.  Integers and the common decimal forms are converted
// by hand (no libc call, no locale: strtod does not scale across threads here); the decimal
// fast path is Clinger's exact case -- at most 15 significant digits and |exponent| <= 22, so
// the mantissa and the power of ten are both exact doubles and one multiply or divide gives the
// correctly rounded result, i.e. bit-for-bit what strtod / the reference's fscanf("%lg") return.
// Everything else (long mantissas, huge exponents, inf/nan, hex floats) falls back to strtod.
struct Scanner {
  const char *p, *end;
  static bool ws(char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }
  void skip_ws() { while (p < end && ws(*p)) ++p; }
  bool next_int(int &v) {
    skip_ws();
    if (p >= end) return false;
    const char *q = p;
    bool neg = false;
    if (*q == '-' || *q == '+') { neg = *q == '-'; ++q; }
    if (q >= end || *q < '0' || *q > '9') return false;
    long long acc = 0;
    int digits = 0;
    while (q < end && *q >= '0' && *q <= '9') {
      if (++digits > 18) { char *e; long t = std::strtol(p, &e, 10); p = e; v = (int)t; return true; }
      acc = acc * 10 + (*q++ - '0');
    }
    p = q;
    v = (int)(neg ? -acc : acc);
    return true;
  }
  bool next_double(double &v) {
    skip_ws();
    if (p >= end) return false;
    static const double p10[] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11,
                                 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    const char *q = p;
    bool neg = false;
    if (*q == '-' || *q == '+') { neg = *q == '-'; ++q; }
    unsigned long long mant = 0;
    int sig = 0, frac = 0;
    bool any = false, lead = true;
    while (q < end && *q >= '0' && *q <= '9') {
      any = true;
      if (!(lead && *q == '0')) { lead = false; if (++sig <= 19) mant = mant * 10 + (unsigned)(*q - '0'); }
      ++q;
    }
    if (q < end && *q == '.') {
      ++q;
      while (q < end && *q >= '0' && *q <= '9') {
        any = true;
        if (!(lead && *q == '0')) { lead = false; if (++sig <= 19) mant = mant * 10 + (unsigned)(*q - '0'); }
        ++frac;
        ++q;
      }
    }
    int ex = 0;
    bool fast = any && sig <= 15;
    if (fast && q < end && (*q == 'e' || *q == 'E')) {
      const char *r = q + 1;
      bool eneg = false;
      if (r < end && (*r == '-' || *r == '+')) { eneg = *r == '-'; ++r; }
      if (r < end && *r >= '0' && *r <= '9') {
        int edig = 0;
        while (r < end && *r >= '0' && *r <= '9') { if (++edig > 4) { fast = false; break; } ex = ex * 10 + (*r++ - '0'); }
        if (eneg) ex = -ex;
        q = r;
      }
    }
    // a following letter (hex float, "inf", "nan", 'd' exponents ...) is strtod's business
    if (fast && q < end && ((*q >= 'a' && *q <= 'z') || (*q >= 'A' && *q <= 'Z'))) fast = false;
    const int e10 = ex - frac;
    if (fast && e10 >= -22 && e10 <= 22) {
      double d = (double)mant;   // exact: < 10^15 < 2^53
      d = e10 >= 0 ? d * p10[e10] : d / p10[-e10];
      v = neg ? -d : d;
      p = q;
      return true;
    }
    char *e;
    v = std::strtod(p, &e);
    if (e == p) return false;
    p = e;
    return true;
  }
  std::string next_line() {
    const char *s = p;
    while (p < end && *p != '\n') ++p;
    std::string l(s, p);
    if (p < end) ++p;
    return l;
  }
};

// tuple value: static_cast<T>(val) (src/sparse_matrix.cpp:59)
template <typename T> inline T tuple_value(double v) { return static_cast<T>(v); }
// row value: the tuple value pushed through `int val` (:107, quirk A-3)
inline float narrow(float f, bool truncate) { return truncate ? static_cast<float>(static_cast<int>(f)) : f; }
inline int narrow(int i, bool) { return i; }

} // namespace

template <typename T> bool &SparseMatrix<T>::truncate_flag() {
  static bool flag = [] {
    const char *e = std::getenv("SH_NO_TRUNCATE");
    return !(e && e[0] == '1');
  }();
  return flag;
}

template <typename T> bool &SparseMatrix<T>::keep_flag() {
  static bool flag = false;
  return flag;
}

// ---- binary CSR cache --------------------------------------------------------
// SH_CSR_CACHE=1 keeps the finished rows next to the matrix file as <file>.shcsr.<f32|i32>[.raw];
// SH_CSR_CACHE=<dir> keeps them in that directory.  A cache file is used only if its header
// matches the source file's size and mtime, the element type and the truncation mode; anything
// else (or any short read) falls back to parsing, and the cache is rewritten.  Not used when the
// file-order entries are kept (PageRank needs them).
namespace {
struct CacheHeader {
  char magic[8];            // "SHCSR\0\1\0"
  int32_t elem_is_int, truncated;
  int64_t src_size, src_mtime_ns;
  int32_t rows, cols, nonz;
  uint32_t max_width;
  int64_t nnz;
};
const char CACHE_MAGIC[8] = {'S', 'H', 'C', 'S', 'R', 0, 1, 0};

bool source_stamp(const std::string &path, int64_t &size, int64_t &mtime_ns) {
  struct stat st;
  if (::stat(path.c_str(), &st) != 0) return false;
  size = (int64_t)st.st_size;
  mtime_ns = (int64_t)st.st_mtim.tv_sec * 1000000000ll + st.st_mtim.tv_nsec;
  return true;
}

std::string cache_path(const std::string &filename, bool is_int, bool truncated) {
  const char *e = std::getenv("SH_CSR_CACHE");
  if (!e || !e[0] || (e[0] == '0' && !e[1])) return "";
  std::string suffix = std::string(".shcsr.") + (is_int ? "i32" : "f32") + (truncated ? "" : ".raw");
  if (e[0] == '1' && !e[1]) return filename + suffix;
  std::string base = filename.substr(filename.find_last_of('/') == std::string::npos ? 0 : filename.find_last_of('/') + 1);
  return std::string(e) + "/" + base + suffix;
}
} // namespace

template <typename T> bool SparseMatrix<T>::load_from_cache(const std::string &filename) {
  const std::string cp = cache_path(filename, std::is_integral<T>::value, truncate());
  if (cp.empty() || keep_flag()) return false;
  int64_t size = 0, mtime = 0;
  if (!source_stamp(filename, size, mtime)) return false;
  FILE *f = std::fopen(cp.c_str(), "rb");
  if (!f) return false;
  CacheHeader h;
  bool ok = std::fread(&h, sizeof h, 1, f) == 1 && std::memcmp(h.magic, CACHE_MAGIC, 8) == 0 &&
            h.elem_is_int == (int)std::is_integral<T>::value && h.truncated == (int)truncate() && h.src_size == size &&
            h.src_mtime_ns == mtime && h.rows >= 0 && h.nnz >= 0;
  if (ok) {
    row_ptr_.resize((std::size_t)h.rows + 1);
    col_idx_.resize((std::size_t)h.nnz);
    val_.resize((std::size_t)h.nnz);
    ok = std::fread(row_ptr_.data(), 4, row_ptr_.size(), f) == row_ptr_.size() &&
         std::fread(col_idx_.data(), 4, col_idx_.size(), f) == col_idx_.size() &&
         std::fread(val_.data(), 4, val_.size(), f) == val_.size() && row_ptr_[0] == 0 &&
         row_ptr_[(std::size_t)h.rows] == h.nnz;
  }
  std::fclose(f);
  if (!ok) {
    row_ptr_.clear(); col_idx_.clear(); val_.clear();
    return false;
  }
  rows = h.rows; cols = h.cols; nonz = h.nonz; max_width = h.max_width;
  LOG_DEBUG("rows read from the binary cache ", cp);
  return true;
}

template <typename T> void SparseMatrix<T>::store_to_cache(const std::string &filename) const {
  const std::string cp = cache_path(filename, std::is_integral<T>::value, truncate());
  if (cp.empty() || keep_flag()) return;
  CacheHeader h{};
  std::memcpy(h.magic, CACHE_MAGIC, 8);
  h.elem_is_int = (int)std::is_integral<T>::value; h.truncated = (int)truncate();
  if (!source_stamp(filename, h.src_size, h.src_mtime_ns)) return;
  h.rows = rows; h.cols = cols; h.nonz = nonz; h.max_width = max_width; h.nnz = (int64_t)col_idx_.size();
  const std::string tmp = cp + ".tmp" + std::to_string((long)::getpid());
  FILE *f = std::fopen(tmp.c_str(), "wb");
  if (!f) return;   // read-only dataset folder: just no cache
  const bool ok = std::fwrite(&h, sizeof h, 1, f) == 1 &&
                  std::fwrite(row_ptr_.data(), 4, row_ptr_.size(), f) == row_ptr_.size() &&
                  std::fwrite(col_idx_.data(), 4, col_idx_.size(), f) == col_idx_.size() &&
                  std::fwrite(val_.data(), 4, val_.size(), f) == val_.size();
  if (std::fclose(f) != 0 || !ok || std::rename(tmp.c_str(), cp.c_str()) != 0)
    std::remove(tmp.c_str());
}

// "synth:powerlaw:<rows>:<nnz>", "synth:rmat:<scale>", "synth:rmat-unpermuted:<scale>", "synth:scircuit" in place
// of a file name: the seeded generators of SURVEY.md 8d (synth.cpp), so that BASELINE.json's synthetic configs
// run through the same apps -- Harness<>::benchmark -> executeKernel -- as a MatrixMarket file does.
template <typename T> static bool synthetic_spec(const std::string &spec, int &rows, int &cols, std::vector<int32_t> &rp,
                                                 std::vector<int32_t> &ci, std::vector<T> &va) {
  if (spec.compare(0, 6, "synth:") != 0)
    return false;
  std::vector<std::string> part;
  for (std::size_t a = 6; a <= spec.size();) {
    const std::size_t b = std::min(spec.find(':', a), spec.size());
    part.push_back(spec.substr(a, b - a));
    a = b + 1;
  }
  int64_t n = 0, nnz = 0;
  int rc = -1;
  std::vector<float> fv;
  auto alloc = [&]() { rp.assign((std::size_t)n + 1, 0); ci.assign((std::size_t)nnz, 0); fv.assign((std::size_t)nnz, 0.f); };
  if (part.size() == 3 && part[0] == "powerlaw") {
    n = std::atoll(part[1].c_str()); nnz = std::atoll(part[2].c_str());
    if (n > 0 && nnz > 0 && n < INT32_MAX && nnz < INT32_MAX) {
      alloc();
      rc = sh_synth_powerlaw(n, n, nnz, 2.1, std::min<int64_t>(1000000, n), 0x5EED1000ull, rp.data(), ci.data(), fv.data());
    }
  } else if (part.size() == 2 && (part[0] == "rmat" || part[0] == "rmat-unpermuted")) {
    const int scale = std::atoi(part[1].c_str());
    if (scale > 0 && scale < 27) {
      n = (int64_t)1 << scale; nnz = 16 * n;
      alloc();
      rc = sh_synth_rmat(scale, 16, 0.57, 0.19, 0.19, 0x5EED0023ull, part[0] == "rmat", rp.data(), ci.data(), fv.data());
    }
  } else if (part.size() == 1 && part[0] == "scircuit") {
    n = 170998; nnz = 958936;
    alloc();
    rc = sh_synth_powerlaw(n, n, nnz, 2.1, 353, 0x5EED5C1Cull, rp.data(), ci.data(), fv.data());
  }
  if (rc != 0) {
    std::cerr << "Bad synthetic matrix spec " << spec << " (synth:powerlaw:<rows>:<nnz> | synth:rmat:<scale> | synth:scircuit)" << ENDL;
    std::exit(-1);
  }
  rows = cols = (int)n;
  va.resize(fv.size());
  for (std::size_t k = 0; k < fv.size(); k++)
    va[k] = (T)fv[k];
  return true;
}

template <typename T> SparseMatrix<T>::SparseMatrix(std::string filename) {
  if (synthetic_spec<T>(filename, rows, cols, row_ptr_, col_idx_, val_)) {
    start_timer(synthetic_matrix, SparseMatrix);
    nonz = (int)col_idx_.size();
    for (int i = 0; i < rows; i++)
      max_width = std::max<unsigned>(max_width, (unsigned)(row_ptr_[i + 1] - row_ptr_[i]));
    if (keep_flag()) {   // (a generated matrix has no file: CSR order stands in for file order)
      raw_val_ = val_;
      file_order_.resize(col_idx_.size());
      for (std::size_t k = 0; k < file_order_.size(); k++)
        file_order_[k] = (int32_t)k;
    }
    return;
  }
  if (load_from_cache(filename))
    return;
  load_from_file(filename);
  if (rows > 0)
    store_to_cache(filename);
}

template <typename T>
SparseMatrix<T>::SparseMatrix(int r, int c, std::vector<int32_t> rp, std::vector<int32_t> ci, std::vector<T> va)
    : rows(r), cols(c), nonz((int)ci.size()), row_ptr_(std::move(rp)), col_idx_(std::move(ci)), val_(std::move(va)) {
  for (int i = 0; i < rows; i++)
    max_width = std::max<unsigned>(max_width, (unsigned)(row_ptr_[i + 1] - row_ptr_[i]));
  if (keep_flag()) { // a generated matrix has no file: CSR order stands in for file order
    raw_val_ = val_;
    file_order_.resize(col_idx_.size());
    for (std::size_t k = 0; k < file_order_.size(); k++)
      file_order_[k] = (int32_t)k;
  }
}

template <typename T> void SparseMatrix<T>::load_from_file(const std::string &filename) {
  start_timer(load_from_file, SparseMatrix);
  FILE *f = std::fopen(filename.c_str(), "rb");
  if (!f) {
    std::cerr << "Failed to open matrix file " << filename << ENDL;
    std::exit(-1);
  }
  std::fseek(f, 0, SEEK_END);
  long sz = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  std::vector<char> buf((std::size_t)sz + 1);
  if (sz > 0 && std::fread(buf.data(), 1, (std::size_t)sz, f) != (std::size_t)sz) {
    std::cerr << "Failed to read matrix file " << filename << ENDL;
    std::exit(-1);
  }
  std::fclose(f);
  buf[(std::size_t)sz] = '\0';
  Scanner sc{buf.data(), buf.data() + sz};

  Banner b = parse_banner(sc.next_line());
  if (!b.ok) {
    std::cerr << "Could not read matrix market banner" << ENDL;
    std::exit(-1);
  }
  std::cerr << "Matcode: " << b.typecode << ENDL;
  if (!(b.coordinate && (b.real || b.integer || b.pattern))) {
    std::cerr << "Cannot process this matrix type. Typecode: " << b.typecode << ENDL;
    std::exit(-1);
  }
  // size line: first non-comment line (reference: src/mmio.cpp:174-199)
  std::string line;
  do {
    if (sc.p >= sc.end) {
      std::cerr << "Cannot read matrix sizes and number of non-zeros" << ENDL;
      return;
    }
    line = sc.next_line();
  } while (!line.empty() && line[0] == '%');
  if (std::sscanf(line.c_str(), "%d %d %d", &rows, &cols, &nonz) != 3) {
    if (!(sc.next_int(rows) && sc.next_int(cols) && sc.next_int(nonz))) {
      std::cerr << "Cannot read matrix sizes and number of non-zeros" << ENDL;
      rows = cols = nonz = 0;
      return;
    }
  }
  std::cerr << "Rows " << rows << " cols " << cols << " non-zeros " << nonz << ENDL;

  const bool trunc = truncate();
  std::vector<int32_t> ei, ej;
  std::vector<T> ev;
  const std::size_t cap = (std::size_t)nonz * (b.symmetric ? 2 : 1);
  ei.reserve(cap); ej.reserve(cap); ev.reserve(cap);
  auto emit = [&](int I, int J, double v) {
    --I; --J;
    const T tv = tuple_value<T>(v);
    ei.push_back(I); ej.push_back(J); ev.push_back(tv);
    if (b.symmetric && I != J) {
      ei.push_back(J); ej.push_back(I); ev.push_back(tv);
    }
  };
  // Fast path: the body is cut at line boundaries into one slice per thread and
  // tokenised in parallel, then stitched together in file order.  It assumes what every
  // MatrixMarket writer produces -- one entry per line -- and is only kept when it finds
  // exactly `nonz` entries; otherwise the sequential token scanner below, which like the
  // reference's fscanf loop (src/sparse_matrix.cpp:49-63) does not care about line structure,
  // redoes the job.
  bool parsed = false;
  const std::size_t body = (std::size_t)(sc.end - sc.p);
  if (nonz >= 50000 && body > (1u << 20)) {
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = std::max(1, omp_get_max_threads());
#endif
    nthreads = (int)std::min<std::size_t>((std::size_t)nthreads, body / (256u << 10) + 1);
    std::vector<const char *> cut((std::size_t)nthreads + 1);
    cut[0] = sc.p;
    cut[(std::size_t)nthreads] = sc.end;
    for (int t = 1; t < nthreads; t++) {
      const char *q = sc.p + body * (std::size_t)t / (std::size_t)nthreads;
      while (q < sc.end && *q != '\n') ++q;
      cut[(std::size_t)t] = q < sc.end ? q + 1 : sc.end;
    }
    // one cache-line-aligned slot per thread: the vectors' end pointers and `out` are written for
    // every entry, and unaligned neighbours made the threads fight over cache lines
    struct alignas(128) Part { std::vector<int32_t> I, J; std::vector<T> V; std::size_t out = 0; bool ok = true; };
    std::vector<Part> parts((std::size_t)nthreads);
#pragma omp parallel for schedule(static, 1) num_threads(nthreads)
    for (int t = 0; t < nthreads; t++) {
      Part &pt = parts[(std::size_t)t];
      Scanner ls{cut[(std::size_t)t], cut[(std::size_t)t + 1]};
      std::size_t lines = 0;
      for (const char *q = ls.p; q < ls.end; ++q)
        lines += *q == '\n';
      pt.I.reserve(lines + 1); pt.J.reserve(lines + 1); pt.V.reserve(lines + 1);
      for (;;) {
        ls.skip_ws();
        if (ls.p >= ls.end) break;
        int I = 0, J = 0;
        double v = 1.0;
        if (!ls.next_int(I) || !ls.next_int(J) || (!b.pattern && !ls.next_double(v))) { pt.ok = false; break; }
        // the rest of the line must be blank: anything else means tokens are not one entry per line
        while (ls.p < ls.end && *ls.p != '\n') {
          if (!Scanner::ws(*ls.p)) { pt.ok = false; break; }
          ++ls.p;
        }
        if (!pt.ok) break;
        pt.I.push_back(I - 1); pt.J.push_back(J - 1); pt.V.push_back(tuple_value<T>(v));
        pt.out += (b.symmetric && I != J) ? 2 : 1;
      }
    }
    std::size_t total = 0, total_out = 0;
    bool ok = true;
    for (auto &pt : parts) { ok = ok && pt.ok; total += pt.I.size(); total_out += pt.out; }
    if (ok && total == (std::size_t)nonz) {
      ei.resize(total_out); ej.resize(total_out); ev.resize(total_out);
      std::vector<std::size_t> base((std::size_t)nthreads + 1, 0);
      for (int t = 0; t < nthreads; t++) base[(std::size_t)t + 1] = base[(std::size_t)t] + parts[(std::size_t)t].out;
#pragma omp parallel for schedule(static, 1) num_threads(nthreads)
      for (int t = 0; t < nthreads; t++) {
        const Part &pt = parts[(std::size_t)t];
        std::size_t o = base[(std::size_t)t];
        for (std::size_t k = 0; k < pt.I.size(); k++) {
          ei[o] = pt.I[k]; ej[o] = pt.J[k]; ev[o] = pt.V[k]; o++;
          if (b.symmetric && pt.I[k] != pt.J[k]) { ei[o] = pt.J[k]; ej[o] = pt.I[k]; ev[o] = pt.V[k]; o++; }
        }
      }
      parsed = true;
    }
    LOG_DEBUG("parallel MatrixMarket parse on ", nthreads, " threads: ", parsed ? "ok" : "fell back to the sequential scanner");
  }
  if (!parsed) {
    for (int k = 0; k < nonz; k++) {
      int I = 0, J = 0;
      double v = 1.0;
      sc.next_int(I);
      sc.next_int(J);
      if (!b.pattern)
        sc.next_double(v);
      emit(I, J, v);
    }
  }
  // counting sort by row (= file column J), stable => file order inside a row
  start_timer(calculate_ellpack, sparse_matrix);
  row_ptr_.assign((std::size_t)rows + 1, 0);
  for (std::size_t k = 0; k < ej.size(); k++) {
    if (ej[k] < 0 || ej[k] >= rows) {
      std::cerr << "Matrix entry " << k << " has column " << ej[k] + 1 << " outside the matrix" << ENDL;
      std::exit(-1);
    }
    row_ptr_[(std::size_t)ej[k] + 1]++;
  }
  for (int r = 0; r < rows; r++) {
    max_width = std::max<unsigned>(max_width, (unsigned)row_ptr_[(std::size_t)r + 1]);
    row_ptr_[(std::size_t)r + 1] += row_ptr_[(std::size_t)r];
  }
  col_idx_.resize(ej.size());
  val_.resize(ej.size());
  std::vector<int32_t> cursor(row_ptr_.begin(), row_ptr_.end() - 1);
  for (std::size_t k = 0; k < ej.size(); k++) {
    const int32_t pos = cursor[(std::size_t)ej[k]]++;
    col_idx_[(std::size_t)pos] = ei[k];
    val_[(std::size_t)pos] = narrow(ev[k], trunc);
  }
  if (keep_flag()) {
    raw_val_.resize(ej.size());
    file_order_.resize(ej.size());
    std::vector<int32_t> cur2(row_ptr_.begin(), row_ptr_.end() - 1);
    for (std::size_t k = 0; k < ej.size(); k++) {
      const int32_t pos = cur2[(std::size_t)ej[k]]++;
      raw_val_[(std::size_t)pos] = ev[k];
      file_order_[k] = pos;
    }
  }
  LOG_DEBUG("max width: ", max_width);
}

template <typename T>
CL_matrix SparseMatrix<T>::cl_encode(unsigned long device_max_alloc_bytes, T zero, bool pad_height, bool pad_width,
                                     bool rsa, int height_pad_modulo, int width_pad_modulo) {
  start_timer(cl_encode, sparse_matrix);
  (void)zero; (void)pad_width; (void)rsa; (void)width_pad_modulo; // ELLPACK/RSA notions
  int concrete_height = rows;
  if (pad_height && height_pad_modulo > 0)
    concrete_height += height_pad_modulo - (concrete_height % height_pad_modulo);
  const unsigned long ixs_bytes = (unsigned long)col_idx_.size() * sizeof(int32_t);
  if (ixs_bytes > device_max_alloc_bytes)
    throw ixs_bytes;
  CL_matrix m;
  m.cl_height = concrete_height;
  m.cl_width = (int)max_width;
  m.indices.resize(ixs_bytes);
  if (ixs_bytes) std::memcpy(m.indices.data(), col_idx_.data(), ixs_bytes);
  m.values.resize(val_.size() * sizeof(T));
  if (!val_.empty()) std::memcpy(m.values.data(), val_.data(), m.values.size());
  std::vector<int32_t> rp(row_ptr_);
  rp.resize((std::size_t)concrete_height + 1, row_ptr_.empty() ? 0 : row_ptr_.back());
  m.row_ptr.resize(rp.size() * sizeof(int32_t));
  std::memcpy(m.row_ptr.data(), rp.data(), m.row_ptr.size());
  LOG_DEBUG("Done encoding");
  return m;
}

template <typename T> typename SparseMatrix<T>::template ellpack_matrix<T> &SparseMatrix<T>::ellpack_encode() {
  if (!ellpack_built_) {
    ellpack_cache_.assign((std::size_t)rows, ellpack_row<T>());
    for (int r = 0; r < rows; r++) {
      auto &row = ellpack_cache_[(std::size_t)r];
      row.reserve((std::size_t)(row_ptr_[r + 1] - row_ptr_[r]));
      for (int32_t j = row_ptr_[r]; j < row_ptr_[r + 1]; j++)
        row.emplace_back(col_idx_[(std::size_t)j], val_[(std::size_t)j]);
    }
    ellpack_built_ = true;
  }
  return ellpack_cache_;
}

template <typename T> void SparseMatrix<T>::pagerank_normalise(float dampingFactor, T zero) {
  start_timer(pagerank_normalise, sparse_matrix);
  if (raw_val_.size() != col_idx_.size() || file_order_.size() != col_idx_.size()) {
    LOG_ERROR("pagerank_normalise needs the file-order entries: call SparseMatrix::set_keep_entries(true) "
              "before loading the matrix");
    throw std::logic_error("pagerank_normalise without kept entries");
  }
  // column sums over the tuples in FILE order, in T arithmetic (:414-419); "column" is the
  // tuple's first field, i.e. this layout's column index
  std::vector<T> column_sums((std::size_t)width(), zero);
  for (std::size_t k = 0; k < file_order_.size(); k++) {
    const std::size_t pos = (std::size_t)file_order_[k];
    const std::size_t x = (std::size_t)col_idx_[pos];
    column_sums[x] = column_sums[x] + raw_val_[pos];
  }
  // (fabs(val) / column_sums[x]) * dampingFactor, evaluated as the reference's expression is:
  // ::fabs(double), so the quotient and product are double and rounded to T once (:428)
  const bool trunc = truncate();
  for (std::size_t pos = 0; pos < raw_val_.size(); pos++) {
    const T new_val =
        static_cast<T>((std::fabs(static_cast<double>(raw_val_[pos])) / column_sums[(std::size_t)col_idx_[pos]]) *
                       dampingFactor);
    raw_val_[pos] = new_val;
    val_[pos] = narrow(new_val, trunc);
  }
  ellpack_built_ = false;
  ellpack_cache_.clear();
}

template <typename T> void SparseMatrix<T>::scc_normalise() {
  start_timer(scc_normalise, sparse_matrix);
  const bool trunc = truncate();
  const bool keep = raw_val_.size() == col_idx_.size();
  for (int r = 0; r < rows; r++)
    for (int32_t k = row_ptr_[(std::size_t)r]; k < row_ptr_[(std::size_t)r + 1]; k++) {
      // tuple (x, y) = (column, row) of this layout (:445-453)
      const T nv = col_idx_[(std::size_t)k] == r ? std::numeric_limits<T>::min() : static_cast<T>(r);
      if (keep)
        raw_val_[(std::size_t)k] = nv;
      val_[(std::size_t)k] = narrow(nv, trunc);
    }
  ellpack_built_ = false;
  ellpack_cache_.clear();
}

template class SparseMatrix<float>;
template class SparseMatrix<int>;

// ---- C ABI for the Python side (sh_host.h) --------------------------------
extern "C" int sh_mm_load_ex(const char *path, int elem_is_int, int truncate_values, int normalise, double damping,
                             sh_host_csr *out) {
  if (!path || !out || normalise < 0 || normalise > 2)
    return -1;
  auto fill = [&](auto &m) {
    out->rows = m.height(); out->cols = m.width(); out->header_nnz = m.nonZeros();
    out->nnz = m.storedNonZeros();
    out->row_ptr = (int32_t *)std::malloc(sizeof(int32_t) * ((std::size_t)m.height() + 1));
    out->col_idx = (int32_t *)std::malloc(sizeof(int32_t) * std::max<std::size_t>(1, m.colIdx().size()));
    out->val = std::malloc(4 * std::max<std::size_t>(1, m.colIdx().size()));
    std::memcpy(out->row_ptr, m.rowPtr().data(), sizeof(int32_t) * m.rowPtr().size());
    std::memcpy(out->col_idx, m.colIdx().data(), sizeof(int32_t) * m.colIdx().size());
    std::memcpy(out->val, m.values().data(), 4 * m.values().size());
  };
  auto run = [&](auto tag) {
    using M = SparseMatrix<decltype(tag)>;
    M::set_truncate(truncate_values != 0);
    M::set_keep_entries(normalise == 1);
    M m{std::string(path)};
    M::set_keep_entries(false);
    if (normalise == 1)
      m.pagerank_normalise((float)damping, 0);
    else if (normalise == 2)
      m.scc_normalise();
    fill(m);
  };
  if (elem_is_int)
    run(int{});
  else
    run(float{});
  return 0;
}

extern "C" int sh_mm_load(const char *path, int elem_is_int, int truncate_values, sh_host_csr *out) {
  return sh_mm_load_ex(path, elem_is_int, truncate_values, 0, 0.0, out);
}

extern "C" void sh_host_csr_release(sh_host_csr *m) {
  if (!m) return;
  std::free(m->row_ptr); std::free(m->col_idx); std::free(m->val);
  m->row_ptr = m->col_idx = nullptr; m->val = nullptr;
}
