// oracle/ref/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Golden-vector generator that runs the *real* reference code in the
// authoring container.  It is compiled by oracle/ref/Makefile against the
// reference sources where they lie under /root/reference (never copied into
// this repo) and writes raw arrays that tests/golden/make_golden.py packs into
// the committed .npz fixtures.
//
// What is executed from the reference:
//   * SparseMatrix<T>::load_from_file / calculate_ellpack / cl_encode
//       (/root/reference/src/sparse_matrix.cpp:11-119,122-399)
//   * Gold<T>::spmv (/root/reference/inc/spmv_gold.h:9-28)
//   * SparseMatrix<T>::pagerank_normalise / scc_normalise (src/sparse_matrix.cpp:409-456)
//   * the Lift `glb-sdp` OpenCL kernels of example/{spmv,sssp,bfs,pr,scc}/kernel5.json,
//     compiled as C99 (oracle/ref/extract_kernels.py), fed with cl_encode's
//     own ELLPACK buffers.
// The iterative drivers below mirror the do/while loops of
// /root/reference/app/sssp.cpp:97-176 and app/bfs.cpp:94-174 (those files
// cannot be built here: they need Boost and an OpenCL device).
#include <cfloat>
#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <unistd.h>
#include <vector>

#include "sparse_matrix.h"
#include "spmv_gold.h"
#include "vector_generator.h"

extern "C" {
void KERNEL_spmv(const int *idx, const float *val, const float *x,
                 const float *y, float alpha, float beta, float *out,
                 float *tmp, int MHeight, int MWidthC, int VLength);
void KERNEL_sssp(const int *idx, const float *val, const float *x,
                 const float *y, float alpha, float beta, float *out,
                 float *tmp, int MHeight, int MWidthC, int VLength);
void KERNEL_bfs(const int *idx, const int *val, const int *x, const int *y,
                int alpha, int beta, int *out, int *tmp, int MHeight,
                int MWidthC, int VLength);
void KERNEL_pr(const int *idx, const float *val, const float *x,
               const float *y, float alpha, float beta, float *out,
               float *tmp, int MHeight, int MWidthC, int VLength);
void KERNEL_scc(const int *idx, const int *val, const int *x, const int *y,
                int alpha, int beta, int *out, int *tmp, int MHeight,
                int MWidthC, int VLength);
}

static std::string g_out;

template <typename T>
static void dump(const std::string &key, const char *dtype, const T *p,
                 size_t n) {
  std::string path = g_out + "." + key + "." + dtype;
  FILE *f = fopen(path.c_str(), "wb");
  if (!f) {
    perror(path.c_str());
    exit(1);
  }
  if (n)
    fwrite(p, sizeof(T), n, f);
  fclose(f);
}

// x[i] = 1 + (i mod 7): a non-constant x so that a wrong gather cannot pass.
template <typename T> class ModXGen : public XVectorGenerator<T> {
public:
  virtual T get(int ix) { return static_cast<T>(1 + (ix % 7)); }
};
// y[i] = i mod 5
template <typename T> class ModYGen : public YVectorGenerator<T> {
public:
  virtual T get(int ix) { return static_cast<T>(ix % 5); }
};
// initial vectors of app/sssp.cpp:179-209 / app/bfs.cpp:177-207
template <typename T> class SourceGen : public XVectorGenerator<T> {
  T at0, other;

public:
  SourceGen(T a, T o) : at0(a), other(o) {}
  virtual T get(int ix) { return ix == 0 ? at0 : other; }
};

// The reference prints whole matrices on stdout inside cl_encode
// (src/sparse_matrix.cpp:387-395); silence fd 1 while it runs.
struct StdoutMute {
  int saved;
  StdoutMute() {
    fflush(stdout);
    std::cout.flush();
    saved = dup(1);
    FILE *n = fopen("/dev/null", "w");
    dup2(fileno(n), 1);
    fclose(n);
  }
  ~StdoutMute() {
    fflush(stdout);
    std::cout.flush();
    dup2(saved, 1);
    close(saved);
  }
};

template <typename T>
static void dump_csr(SparseMatrix<T> &m, const char *tag, const char *vdtype) {
  auto &rows = m.ellpack_encode();
  std::vector<int> row_ptr(rows.size() + 1, 0), col;
  std::vector<T> val;
  for (size_t r = 0; r < rows.size(); r++) {
    for (auto &e : rows[r]) {
      col.push_back(e.first);
      val.push_back(e.second);
    }
    row_ptr[r + 1] = (int)col.size();
  }
  dump(std::string(tag) + "_row_ptr", "i32", row_ptr.data(), row_ptr.size());
  dump(std::string(tag) + "_col_idx", "i32", col.data(), col.size());
  dump(std::string(tag) + "_val", vdtype, val.data(), val.size());
}

int main(int argc, char **argv) {
  if (argc != 3) {
    fprintf(stderr, "usage: ref_driver <matrix.mtx> <out_prefix>\n");
    return 2;
  }
  std::string file = argv[1];
  g_out = argv[2];
  const int ITER_CAP = 2000;

  // ---------------- float matrix: CSR view, gold, spmv + sssp kernels ------
  {
    SparseMatrix<float> m(file);
    int dims[3] = {m.height(), m.width(), m.nonZeros()};
    dump("dims", "i32", dims, 3);
    dump_csr<float>(m, "f32", "f32");

    ConstXVectorGenerator<float> x1(1.0f);
    ConstYVectorGenerator<float> y0(0.0f);
    auto g1 = Gold<float>::spmv(m, x1, y0, 1.0f, 0.0f, 0.0f);
    dump("gold_x1", "f32", g1.data(), g1.size());

    ModXGen<float> xm;
    auto g2 = Gold<float>::spmv(m, xm, y0, 1.0f, 0.0f, 0.0f);
    dump("gold_xmod", "f32", g2.data(), g2.size());

    // quirk A-4: beta*y.get(value) added per non-zero
    ConstYVectorGenerator<float> y3(3.0f);
    auto g3 = Gold<float>::spmv(m, xm, y3, 2.0f, 0.5f, 0.0f);
    dump("gold_ab", "f32", g3.data(), g3.size());

    // ---- Lift spmv kernel on the reference's own ELLPACK encoding
    CL_matrix *enc;
    {
      StdoutMute mute;
      enc = new CL_matrix(
          m.cl_encode(0xFFFFFFFFu, 0.0f, false, false, false, -1, -1));
    }
    int H = enc->cl_height, W = enc->cl_width;
    int ew[2] = {H, W};
    dump("ell_hw", "i32", ew, 2);
    const int *idx = reinterpret_cast<const int *>(enc->indices.data());
    const float *val = reinterpret_cast<const float *>(enc->values.data());
    std::vector<float> tmp((size_t)H * W), out(H);
    {
      auto xv = x1.generate(H);
      auto yv = y0.generate(H);
      KERNEL_spmv(idx, val, xv.data(), yv.data(), 1.0f, 0.0f, out.data(),
                  tmp.data(), H, W, H);
      dump("kern_spmv_x1", "f32", out.data(), out.size());
    }
    {
      ModYGen<float> ym;
      auto xv = xm.generate(H);
      auto yv = ym.generate(H);
      KERNEL_spmv(idx, val, xv.data(), yv.data(), 2.0f, 0.5f, out.data(),
                  tmp.data(), H, W, H);
      dump("kern_spmv_ab", "f32", out.data(), out.size());
    }
    delete enc;

    // ---- SSSP: pad value FLT_MAX (app/sssp.cpp:231), alpha=beta=0 (:219-220)
    {
      StdoutMute mute;
      enc = new CL_matrix(
          m.cl_encode(0xFFFFFFFFu, FLT_MAX, false, false, false, -1, -1));
    }
    idx = reinterpret_cast<const int *>(enc->indices.data());
    val = reinterpret_cast<const float *>(enc->values.data());
    SourceGen<float> d0(0.0f, FLT_MAX);
    std::vector<float> in = d0.generate(H), yv = d0.generate(H), o(H, 0.0f);
    // host mirrors (inc/cl_memory_manager.h:10-12): input = x, output = zeros
    std::vector<float> *pin = &in, *pout = &o;
    const float *ydev = yv.data();
    const double delta = 0.0001; // inc/common.h:28-29
    int iters = 0;
    bool term = false;
    std::vector<float> first;
    do {
      KERNEL_sssp(idx, val, pin->data(), ydev, 0.0f, 0.0f, pout->data(),
                  tmp.data(), H, W, H);
      if (iters == 0)
        first = *pout;
      bool equal = true; // app/sssp.cpp:157-176
      for (int i = 0; equal && i < H; i++)
        equal = fabs((*pin)[i] - (*pout)[i]) < delta;
      term = equal;
      std::swap(pin, pout);
      ydev = pin->data(); // setGlobalArg(3, input_mem_ptr), app/sssp.cpp:150
      iters++;
    } while (!term && iters < ITER_CAP);
    int meta[2] = {iters, term ? 1 : 0};
    dump("sssp_meta", "i32", meta, 2);
    dump("sssp_first", "f32", first.data(), first.size());
    dump("sssp_final", "f32", pin->data(), pin->size());
    delete enc;
  }

  // ---------------- int matrix: bfs kernel --------------------------------
  {
    SparseMatrix<int> m(file);
    dump_csr<int>(m, "i32", "i32");
    CL_matrix *enc;
    {
      StdoutMute mute;
      enc = new CL_matrix(m.cl_encode(0xFFFFFFFFu, 0, false, false, false, -1, -1));
    }
    int H = enc->cl_height, W = enc->cl_width;
    const int *idx = reinterpret_cast<const int *>(enc->indices.data());
    const int *val = reinterpret_cast<const int *>(enc->values.data());
    std::vector<int> tmp((size_t)H * W);
    SourceGen<int> f0(1, 0);
    std::vector<int> in = f0.generate(H), yv = f0.generate(H), o(H, 0);
    std::vector<int> *pin = &in, *pout = &o;
    const int *ydev = yv.data();
    int iters = 0;
    bool term = false;
    std::vector<int> first;
    do {
      KERNEL_bfs(idx, val, pin->data(), ydev, 1, 0, pout->data(), tmp.data(),
                 H, W, H);
      if (iters == 0)
        first = *pout;
      bool equal = true; // app/bfs.cpp:154-174
      for (int i = 0; equal && i < H; i++)
        equal = (*pin)[i] == (*pout)[i];
      term = equal;
      std::swap(pin, pout);
      ydev = pin->data();
      iters++;
    } while (!term && iters < ITER_CAP);
    int meta[2] = {iters, term ? 1 : 0};
    dump("bfs_meta", "i32", meta, 2);
    dump("bfs_first", "i32", first.data(), first.size());
    dump("bfs_final", "i32", pin->data(), pin->size());
    delete enc;
  }
  // ---------------- PageRank: app/pr.cpp:179-215 (x = 1/N, y = 1, alpha = 1,
  // beta = (1-d)/N, pagerank_normalise(0.85, 0) before encoding) -------------
  {
    SparseMatrix<float> m(file);
    const float damping = 0.85f;
    m.pagerank_normalise(damping, 0.0f);
    dump_csr<float>(m, "pr", "f32");   // rows AFTER normalise + int narrowing (quirk A-3)
    CL_matrix *enc;
    {
      StdoutMute mute;
      enc = new CL_matrix(m.cl_encode(0xFFFFFFFFu, 0.0f, false, false, false, -1, -1));
    }
    int H = enc->cl_height, W = enc->cl_width;
    const int *idx = reinterpret_cast<const int *>(enc->indices.data());
    const float *val = reinterpret_cast<const float *>(enc->values.data());
    std::vector<float> tmp((size_t)H * W);
    ConstXVectorGenerator<float> x0(1.0f / (float)m.height());
    ConstYVectorGenerator<float> y0(1.0f);
    const float alpha = 1.0f, beta = (1.0f - damping) / (float)m.height();
    std::vector<float> in = x0.generate(H), yv = y0.generate(H), o(H, 0.0f);
    std::vector<float> *pin = &in, *pout = &o;
    const float *ydev = yv.data();
    const double delta = 0.0001;
    int iters = 0;
    bool term = false;
    std::vector<float> first;
    do {
      KERNEL_pr(idx, val, pin->data(), ydev, alpha, beta, pout->data(), tmp.data(), H, W, H);
      if (iters == 0) first = *pout;
      bool equal = true; // app/pr.cpp:157-176
      for (int i = 0; equal && i < H; i++) equal = fabs((*pin)[i] - (*pout)[i]) < delta;
      term = equal;
      std::swap(pin, pout);
      ydev = pin->data();
      iters++;
    } while (!term && iters < ITER_CAP);
    int meta[2] = {iters, term ? 1 : 0};
    dump("pr_meta", "i32", meta, 2);
    dump("pr_first", "f32", first.data(), first.size());
    dump("pr_final", "f32", pin->data(), pin->size());
    delete enc;
  }
  // ---------------- SCC: app/scc.cpp:179-251 (x[i] = i, y = INT_MIN, alpha = INT_MAX,
  // beta = INT_MIN, zero = INT_MIN, scc_normalise() before encoding) --------------
  {
    SparseMatrix<int> m(file);
    m.scc_normalise();
    dump_csr<int>(m, "scc", "i32");
    CL_matrix *enc;
    {
      StdoutMute mute;
      enc = new CL_matrix(m.cl_encode(0xFFFFFFFFu, INT_MIN, false, false, false, -1, -1));
    }
    int H = enc->cl_height, W = enc->cl_width;
    const int *idx = reinterpret_cast<const int *>(enc->indices.data());
    const int *val = reinterpret_cast<const int *>(enc->values.data());
    std::vector<int> tmp((size_t)H * W);
    std::vector<int> in(H), yv(H, INT_MIN), o(H, 0);
    for (int i = 0; i < H; i++) in[i] = i;   // InitialComponentsGeneratorX, app/scc.cpp:176-185
    std::vector<int> *pin = &in, *pout = &o;
    const int *ydev = yv.data();
    int iters = 0;
    bool term = false;
    std::vector<int> first;
    do {
      KERNEL_scc(idx, val, pin->data(), ydev, INT_MAX, INT_MIN, pout->data(), tmp.data(), H, W, H);
      if (iters == 0) first = *pout;
      bool equal = true; // app/scc.cpp:154-172
      for (int i = 0; equal && i < H; i++) equal = (*pin)[i] == (*pout)[i];
      term = equal;
      std::swap(pin, pout);
      ydev = pin->data();
      iters++;
    } while (!term && iters < ITER_CAP);
    int meta[2] = {iters, term ? 1 : 0};
    dump("scc_meta", "i32", meta, 2);
    dump("scc_first", "i32", first.data(), first.size());
    dump("scc_final", "i32", pin->data(), pin->size());
    delete enc;
  }
  return 0;
}
