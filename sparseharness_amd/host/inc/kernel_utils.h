// kernel_utils.h -- argument marshalling: SparseMatrix x KernelConfig x vector
// generators -> ArgContainer<T> (reference: inc/kernel_utils.h:18-31,35-168).
// Same container members and the same size rules (MHeight = VLength =
// encoded height incl. chunk padding, MWidthC = width / |splitSize| or the
// matrix width for "ragged" kernels, x and y generated at VLength, output
// and temp sizes evaluated from the JSON strings, size args ordered by
// paramVars).  Native differences: the matrix travels as CSR (m_row_ptr is
// new; m_idxs / m_vals hold col_idx / val), max_alloc is a full
// `unsigned long` (fixes quirk A-7) and the matrix is taken by reference.
#pragma once
#include <cstdlib>
#include <iostream>
#include <map>
#include <vector>

#include "arithexpr_evaluator.h"
#include "buffer_utils.h"
#include "csds_timer.h"
#include "kernel_config.h"
#include "logger.h"
#include "sparse_matrix.h"
#include "vector_generator.h"

typedef std::vector<char> raw_arg;

template <typename T> class ArgContainer {
public:
  raw_arg m_idxs;    // int32 col_idx[nnz]
  raw_arg m_vals;    // T val[nnz]
  raw_arg m_row_ptr; // int32 row_ptr[MHeight + 1]   (CSR addition)
  raw_arg x_vect;
  raw_arg y_vect;
  T alpha = T(1);
  T beta = T(1);
  // sizes ready for allocation (bytes), as in the reference
  std::vector<unsigned int> temp_globals;
  unsigned int output = 0;
  std::vector<unsigned int> temp_locals;
  std::vector<unsigned int> size_args;
  // CSR dimensions of the encoded matrix
  int rows = 0;
  int cols = 0;
};

template <typename T>
ArgContainer<T> executorEncodeMatrix(unsigned long device_max_alloc_bytes, KernelConfig<T> &kernel,
                                     SparseMatrix<T> &matrix, T zero, XVectorGenerator<T> &xgen,
                                     YVectorGenerator<T> &ygen, T alpha = static_cast<T>(1),
                                     T beta = static_cast<T>(1)) {
  start_timer(executorEncodeMatrix, kernel_utils);
  auto kprops = kernel.getProperties();
  const bool ragged = kprops.arrayType == "ragged";
  auto cl_matrix = matrix.cl_encode(device_max_alloc_bytes, zero, kprops.chunkSize != -1, kprops.splitSize != -1,
                                    ragged, kprops.chunkSize, kprops.splitSize);

  // Width the reference's ELLPACK encoding would have had (src/sparse_matrix.cpp:166-177)
  int regular_width = cl_matrix.cl_width;
  if (!ragged && kprops.splitSize != -1)
    regular_width += kprops.splitSize - (regular_width % kprops.splitSize);
  const int v_MWidth_1 = ragged ? matrix.width() : regular_width / std::abs(kprops.splitSize);
  const int v_MHeight_2 = cl_matrix.cl_height;
  const int v_VLength_3 = cl_matrix.cl_height;
  std::cerr << "Encoding matrix with sizes:"
            << "\n\tv_MWidth_1 = " << v_MWidth_1 << "\n\tv_MHeight_2 = " << v_MHeight_2
            << "\n\tv_VLength_3 = " << v_VLength_3 << "\n";

  ArgContainer<T> arg_cnt;
  arg_cnt.rows = cl_matrix.cl_height;
  arg_cnt.cols = v_VLength_3;
  arg_cnt.m_idxs = std::move(cl_matrix.indices);
  arg_cnt.m_vals = std::move(cl_matrix.values);
  arg_cnt.m_row_ptr = std::move(cl_matrix.row_ptr);
  arg_cnt.x_vect = enchar<T>(xgen.generate(v_VLength_3));
  arg_cnt.y_vect = enchar<T>(ygen.generate(v_VLength_3));
  arg_cnt.alpha = alpha;
  arg_cnt.beta = beta;

  arg_cnt.output = (unsigned int)Evaluator::evaluate(kernel.getOutputArg()->size, v_MWidth_1, v_MHeight_2, v_VLength_3);
  if (arg_cnt.output < (unsigned int)v_MHeight_2 * sizeof(T))
    arg_cnt.output = (unsigned int)v_MHeight_2 * sizeof(T); // "?"-sized or missing: one element per row
  for (auto &arg : kernel.getTempGlobals())
    arg_cnt.temp_globals.push_back((unsigned int)Evaluator::evaluate(arg.size, v_MWidth_1, v_MHeight_2, v_VLength_3));
  for (auto &arg : kernel.getTempLocals())
    arg_cnt.temp_locals.push_back((unsigned int)Evaluator::evaluate(arg.size, v_MWidth_1, v_MHeight_2, v_VLength_3));

  std::map<std::string, int> sizeMap{{"MWidthC", v_MWidth_1}, {"MHeight", v_MHeight_2}, {"VLength", v_VLength_3}};
  for (auto &sizeArg : kernel.getParamVars()) {
    LOG_DEBUG("Size argument - name: ", sizeArg, " value: ", sizeMap[sizeArg]);
    arg_cnt.size_args.push_back((unsigned int)sizeMap[sizeArg]);
  }
  return arg_cnt;
}
