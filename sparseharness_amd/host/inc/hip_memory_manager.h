// hip_memory_manager.h -- the bag of device + host buffers a harness owns.
// Replaces CLMemoryManager (inc/cl_memory_manager.h:6-29) member for member:
// cl_mem becomes device_mem (an engine vector handle); _matrix_idxs and
// _matrix_vals both designate the one device CSR object, because the native
// layout uploads indices, values and row_ptr together.
#pragma once
#include <vector>

#include "kernel_utils.h"
#include "sparseharness_hip.h"

typedef sh_vec *device_mem;

template <typename SemiringType> class HipMemoryManager {
public:
  explicit HipMemoryManager(ArgContainer<SemiringType> &args)
      : _args(args), _temp_global(args.temp_globals.size(), nullptr),
        _input_host_buffer(args.x_vect.begin(), args.x_vect.end()), _output_host_buffer(args.output, 0),
        _temp_out_buffer(args.output, 0) {}

  ArgContainer<SemiringType> &_args;
  sh_csr *_matrix = nullptr; // device CSR (+ launch schedule)
  sh_csr *&_matrix_idxs = _matrix;
  sh_csr *&_matrix_vals = _matrix;
  device_mem _x_vect = nullptr;
  device_mem _y_vect = nullptr;
  device_mem _output = nullptr;
  std::vector<device_mem> _temp_global; // never allocated: native kernels use no global temporaries

  unsigned int _arg_index = 0;
  unsigned int _input_idx = 2;
  unsigned int _output_idx = 0;

  std::vector<char> _input_host_buffer;
  std::vector<char> _output_host_buffer;
  std::vector<char> _temp_out_buffer;
};
