// hip_utils.h -- error convention + small device helpers of the host mirror.
// Replaces inc/opencl_utils.h of the reference: checkCLError -> LOG_ERROR +
// exit(1) (:15-23) becomes checkSHError over the C ABI's return codes;
// deviceGetMaxAllocSize (:216-226) asks the HIP engine; assertBuffersNotEqual
// (:247-258) is kept verbatim in behaviour (it only logs).
#pragma once
#include <cstdlib>
#include <cstring>
#include <vector>

#include "logger.h"
#include "sparseharness_hip.h"

#define checkSHError(engine, call)                                                   \
  do {                                                                               \
    int _sh_rc = (call);                                                             \
    if (_sh_rc != SH_OK) {                                                           \
      LOG_ERROR("HIP engine error ", _sh_rc, " in ", #call, ": ", sh_last_error(engine)); \
      std::exit(1);                                                                  \
    }                                                                                \
  } while (0)

inline unsigned long deviceGetMaxAllocSize(unsigned int /*platform*/, unsigned int device) {
  sh_engine *e = nullptr;
  if (sh_engine_create((int)device, &e) != SH_OK) {
    LOG_ERROR("Cannot open HIP device ", device, ": ", sh_last_error(nullptr));
    std::exit(1);
  }
  uint64_t bytes = 0;
  checkSHError(e, sh_engine_max_alloc(e, &bytes));
  sh_engine_destroy(e);
  return (unsigned long)bytes;
}

inline void assertBuffersNotEqual(std::vector<char> &v1, std::vector<char> &v2) {
  if (v1.size() != v2.size()) {
    LOG_DEBUG_INFO("Buffers have different sizes: ", v1.size(), " vs ", v2.size());
    return;
  }
  if (v1.empty() || std::memcmp(v1.data(), v2.data(), v1.size()) == 0)
    LOG_WARNING("Buffers are equal: the kernel output did not change");
}
