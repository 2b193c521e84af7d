// hip_common.h -- shared helpers for the HIP translation units of libqemb_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include "dev_ops.h"

namespace qemb {
hipStream_t hip_stream();  // the library stream (created by dev_init)

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      ::qemb::set_error(std::string(#expr) + " failed: " + hipGetErrorString(_e) + " at " +   \
                        __FILE__ + ":" + std::to_string(__LINE__));                           \
      return QEMB_ERR_DEVICE;                                                              \
    }                                                                                         \
  } while (0)
}  // namespace qemb
