// Shared host-side helpers of the slamhip C ABI: thread-local error text and
// argument / HIP status checks that turn failures into return codes.
#pragma once
#include <cstdarg>
#include <cstdio>

#include "../../include/slamhip.h"

namespace slamhip {

char* error_buffer();          // thread-local, 512 bytes
int fail(int code, const char* fmt, ...);

}  // namespace slamhip

#define SLAMHIP_REQUIRE(cond, code, ...)            \
  do {                                              \
    if (!(cond)) return slamhip::fail(code, __VA_ARGS__); \
  } while (0)

#define SLAMHIP_HIP(call)                                                              \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess)                                                              \
      return slamhip::fail(SLAMHIP_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
  } while (0)
