// Shared host-side helpers for libstargcn_hip.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/stargcn.h"

#define SG_API extern "C" __attribute__((visibility("default")))

namespace sg {

// thread-local message returned by sg_last_error()
void set_error(const char* fmt, ...);

inline int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
inline int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  set_error("%s", buf);
  return code;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(SG_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
  return SG_OK;
}

inline bool valid_req(int req) { return req == SG_REQ_NULL || req == SG_REQ_WRITE || req == SG_REQ_ADD; }

inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

constexpr int kWave = 64;  // CDNA4 wavefront

}  // namespace sg
