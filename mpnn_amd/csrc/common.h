// Shared host-side helpers for the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/mpnn_amd.h"

namespace mpnn {

// Thread-local text of the last failure, returned by mpnn_last_error().
char* error_buffer();

inline int32_t fail(int32_t code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
inline int32_t fail(int32_t code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}

#define MPNN_HIP_CHECK(expr)                                                              \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return ::mpnn::fail(MPNN_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                          __FILE__, __LINE__);                                            \
  } while (0)

#define MPNN_LAUNCH_CHECK(name)                                                           \
  do {                                                                                    \
    hipError_t e_ = hipGetLastError();                                                    \
    if (e_ != hipSuccess)                                                                 \
      return ::mpnn::fail(MPNN_E_HIP, "launch of %s failed: %s", name, hipGetErrorString(e_)); \
  } while (0)

inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

// Bump allocator over the caller's workspace.
struct Arena {
  char* base;
  int64_t size;
  int64_t used = 0;
  Arena(void* p, int64_t n) : base(static_cast<char*>(p)), size(n) {}
  template <typename T>
  T* take(int64_t count) {
    int64_t bytes = align_up(count * (int64_t)sizeof(T), 256);
    if (used + bytes > size) return nullptr;
    T* r = reinterpret_cast<T*>(base + used);
    used += bytes;
    return r;
  }
};

// Launch geometry shared by the row kernels: 256 threads (4 waves) per block and at
// most 8 blocks per CU worth of blocks on a 256-CU part; the rest is grid-strided.
constexpr int kBlock = 256;
constexpr int kMaxBlocks = 2048;

}  // namespace mpnn
