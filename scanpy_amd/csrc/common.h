// Shared host-side helpers for libscanpy_amd.so (error string, workspace carving, launch checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/scanpy_amd.h"

namespace scamd {

void set_error(const char* fmt, ...);

#define SCAMD_HIP_CHECK(expr)                                                              \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      ::scamd::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return SCAMD_EHIP;                                                                   \
    }                                                                                      \
  } while (0)

#define SCAMD_LAUNCH_CHECK() SCAMD_HIP_CHECK(hipGetLastError())

#define SCAMD_REQUIRE(cond, code, ...)   \
  do {                                   \
    if (!(cond)) {                       \
      ::scamd::set_error(__VA_ARGS__);   \
      return (code);                     \
    }                                    \
  } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Bump allocator over the caller's workspace.  In "measure" mode (base == nullptr) it only adds up
// sizes, so the same carving code yields scamd_*_workspace_bytes().
struct Workspace {
  char* base;
  size_t cap;
  size_t off = 0;
  bool ok = true;
  Workspace(void* b, size_t c) : base(static_cast<char*>(b)), cap(c) {}
  template <typename T>
  T* take(size_t count) {
    off = align_up(off, 256);
    size_t bytes = count * sizeof(T);
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += bytes;
    if (base && off > cap) ok = false;
    return p;
  }
  size_t used() const { return align_up(off, 256); }
};

inline int ceil_div(int64_t a, int64_t b) { return static_cast<int>((a + b - 1) / b); }

}  // namespace scamd
