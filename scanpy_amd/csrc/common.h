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

// ---- small device -> host read-backs (counters, sizes) between the launches of a host-driven loop ------------------------
// hipMemcpyAsync into PAGEABLE host memory is a blocking round trip of its own (the runtime stages it): a kernel + two such
// copies + hipStreamSynchronize costs 39.6 us, one copy 25.9 us; into PINNED memory 17.8 / 15.3 us
// (tools/probes/host_roundtrip_probe.hip, MI355X).  `fetch` queues a copy into a per-thread pinned page and remembers where
// the caller wants the bytes; `sync` drains the stream once and hands them out.  The page is allocated on first use and kept
// for the life of the thread (8 KiB).
// Round 6: the copies themselves were blit kernels of the runtime (`__amd_rocclr_copyBuffer`, one per fetch: 182 per pass of the
// path, 3.5 ms of device time); now ONE small kernel of ours per synchronisation writes every queued item into the page
// (mapped pinned memory: the probe's 12.4 us variant).  Contract: the bytes handed out are the source's AT THE SYNCHRONISATION,
// so a source must not be overwritten between its `fetch` and the `sync` (no call site does).
struct HostReadbackItems {
  static constexpr int MAX_ITEMS = 8;
  const void* src[MAX_ITEMS];
  unsigned int off[MAX_ITEMS], bytes[MAX_ITEMS];
  int n;
};
static __global__ void host_readback_publish_kernel(HostReadbackItems it, char* page) {
  for (int i = 0; i < it.n; ++i) {
    const unsigned int nb = it.bytes[i];
    const char* s = static_cast<const char*>(it.src[i]);
    char* d = page + it.off[i];
    if ((reinterpret_cast<uintptr_t>(s) & 3u) == 0) {
      const unsigned int words = nb >> 2;
      for (unsigned int w = threadIdx.x; w < words; w += blockDim.x)
        reinterpret_cast<unsigned int*>(d)[w] = reinterpret_cast<const unsigned int*>(s)[w];
      for (unsigned int b = (words << 2) + threadIdx.x; b < nb; b += blockDim.x) d[b] = s[b];
    } else {
      for (unsigned int b = threadIdx.x; b < nb; b += blockDim.x) d[b] = s[b];
    }
  }
}
struct HostReadback {
  static constexpr size_t CAP = 8192;
  static constexpr int MAX_ITEMS = HostReadbackItems::MAX_ITEMS;
  struct Item {
    void* dst;
    size_t off, bytes;
  };
  char* pin = nullptr;      // host address of the page
  char* pin_dev = nullptr;  // the same page as the device sees it
  size_t used = 0;
  int n = 0;
  Item items[MAX_ITEMS];
  HostReadbackItems queued;
  void reset() { used = 0, n = 0; }  // (an error path may leave queued items behind: every C entry point starts with this)
  int fetch(void* dst, const void* src_device, size_t bytes, hipStream_t s) {
    (void)s;
    if (!pin) {
      SCAMD_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&pin), CAP, 0));
      SCAMD_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&pin_dev), pin, 0));
    }
    const size_t off = align_up(used, 8);
    SCAMD_REQUIRE(off + bytes <= CAP && n < MAX_ITEMS, SCAMD_EINTERNAL, "read-back staging page overflow (%zu + %zu bytes, %d items)",
                  off, bytes, n);
    queued.src[n] = src_device;
    queued.off[n] = (unsigned int)off;
    queued.bytes[n] = (unsigned int)bytes;
    items[n++] = Item{dst, off, bytes};
    used = off + bytes;
    return SCAMD_OK;
  }
  int sync(hipStream_t s) {
    if (n > 0) {
      queued.n = n;
      hipLaunchKernelGGL(host_readback_publish_kernel, dim3(1), dim3(256), 0, s, queued, pin_dev);
      const hipError_t le = hipGetLastError();
      if (le != hipSuccess) {
        reset();
        SCAMD_HIP_CHECK(le);
      }
    }
    const hipError_t e = hipStreamSynchronize(s);
    if (e == hipSuccess)
      for (int i = 0; i < n; ++i) memcpy(items[i].dst, pin + items[i].off, items[i].bytes);
    reset();
    SCAMD_HIP_CHECK(e);
    return SCAMD_OK;
  }
};
inline HostReadback& host_readback() {
  static thread_local HostReadback rb;
  return rb;
}
// declared by the host function (or its context) whose locals are the destinations of split fetches: nothing queued survives
// the function, whichever way it returns -- a later hand-out would write to its dead stack
struct HostReadbackScope {
  HostReadbackScope() { host_readback().reset(); }
  ~HostReadbackScope() { host_readback().reset(); }
};

// fetch + hand-out for code that returns SCAMD_* codes (every host function of the library).  SCAMD_READBACK_NOW = one value
// read at once (it also drops whatever an earlier call's error path may have left queued -- the hand-out would write to
// that call's dead stack); the split form is for values fetched at different points and handed out by one synchronisation.
#define SCAMD_READBACK(dst, src, bytes, stream)                                               \
  do {                                                                                        \
    const int rc_rb_ = ::scamd::host_readback().fetch((dst), (src), (bytes), (stream));       \
    if (rc_rb_ != SCAMD_OK) return rc_rb_;                                                    \
  } while (0)
#define SCAMD_READBACK_NOW(dst, src, bytes, stream)                                            \
  do {                                                                                        \
    ::scamd::host_readback().reset();                                                         \
    int rc_rb_ = ::scamd::host_readback().fetch((dst), (src), (bytes), (stream));             \
    if (rc_rb_ == SCAMD_OK) rc_rb_ = ::scamd::host_readback().sync(stream);                   \
    if (rc_rb_ != SCAMD_OK) return rc_rb_;                                                    \
  } while (0)
#define SCAMD_READBACK_SYNC(stream)                                  \
  do {                                                               \
    const int rc_rb_ = ::scamd::host_readback().sync(stream);        \
    if (rc_rb_ != SCAMD_OK) return rc_rb_;                           \
  } while (0)

}  // namespace scamd
