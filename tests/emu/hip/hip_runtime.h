// TEST INFRASTRUCTURE -- not part of the product, never loaded by `scanpy_amd`.
//
// A stand-in for <hip/hip_runtime.h> that lets the UNMODIFIED kernel sources of scanpy_amd/csrc be compiled for the
// host and executed lane by lane (tests/emu/README.md).  Every GPU thread is a fibre; `__syncthreads`, the wave
// shuffles / ballots / readlane / DPP / swizzle and the MFMA builtins are rendezvous points of the fibres of a
// workgroup or of a wave.  What this buys: the kernels' INDEXING and control flow run under AddressSanitizer and
// under a checker for cross-lane operations executed by a partial wave (the bug class behind round 2's irreproducible
// Leiden), with no GPU.  What it does not: timing, the memory model, occupancy, anything about performance.
#pragma once

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>
#include <ctime>
#include <vector>
#include <cstdarg>

// ---------------------------------------------------------------------------------------------- qualifiers
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define EMU_INL static inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define __constant__ static
#define HIP_KERNEL_NAME(...) __VA_ARGS__

// ---------------------------------------------------------------------------------------------- vector types
#define EMU_VEC2(T, N) \
  struct N { T x, y; }; \
  static inline N make_##N(T x, T y) { return N{x, y}; }
#define EMU_VEC3(T, N) \
  struct N { T x, y, z; }; \
  static inline N make_##N(T x, T y, T z) { return N{x, y, z}; }
#define EMU_VEC4(T, N) \
  struct alignas(sizeof(T) * 4 > 16 ? 16 : sizeof(T) * 4) N { T x, y, z, w; }; \
  static inline N make_##N(T x, T y, T z, T w) { return N{x, y, z, w}; }
EMU_VEC2(int, int2) EMU_VEC2(unsigned, uint2) EMU_VEC2(float, float2) EMU_VEC2(double, double2)
EMU_VEC2(long long, longlong2) EMU_VEC2(unsigned long long, ulonglong2) EMU_VEC2(short, short2) EMU_VEC2(unsigned short, ushort2)
EMU_VEC3(int, int3) EMU_VEC3(unsigned, uint3) EMU_VEC3(float, float3)
EMU_VEC4(int, int4) EMU_VEC4(unsigned, uint4) EMU_VEC4(float, float4) EMU_VEC4(double, double4)
EMU_VEC4(short, short4) EMU_VEC4(unsigned short, ushort4)

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

// ---------------------------------------------------------------------------------------------- runtime API
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
typedef struct ihipStream_t* hipStream_t;  // (the name include/scanpy_amd.h spells out for its scamd_stream_t)
typedef struct emuEvent { double t; } * hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };

static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "emulated HIP error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <typename T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc(reinterpret_cast<void**>(p), n); }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned = 0) { *d = h; return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height,
                                          hipMemcpyKind, hipStream_t = nullptr) {
  for (size_t r = 0; r < height; ++r) memmove(static_cast<char*>(d) + r * dpitch, static_cast<const char*>(s) + r * spitch, width);
  return hipSuccess;
}
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new emuEvent{0.0}; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
template <typename F> static inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }

// ---------------------------------------------------------------------------------------------- the emulator
namespace emu {
struct Idx3 { unsigned x, y, z; };
struct Lane;
// every lane of the collective that arrived: `opnd` / `res` point into the (suspended) fibres' stacks
struct Arrived {
  const void* opnd;
  void* res;
};
typedef void (*ComputeAll)(unsigned long long active, const Arrived* lanes /*[64]*/, const void* uniform);
struct Lane {
  Idx3 tidx, bidx, bdim, gdim;
  int lane, wave, flat;
  // scheduler state (tests/emu/emu_runtime.cpp)
  void* sp;
  int state;
  long long seq;  // when the lane last blocked
  int kind;
  const void* site;
  ComputeAll fn;
  const void* uniform;
  Arrived arr;
  // LDS-DMA requests of this lane that have not landed yet (EMU_DMA=late, see dma_issue)
  struct Dma { void* dst; const void* src; int size; } dma[96];
  int n_dma;
};
extern Lane* g_cur;
// LDS-DMA (`__builtin_amdgcn_global_load_lds`): on the hardware the bytes land some time between the request and the
// `s_waitcnt vmcnt(k)` that covers it.  Two legal extremes are emulated: EMU_DMA=early (default) copies at the request --
// a request into a buffer somebody still reads corrupts that read (write-after-read hazards show); EMU_DMA=late
// (emu_set_dma_late(1)) copies at the wait that retires the request -- SCAMD_BARRIER_VM(k) lands all but the lane's k
// youngest, `__syncthreads()` (whose fence waits for vmcnt(0)) and the end of the kernel land everything -- so that a read
// placed before its wait sees the OLD bytes (read-after-write hazards show).  The tests run the kNN both ways.
extern int g_dma_late;
static inline void dma_land(int keep) {
  Lane* me = g_cur;
  const int n = me->n_dma - keep;
  for (int i = 0; i < n; ++i) memcpy(me->dma[i].dst, me->dma[i].src, (size_t)me->dma[i].size);
  if (n > 0) {
    for (int i = n; i < me->n_dma; ++i) me->dma[i - n] = me->dma[i];
    me->n_dma -= n;
  }
}
static inline void dma_issue(void* dst, const void* src, int size) {
  Lane* me = g_cur;
  if (!g_dma_late) {
    memcpy(dst, src, (size_t)size);
    return;
  }
  if (me->n_dma == 96) dma_land(95);  // (more outstanding requests than the counter could hold: the oldest has landed)
  me->dma[me->n_dma++] = Lane::Dma{dst, src, size};
}
// (convergent + noduplicate: the host compiler must treat a rendezvous as the device compiler treats a cross-lane
// instruction -- never clone it into the two arms of a branch: each clone would be a call site of its own and the lanes of
// the two arms would stop meeting)
__attribute__((convergent, noduplicate)) void wave_collective(int kind, const void* opnd, void* res, ComputeAll fn,
                                                              const void* uniform = nullptr);
__attribute__((convergent, noduplicate)) void block_barrier();
void* dyn_lds();
void launch_body(dim3 grid, dim3 block, size_t shmem, void (*tramp)(void*), void* closure);

template <typename... KArgs, typename... Args>
static inline void launch_ggl(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t shmem, hipStream_t, Args&&... args) {
  auto body = [&]() { kernel(static_cast<KArgs>(args)...); };
  using B = decltype(body);
  launch_body(grid, block, shmem, [](void* c) { (*static_cast<B*>(c))(); }, &body);
}

enum Kind { K_SHFL = 1, K_BALLOT, K_READLANE, K_READFIRST, K_DPP, K_SWIZZLE, K_MFMA_F32_32X32X2, K_MFMA_BF16_32X32X16,
            K_MFMA_F64_16X16X4, K_WAVE_BARRIER, K_ANY, K_SOFT_RMW, K_SOFT_LOAD };

// ---- generic helpers: a 64-bit payload per lane covers every shuffled type used
template <typename T> static inline unsigned long long to_bits(T v) {
  static_assert(sizeof(T) <= 8, "shuffle payload");
  unsigned long long b = 0;
  memcpy(&b, &v, sizeof(T));
  return b;
}
template <typename T> static inline T from_bits(unsigned long long b) {
  T v;
  memcpy(&v, &b, sizeof(T));
  return v;
}
struct ShflOp { unsigned long long val; int src; };
void shfl_all(unsigned long long active, const Arrived* l, const void*);
void ballot_all(unsigned long long active, const Arrived* l, const void*);
void readlane_all(unsigned long long active, const Arrived* l, const void*);
void readfirst_all(unsigned long long active, const Arrived* l, const void*);
void barrier_all(unsigned long long active, const Arrived* l, const void*);
void mfma_f32_32x32x2_all(unsigned long long active, const Arrived* l, const void*);
void mfma_bf16_32x32x16_all(unsigned long long active, const Arrived* l, const void*);
void mfma_f64_16x16x4_all(unsigned long long active, const Arrived* l, const void*);
void thunks_all(unsigned long long active, const Arrived* l, const void*);
struct Thunk { void (*run)(void*); void* ctx; };
// `f` of every lane that reached this point runs when the LAST of them arrives, in lane order: one instruction of a wave
template <int KIND, typename F> EMU_INL void lockstep(F&& f) {
  using Fn = typename std::remove_reference<F>::type;
  Thunk t{[](void* c) { (*static_cast<Fn*>(c))(); }, &f};
  wave_collective(KIND, &t, nullptr, thunks_all);
}

template <typename T> EMU_INL T shfl_from(T v, int src_lane) {  // src_lane: absolute lane 0..63 (or < 0: own value)
  ShflOp op{to_bits(v), src_lane};
  unsigned long long r = 0;
  wave_collective(K_SHFL, &op, &r, shfl_all);
  return from_bits<T>(r);
}
}  // namespace emu

// event counters of the kernels (emu_runtime.cpp); SCAMD_EMU_COUNT(i, n) in kernel code, a no-op in the product build
extern "C" long long emu_user_counters[16];

#define threadIdx (::emu::g_cur->tidx)
#define blockIdx (::emu::g_cur->bidx)
#define blockDim (::emu::g_cur->bdim)
#define gridDim (::emu::g_cur->gdim)
#define warpSize 64
#define hipLaunchKernelGGL(...) ::emu::launch_ggl(__VA_ARGS__)

static inline void __syncthreads() {
  ::emu::dma_land(0);  // (the fence of __syncthreads waits for vmcnt(0): pending LDS-DMA lands first)
  ::emu::block_barrier();
}
// barrier that leaves the lane's `keep` youngest LDS-DMA requests in flight / LDS traffic only (csrc: SCAMD_BARRIER_VM / _LDS)
static inline void emu_barrier_vm(int keep) {
  ::emu::dma_land(keep);
  ::emu::block_barrier();
}
static inline void emu_barrier_lds() { ::emu::block_barrier(); }
static inline unsigned long long wall_clock64() { return 0ull; }
static inline long long clock64() { return 0ll; }
static inline void __threadfence() {}
static inline void __threadfence_block() {}

// ---- shuffles (width = power of two <= 64; the sub-wave forms address lanes inside the caller's own segment)
template <typename T> EMU_INL T __shfl(T v, int src, int width = 64) {
  const int me = ::emu::g_cur->lane;
  return ::emu::shfl_from(v, (me & ~(width - 1)) | (src & (width - 1)));
}
template <typename T> EMU_INL T __shfl_xor(T v, int mask, int width = 64) {
  const int me = ::emu::g_cur->lane;
  const int s = me ^ mask;
  return ::emu::shfl_from(v, (s & ~(width - 1)) == (me & ~(width - 1)) ? s : me);
}
template <typename T> EMU_INL T __shfl_down(T v, unsigned d, int width = 64) {
  const int me = ::emu::g_cur->lane;
  const int s = me + (int)d;
  return ::emu::shfl_from(v, (s & ~(width - 1)) == (me & ~(width - 1)) ? s : me);
}
template <typename T> EMU_INL T __shfl_up(T v, unsigned d, int width = 64) {
  const int me = ::emu::g_cur->lane;
  const int s = me - (int)d;
  return ::emu::shfl_from(v, s >= 0 && (s & ~(width - 1)) == (me & ~(width - 1)) ? s : me);
}
EMU_INL unsigned long long __ballot(int pred) {
  int p = pred != 0;
  unsigned long long r = 0;
  ::emu::wave_collective(::emu::K_BALLOT, &p, &r, ::emu::ballot_all);
  return r;
}
EMU_INL int __any(int pred) { return __ballot(pred) != 0ull; }
EMU_INL int __all(int pred) { return __ballot(!pred) == 0ull; }

// ---- bit / conversion intrinsics
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline int __float_as_int(float f) { return ::emu::from_bits<int>(::emu::to_bits(f)); }
static inline unsigned __float_as_uint(float f) { return ::emu::from_bits<unsigned>(::emu::to_bits(f)); }
static inline float __int_as_float(int i) { return ::emu::from_bits<float>(::emu::to_bits(i)); }
static inline float __uint_as_float(unsigned i) { return ::emu::from_bits<float>(::emu::to_bits(i)); }
static inline long long __double_as_longlong(double d) { return ::emu::from_bits<long long>(::emu::to_bits(d)); }
static inline double __longlong_as_double(long long i) { return ::emu::from_bits<double>(::emu::to_bits(i)); }
// (the sources are compiled with -ffp-contract=off: one rounding per operation, as the _rn forms promise)
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fdiv_rn(float a, float b) { volatile float r = a / b; return r; }
static inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
static inline double __dadd_rn(double a, double b) { volatile double r = a + b; return r; }
static inline double __dmul_rn(double a, double b) { volatile double r = a * b; return r; }
static inline double __fma_rn(double a, double b, double c) { return fma(a, b, c); }
static inline float __fsqrt_rn(float x) { return sqrtf(x); }
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline double rsqrt(double x) { return 1.0 / sqrt(x); }
static inline float __saturatef(float x) { return x < 0.f ? 0.f : (x > 1.f ? 1.f : x); }

// HIP's ::min / ::max overloads
#define EMU_MINMAX(T) \
  static inline T min(T a, T b) { return b < a ? b : a; } \
  static inline T max(T a, T b) { return a < b ? b : a; }
EMU_MINMAX(int) EMU_MINMAX(unsigned) EMU_MINMAX(long) EMU_MINMAX(unsigned long) EMU_MINMAX(long long) EMU_MINMAX(unsigned long long)
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }
static inline double min(double a, double b) { return fmin(a, b); }
static inline double max(double a, double b) { return fmax(a, b); }
static inline long long min(long long a, int b) { return b < a ? b : a; }
static inline long long min(int a, long long b) { return b < a ? b : a; }
static inline long min(long a, int b) { return b < a ? b : a; }
static inline long min(int a, long b) { return b < a ? b : a; }
static inline long long max(long long a, int b) { return a < b ? b : a; }
static inline long max(long a, int b) { return a < b ? b : a; }
static inline long max(int a, long b) { return a < b ? b : a; }

// ---- atomics: one fibre runs at a time, a plain read-modify-write is atomic.
// The kernels let the lanes of ONE wave talk through LDS with no barrier in between (a lane clears its table slots with
// plain stores, every lane inserts with atomicCAS / atomicAdd, every lane reads slots with __hip_atomic_load): correct on
// the hardware because a wave executes its LDS instructions in program order for all lanes at once.  Fibres do not run
// in lock step, so every atomic is a LOCK-STEP POINT here: the lanes of the wave that reach it wait for each other
// first, and the operations of all of them are carried out at that moment, in lane order.  Lanes that took another path
// are not waited for, with one rule of precedence when the wave is stuck (emu_runtime.cpp): pending read-modify-writes
// are carried out before pending atomic LOADS, and those before shuffles -- the lanes that skipped a loop of inserts and
// wait at the load behind it see the table only after the lanes inside the loop are done, as on the hardware.
#define EMU_ATOMIC(NAME, BODY) \
  template <typename T, typename U> EMU_INL T NAME(T* p, U v) { T o; ::emu::lockstep<::emu::K_SOFT_RMW>([&]() { o = *p; BODY; }); return o; }
EMU_ATOMIC(atomicAdd, *p = (T)(o + (T)v))
EMU_ATOMIC(atomicSub, *p = (T)(o - (T)v))
EMU_ATOMIC(atomicMax, if ((T)v > o) *p = (T)v)
EMU_ATOMIC(atomicMin, if ((T)v < o) *p = (T)v)
EMU_ATOMIC(atomicOr, *p = (T)(o | (T)v))
EMU_ATOMIC(atomicAnd, *p = (T)(o & (T)v))
EMU_ATOMIC(atomicXor, *p = (T)(o ^ (T)v))
EMU_ATOMIC(atomicExch, *p = (T)v)
template <typename T, typename U, typename V> EMU_INL T atomicCAS(T* p, U cmp, V v) {
  T o;
  ::emu::lockstep<::emu::K_SOFT_RMW>([&]() { o = *p; if (o == (T)cmp) *p = (T)v; });
  return o;
}
template <typename T> EMU_INL T emu_atomic_load(T* p) {
  T o;
  ::emu::lockstep<::emu::K_SOFT_LOAD>([&]() { o = *p; });
  return o;
}
template <typename T, typename V> EMU_INL void emu_atomic_store(T* p, V v) { ::emu::lockstep<::emu::K_SOFT_RMW>([&]() { *p = (T)v; }); }
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#define __hip_atomic_load(p, order, scope) emu_atomic_load((p))
#define __hip_atomic_store(p, v, order, scope) emu_atomic_store((p), (v))
#define __hip_atomic_fetch_add(p, v, order, scope) atomicAdd((p), (v))

// ---- amdgcn builtins
namespace emu {
EMU_INL int readlane_i(int v, int lane) {
  ShflOp op{to_bits(v), lane};
  unsigned long long r = 0;
  wave_collective(K_READLANE, &op, &r, readlane_all);
  return from_bits<int>(r);
}
EMU_INL int readfirstlane_i(int v) {
  unsigned long long b = to_bits(v), r = 0;
  wave_collective(K_READFIRST, &b, &r, readfirst_all);
  return from_bits<int>(r);
}
// DPP control word -> source lane (or -1: no source, `old` / 0 is used); gfx9 encodings
static inline int dpp_source(int lane, int ctrl) {
  const int row = lane & ~15, r = lane & 15;
  if (ctrl >= 0x000 && ctrl <= 0x0ff) return (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);   // quad_perm
  if (ctrl >= 0x101 && ctrl <= 0x10f) { const int s = r + (ctrl & 15); return s < 16 ? row | s : -1; }  // row_shl
  if (ctrl >= 0x111 && ctrl <= 0x11f) { const int s = r - (ctrl & 15); return s >= 0 ? row | s : -1; }  // row_shr
  if (ctrl >= 0x121 && ctrl <= 0x12f) return row | ((r - (ctrl & 15)) & 15);                        // row_ror
  if (ctrl == 0x130) return lane + 1 < 64 ? lane + 1 : -1;                                          // wave_shl:1
  if (ctrl == 0x134) return (lane + 1) & 63;                                                        // wave_rol:1
  if (ctrl == 0x138) return lane - 1 >= 0 ? lane - 1 : -1;                                          // wave_shr:1
  if (ctrl == 0x13c) return (lane - 1) & 63;                                                        // wave_ror:1
  if (ctrl == 0x140) return row | (15 - r);                                                         // row_mirror
  if (ctrl == 0x141) return row | (r < 8 ? 7 - r : 23 - r);                                         // row_half_mirror
  if (ctrl == 0x142) return r == 15 || lane < 16 ? -2 : ((lane & ~15) - 1);                          // row_bcast:15 (-2: keep)
  if (ctrl == 0x143) return lane < 32 ? -2 : 31;                                                    // row_bcast:31
  fprintf(stderr, "emu: unsupported DPP control 0x%x\n", ctrl);
  abort();
}
EMU_INL int update_dpp_i(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  const int me = g_cur->lane;
  int s = dpp_source(me, ctrl);
  const bool enabled = ((row_mask >> (me >> 4)) & 1) && ((bank_mask >> ((me >> 2) & 3)) & 1);
  // all lanes take part in the exchange; a lane without a source reads its own value and discards it
  const int got = shfl_from(src, s >= 0 ? s : me);
  if (!enabled || s == -2) return old;
  if (s == -1) return bound_ctrl ? 0 : old;
  return got;
}
EMU_INL int ds_swizzle_i(int v, int pattern) {
  const int me = g_cur->lane;
  if (pattern & 0x8000) {  // quad-perm mode
    return shfl_from(v, (me & ~3) | ((pattern >> (2 * (me & 3))) & 3));
  }
  const int and_mask = pattern & 31, or_mask = (pattern >> 5) & 31, xor_mask = (pattern >> 10) & 31;
  const int j = (((me & 31) & and_mask) | or_mask) ^ xor_mask;
  return shfl_from(v, (me & 32) | j);
}
EMU_INL void wave_barrier() { wave_collective(K_WAVE_BARRIER, nullptr, nullptr, barrier_all); }

typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef double f64x4_t __attribute__((ext_vector_type(4)));
struct MfmaF32Op { float a, b; f32x16_t c; };
EMU_INL f32x16_t mfma_f32_32x32x2(float a, float b, f32x16_t c) {
  MfmaF32Op op{a, b, c};
  f32x16_t d;
  wave_collective(K_MFMA_F32_32X32X2, &op, &d, mfma_f32_32x32x2_all);
  return d;
}
struct MfmaBf16Op { unsigned short a[8], b[8]; f32x16_t c; };
template <typename V> EMU_INL f32x16_t mfma_bf16_32x32x16(V a, V b, f32x16_t c) {
  static_assert(sizeof(V) == 16, "8 bf16 per lane");
  MfmaBf16Op op;
  memcpy(op.a, &a, 16);
  memcpy(op.b, &b, 16);
  op.c = c;
  f32x16_t d;
  wave_collective(K_MFMA_BF16_32X32X16, &op, &d, mfma_bf16_32x32x16_all);
  return d;
}
struct MfmaF64Op { double a, b; f64x4_t c; };
EMU_INL f64x4_t mfma_f64_16x16x4(double a, double b, f64x4_t c) {
  MfmaF64Op op{a, b, c};
  f64x4_t d;
  wave_collective(K_MFMA_F64_16X16X4, &op, &d, mfma_f64_16x16x4_all);
  return d;
}
}  // namespace emu

#define __builtin_amdgcn_readlane(v, l) ::emu::readlane_i((v), (l))
#define __builtin_amdgcn_readfirstlane(v) ::emu::readfirstlane_i((v))
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) ::emu::update_dpp_i((old), (src), (ctrl), (rm), (bm), (bc))
#define __builtin_amdgcn_ds_swizzle(v, p) ::emu::ds_swizzle_i((v), (p))
#define __builtin_amdgcn_wave_barrier() ::emu::wave_barrier()
#define __builtin_amdgcn_sched_group_barrier(a, b, c) ((void)0)
#define __builtin_amdgcn_sched_barrier(a) ((void)0)
// (gfx9 encoding: vmcnt = bits 3:0 and 15:14 -- the lane's LDS-DMA requests beyond that many youngest have landed)
#define __builtin_amdgcn_s_waitcnt(a) ::emu::dma_land(((a) & 0xF) | ((((a) >> 14) & 3) << 4))
#define __builtin_amdgcn_s_setprio(a) ((void)0)
// LDS-DMA: lane l's `size` bytes land at the wave-uniform base + l * size (the destination pointer of the first lane is
// the base; every lane passes the same one)
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) \
  ::emu::dma_issue(reinterpret_cast<char*>(l) + (off) + ::emu::g_cur->lane * (size), reinterpret_cast<const char*>(g), (size))
#define __builtin_amdgcn_logf(x) log2f(x)
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_fmed3f(a, b, c) fmaxf(fminf((a), (b)), fminf(fmaxf((a), (b)), (c)))
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) ::emu::mfma_f32_32x32x2((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) ::emu::mfma_bf16_32x32x16((a), (b), (c))
#define __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, x, y, z) ::emu::mfma_f64_16x16x4((a), (b), (c))

// (glibc declares __expf / __logf itself: macros, after every libc header this file pulls in)
#define __expf(x) expf(x)
#define __logf(x) logf(x)
