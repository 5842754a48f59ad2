// Exact brute-force Euclidean kNN for gfx950 (MI355X).
//
// Replaces sklearn KNeighborsTransformer(algorithm='brute') as called by the reference at
// src/scanpy/neighbors/__init__.py:754-768 plus the self-column conventions of
// src/scanpy/neighbors/_common.py:74-98.  Three passes:
//
//   1. knn_select_kernel   FP32 MFMA (v_mfma_f32_32x32x2_f32): s(q,c) = ||c||^2 - 2 q.c for a
//      32-query x 32-candidate tile per instruction chain; every wave owns 32 queries (their
//      -2q fragments stay in VGPRs for the whole sweep) and streams ALL candidates through an
//      LDS-staged, double-buffered tile shared by the block's waves.  A per-query threshold
//      (current KP-th best) filters the 16 results each lane holds; survivors are inserted into a
//      KP-slot list in LDS.  KP > k, so the list is a superset of the answer unless float32
//      rounding interferes -- which pass 2 proves or disproves.
//   2. knn_rerank_kernel   float64 (q-c)^2 re-evaluation of the KP candidates, ordering by
//      (distance, index), explicit self column, and a certificate: the k-th exact distance must
//      lie below the final threshold by more than the float32 error bound of pass 1.
//   3. knn_fallback_scan/rank_kernel  float64 scan of all rows for the (rare) uncertified queries.
//
// Roofline: pass 1 is FP32-MFMA bound (2*n_query*n*2H flop); passes 2/3 are negligible.
#include "common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <vector>

static thread_local float g_last_select_ms = -1.f;
static thread_local double g_last_select_pairs = -1.0;
static thread_local double g_last_select_prepass_pairs = -1.0;
static thread_local int g_last_select_engine = -1;
static thread_local int g_last_second_tier = 0;
static thread_local int g_last_coarse = 0;  // 1: the last pruned sweep ran with the coarse first stage
static thread_local int g_last_nprobe = 0;  // cells probed by the last search on this thread (0: it was answered exactly)

namespace scamd {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using i32x4 = __attribute__((ext_vector_type(4))) int;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;

// destination operand of __builtin_amdgcn_global_load_lds (an LDS-address-space pointer; the host emulation of the kernels,
// tests/emu, has one address space)
#ifdef SCAMD_EMU
#define SCAMD_LDS_PTR(p) (p)
#else
#define SCAMD_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#endif
// event counters of the host emulation (tools/emu_knn_insertions.py counts what a query of the sweep costs in list
// insertions -- no hardware counter sees that); nothing in the product build
#ifdef SCAMD_EMU
#define SCAMD_EMU_COUNT(i, n) (emu_user_counters[i] += (n))
#else
#define SCAMD_EMU_COUNT(i, n) ((void)0)
#endif
// Workgroup barrier that leaves LDS-DMA requests in flight: `__syncthreads()` carries a fence that waits for vmcnt(0), which
// drains them.  SCAMD_BARRIER_VM(k): this wave's requests except the k youngest have landed (s_waitcnt vmcnt(k)), its LDS
// reads and writes are done (lgkmcnt(0)), then the barrier -- what the OTHER waves requested and waited for the same way is
// visible behind it.  SCAMD_BARRIER_LDS(): LDS traffic only.  The empty asm statements keep the compiler from moving memory
// accesses across (the barrier builtin alone is not a memory barrier to it).  gfx9 s_waitcnt immediate: vmcnt in bits 3:0
// (and 15:14), expcnt 6:4, lgkmcnt 11:8.  The host emulation of the tests (tests/emu/hip/hip_runtime.h: dma_issue) models
// both extremes of when a request may land.
#ifdef SCAMD_EMU
#define SCAMD_BARRIER_VM(k) emu_barrier_vm(k)
#define SCAMD_BARRIER_LDS() emu_barrier_lds()
#define SCAMD_WAIT_VM0() ::emu::dma_land(0)
#else
#define SCAMD_BARRIER_VM(k)                          \
  do {                                               \
    asm volatile("" ::: "memory");                   \
    __builtin_amdgcn_s_waitcnt(0x0070 | (k));        \
    __builtin_amdgcn_s_barrier();                    \
    asm volatile("" ::: "memory");                   \
  } while (0)
#define SCAMD_BARRIER_LDS()                          \
  do {                                               \
    asm volatile("" ::: "memory");                   \
    __builtin_amdgcn_s_waitcnt(0xC07F);              \
    __builtin_amdgcn_s_barrier();                    \
    asm volatile("" ::: "memory");                   \
  } while (0)
#define SCAMD_WAIT_VM0() __builtin_amdgcn_s_waitcnt(0x0F70)
#endif

constexpr int FALLBACK_CAP = 2048;    // collected rows per uncertified query
constexpr int FALLBACK_CHUNK = 1024;  // uncertified queries processed per launch

// ------------------------------------------------------------------------------------------------
// centring: the float32 pass works on x - mu (mu = column means, float32).  Euclidean distances do not change, the
// scores ||c||^2 - 2 q.c do: far from the origin (||x||^2 >> d^2) they lose every digit that matters -- at an offset
// of 300 per coordinate a float32 ulp of a score is 0.5 against squared neighbour distances of ~100, every query
// failed its certificate and went through the float64 scan.  fl(x - mu) is exact to 2^-24 relative per coordinate;
// that perturbation is part of the certificate's error bound (knn_rerank_kernel).  The exact passes (re-rank, fallback)
// read the caller's x.  Two stages, fixed summation order: the same mu for the same input, run to run.
// ------------------------------------------------------------------------------------------------
constexpr int MEAN_BLOCKS = 256;
constexpr int KNN_MAX_D = 256;  // widest row the search takes (stride of the column-mean tables)
__global__ __launch_bounds__(1024) void knn_colsum_kernel(const float* __restrict__ x, int64_t n, int d, int64_t ld,
                                                          double* __restrict__ partial /* [MEAN_BLOCKS][KNN_MAX_D] */) {
  // 1024 threads = 16 (8) rows at a time per workgroup, four independent loads in flight per thread (a 256-thread
  // workgroup with one load per thread in flight read the 200 MB of the 1M x 50 matrix at 0.5 TB/s)
  __shared__ double sh[1024];
  const int cols = d <= 64 ? 64 : (d <= 128 ? 128 : 256), rpar = 1024 / cols;
  const int c = threadIdx.x % cols, rr = threadIdx.x / cols;
  const int64_t chunk = (n + MEAN_BLOCKS - 1) / MEAN_BLOCKS;
  const int64_t r0 = (int64_t)blockIdx.x * chunk, r1 = std::min<int64_t>(n, r0 + chunk);
  double s = 0.0;
  if (c < d) {
    int64_t r = r0 + rr;
    for (; r + 3 * rpar < r1; r += 4 * rpar) {
      const float a0 = x[r * ld + c], a1 = x[(r + rpar) * ld + c], a2 = x[(r + 2 * rpar) * ld + c],
                  a3 = x[(r + 3 * rpar) * ld + c];
      s += (double)a0;
      s += (double)a1;
      s += (double)a2;
      s += (double)a3;
    }
    for (; r < r1; r += rpar) s += (double)x[r * ld + c];
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  if (rr == 0) {
    for (int j = 1; j < rpar; ++j) s += sh[j * cols + c];
    partial[(int64_t)blockIdx.x * KNN_MAX_D + c] = (c < d) ? s : 0.0;
  }
  // (columns beyond the ones this launch's layout covers: zero)
  if ((int)threadIdx.x >= cols && (int)threadIdx.x < KNN_MAX_D) partial[(int64_t)blockIdx.x * KNN_MAX_D + threadIdx.x] = 0.0;
}
__global__ __launch_bounds__(KNN_MAX_D) void knn_colmean_kernel(const double* __restrict__ partial, int64_t n, int d,
                                                                float* __restrict__ mu /* [KNN_MAX_D] */) {
  const int c = threadIdx.x;
  double s = 0.0;
  for (int b = 0; b < MEAN_BLOCKS; ++b) s += partial[(int64_t)b * KNN_MAX_D + c];
  mu[c] = (c < d && n > 0) ? (float)(s / (double)n) : 0.f;
}

// ------------------------------------------------------------------------------------------------
// pack: zero-padded [n_pad][DP] copy of x - mu, float32 squared norms (rounded once from float64), max norm
// ------------------------------------------------------------------------------------------------
__global__ void knn_pack_kernel(const float* __restrict__ x, const float* __restrict__ mu, int64_t n, int d, int64_t ld,
                                int DP, int64_t n_pad, float* __restrict__ xp, float* __restrict__ cn,
                                unsigned int* __restrict__ cmax_bits) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  float wmax = 0.f;
  for (int64_t r = wave; r < n_pad; r += nwaves) {
    double s = 0.0;
    for (int c = lane; c < DP; c += 64) {
      float v = (r < n && c < d) ? __fsub_rn(x[r * ld + c], mu[c]) : 0.f;
      xp[r * DP + c] = v;
      s += (double)v * (double)v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    float nf = (r < n) ? (float)s : INFINITY;  // pad rows can never be selected
    if (lane == 0) cn[r] = nf;
    if (r < n) wmax = fmaxf(wmax, nf);
  }
  if (lane == 0 && wmax > 0.f) atomicMax(cmax_bits, __float_as_uint(wmax));
}

// ------------------------------------------------------------------------------------------------
// pass 1
// ------------------------------------------------------------------------------------------------
template <int H, int TC, int NW, int KP>
struct SelectCfg {
  static constexpr int DP = 2 * H;
  // LDS row = [dims 0..H-1 | pad | dims H..2H-1 | pad]: each half starts 16-B aligned so a lane reads its
  // fragment with HP/4 ds_read_b128; row stride = 4 (mod 8) dwords keeps those reads bank-conflict free
  // (16-lane groups, 64 banks: stride*l mod 64 must hit 16 distinct 4-dword slots).
  static constexpr int HP = (H + 3) / 4 * 4;
  static constexpr int DPL = (2 * HP) % 8 == 4 ? 2 * HP : 2 * HP + 4;
  static constexpr int QB = NW * 32;
  static constexpr int NT = NW * 64;
  static constexpr int TILE_F4 = TC * DP / 4;
  static constexpr int F4_PER_THREAD = (TILE_F4 + NT - 1) / NT;
  static constexpr size_t LDS_BYTES = (size_t)(2 * TC * DPL + 2 * TC + 2 * KP * QB + 2 * QB) * 4;
  static_assert(TC <= NT, "thread t stages the norm of candidate t of a tile");
  static_assert(LDS_BYTES <= 160 * 1024, "tiles + lists must fit the 160 KB of LDS");
};

template <int H, int TC, int NW, int KP>
__global__ __launch_bounds__(NW * 64) void knn_select_kernel(const float* __restrict__ xp,
                                                             const float* __restrict__ cn,
                                                             int n_tiles, int64_t n_pad,
                                                             int64_t q_begin,
                                                             int* __restrict__ cand_idx,
                                                             float* __restrict__ cand_tau) {
  using C = SelectCfg<H, TC, NW, KP>;
  constexpr int DP = C::DP, DPL = C::DPL, HP = C::HP, QB = C::QB, NT = C::NT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* tile = smem;                   // [2][TC][DPL]
  float* tnorm = tile + 2 * TC * DPL;   // [2][TC]
  float* ld = tnorm + 2 * TC;           // [KP][QB] candidate keys s = ||c||^2 - 2 q.c
  int* li = (int*)(ld + KP * QB);       // [KP][QB] candidate row ids
  float* lmax = (float*)(li + KP * QB); // [QB] current threshold (max key in the list)
  int* lpos = (int*)(lmax + QB);        // [QB] slot holding that max

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int ql = wave * 32 + l31;

  // B operand: lane l holds query (l&31), dims [half*H, half*H+H), pre-scaled by -2.
  float bq[H];
  {
    int64_t qrow = q_begin + (int64_t)blockIdx.x * QB + ql;
    if (qrow > n_pad - 1) qrow = n_pad - 1;  // padded query slot: results are never read
    const float* qp = xp + qrow * DP + half * H;
#pragma unroll
    for (int s = 0; s < H; ++s) bq[s] = -2.0f * qp[s];
  }

  for (int e = tid; e < KP * QB; e += NT) {
    ld[e] = INFINITY;
    li[e] = -1;
  }
  if (tid < QB) {
    lmax[tid] = INFINITY;
    lpos[tid] = 0;
  }

  float4 stage[C::F4_PER_THREAD];
  float nstage = 0.f;
  auto gload = [&](int t) {
    const float4* src = reinterpret_cast<const float4*>(xp + (int64_t)t * TC * DP);
#pragma unroll
    for (int i = 0; i < C::F4_PER_THREAD; ++i) {
      int e = tid + i * NT;
      if (e < C::TILE_F4) stage[i] = src[e];
    }
    if (tid < TC) nstage = cn[(int64_t)t * TC + tid];
  };
  auto lstore = [&](int b) {
    float* dst = tile + b * TC * DPL;
#pragma unroll
    for (int i = 0; i < C::F4_PER_THREAD; ++i) {
      int e = tid + i * NT;
      if (e < C::TILE_F4) {
        int f = e * 4;
        float v[4] = {stage[i].x, stage[i].y, stage[i].z, stage[i].w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          int ff = f + c;
          int row = ff / DP, col = ff - row * DP;
          dst[row * DPL + (col < H ? col : col - H + HP)] = v[c];
        }
      }
    }
    if (tid < TC) tnorm[b * TC + tid] = nstage;
  };

  gload(0);
  lstore(0);
  __syncthreads();

  // A operand (lane l: candidate (l&31) of a 32-row sub-tile, same dim slice as B) and the C-in
  // registers (||c||^2 of the candidate each accumulator register belongs to: register r of lane l is
  // D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]) of one sub-tile.
  auto load_frag = [&](const float* tb, const float* nb, int sub, float (&a)[HP], f32x16& c0) {
    const float4* ap = reinterpret_cast<const float4*>(tb + (sub * 32 + l31) * DPL + half * HP);
#pragma unroll
    for (int s4 = 0; s4 < HP / 4; ++s4) {
      float4 v = ap[s4];
      a[4 * s4 + 0] = v.x;
      a[4 * s4 + 1] = v.y;
      a[4 * s4 + 2] = v.z;
      a[4 * s4 + 3] = v.w;
    }
#pragma unroll
    for (int a4 = 0; a4 < 4; ++a4) {
      float4 v = *reinterpret_cast<const float4*>(nb + sub * 32 + 8 * a4 + 4 * half);
      c0[4 * a4 + 0] = v.x;
      c0[4 * a4 + 1] = v.y;
      c0[4 * a4 + 2] = v.z;
      c0[4 * a4 + 3] = v.w;
    }
  };

  constexpr int NSUB = TC / 32;
  constexpr bool PF = H <= 32;
  float thr = INFINITY;
  for (int t = 0; t < n_tiles; ++t) {
    const int b = t & 1;
    if (t + 1 < n_tiles) gload(t + 1);
    const float* tb = tile + b * TC * DPL;
    const float* nb = tnorm + b * TC;
    float a_cur[HP];
    f32x16 c_cur;
    if (PF) load_frag(tb, nb, 0, a_cur, c_cur);
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
      // software pipeline (PF): the next sub-tile's fragments are in flight while this one's MFMA chain
      // runs; without it (d > 64, register budget) the fragments are fetched right before their chain
      float a_nxt[PF ? HP : 1];
      f32x16 c_nxt;
      if constexpr (PF) {
        if (sub + 1 < NSUB) load_frag(tb, nb, sub + 1, a_nxt, c_nxt);
      } else {
        load_frag(tb, nb, sub, a_cur, c_cur);
      }
      __builtin_amdgcn_sched_barrier(0);
      f32x16 acc = c_cur;
#pragma unroll
      for (int s = 0; s < H; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[s], bq[s], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);

      float m01 = fminf(fminf(acc[0], acc[1]), fminf(acc[2], acc[3]));
      float m23 = fminf(fminf(acc[4], acc[5]), fminf(acc[6], acc[7]));
      float m45 = fminf(fminf(acc[8], acc[9]), fminf(acc[10], acc[11]));
      float m67 = fminf(fminf(acc[12], acc[13]), fminf(acc[14], acc[15]));
      float m = fminf(fminf(m01, m23), fminf(m45, m67));
      if (__any(m < thr)) {
        // Rare path.  Lanes l and l+32 share a query: run the halves one after the other so the list
        // of a query is only ever touched by one lane at a time (the LDS ops of a wave retire in
        // order, and every access goes through the same LDS arrays so the compiler keeps their order).
        const int cbase = t * TC + sub * 32 + 4 * half;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          if (half == hh) {
            float lthr = lmax[ql];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float v = acc[r];
              if (v < lthr) {
                const int p = lpos[ql];
                ld[p * QB + ql] = v;
                li[p * QB + ql] = cbase + (r & 3) + 8 * (r >> 2);
                float mx = -INFINITY;
                int mp = 0;
#pragma unroll 1
                for (int u0 = 0; u0 < KP; u0 += 32) {
                  float vals[32];
#pragma unroll
                  for (int u = 0; u < 32; ++u) vals[u] = ld[(u0 + u) * QB + ql];
#pragma unroll
                  for (int u = 0; u < 32; ++u) {
                    const bool gt = vals[u] > mx;
                    mx = gt ? vals[u] : mx;
                    mp = gt ? (u0 + u) : mp;
                  }
                }
                lmax[ql] = mx;
                lpos[ql] = mp;
                lthr = mx;
              }
            }
          }
          __builtin_amdgcn_wave_barrier();
        }
        thr = lmax[ql];
      }
      if constexpr (PF) {
        if (sub + 1 < NSUB) {
#pragma unroll
          for (int s = 0; s < HP; ++s) a_cur[s] = a_nxt[s];
          c_cur = c_nxt;
        }
      }
    }
    if (t + 1 < n_tiles) lstore(b ^ 1);
    __syncthreads();
  }

  for (int e = tid; e < KP * QB; e += NT) {
    int q = e / KP, u = e - q * KP;
    cand_idx[((int64_t)blockIdx.x * QB + q) * KP + u] = li[u * QB + q];
  }
  if (tid < QB) cand_tau[(int64_t)blockIdx.x * QB + tid] = lmax[tid];
}

// ------------------------------------------------------------------------------------------------
// pass 1, register-list variant (k <= 24 -> KP = 32, d <= 64): the default path.
//
// Roles are transposed w.r.t. knn_select_kernel: the wave's 32 QUERIES are the A operand (rows of D), the 32
// candidates of a sub-tile are the B operand (columns of D).  Accumulator register r of lane l belongs to
// (query i(r, l>>5), candidate l&31) with i(r, h) = (r&3) + 8*(r>>2) + 4*h.
//   * The KP = 32 best (score, row id) pairs of query i(r, h) live in registers key[r] / idx[r] of the 32 lanes
//     of half h, sorted ascending along the lane index -- no LDS list, no per-lane serial scan.  A survivor is
//     inserted with one lane shift (ds_bpermute) + select per register, both halves at once.
//   * The filter costs 9 VALU per sub-tile: one extra MFMA k-pair (A = [-thr_i | 1], B = [1 | ||c_j||^2]) makes
//     the accumulator hold score - thr_i, so "some score beats its query's threshold" is the OR of 16 sign bits.
//     (On gfx950 the f32 MFMA shares the SIMD's FMA lanes with the VALU: every VALU instruction beside the chain
//     costs ~5 cycles of matrix throughput -- 16 adds + 16 compares + 16 scalar ORs cost 16 %, one more MFMA 4 %.)
//     thr_i is the thr_rank-th smallest key of query i, thr_rank = k + margin <= 32: a tighter threshold than
//     the list's last entry means fewer survivors; everything below it is still guaranteed to be in the list.
//   * One scheduling region per sub-tile: the MFMA chain of sub-tile g, the fragment reads of g+1 and the
//     filter of g-1 are interleaved 1 MFMA : 1 LDS read : <= 2 VALU (sched_group_barrier); only the branch to the
//     rare insertion sits between two chains.
// Candidates are streamed as a verbatim image of the LDS tile (row = [dims 0..H-1, 1.0, 0.. | dims H..2H-1,
// ||c||^2, 0.. | pad], stride DPL dwords = 4 mod 8 so the b128 fragment reads are conflict free), staged
// global_load_dwordx4 -> ds_write_b128, double buffered, one barrier per 64-candidate tile.  (An LDS-DMA
// variant measured the same: hipcc puts a vmcnt(0) in front of every ds_read that follows an LDS-DMA.)  LDS holds
// only the two tiles (30 KB) and the kernel fits 168 VGPRs, so three 4-wave blocks share a CU and cover each
// other's barrier and insertion stalls.  (Tried and dropped: wave-private sub-tiles without any barrier, 4x the
// L2->LDS traffic, 973 vs 828 ms; splitting the last partial round of the grid over candidate segments, no gain:
// a partially filled round already runs its blocks up to 3x faster.)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float readlane_f32(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
// v_writelane_b32 (value and lane uniform): clang has no builtin for it, the LLVM intrinsic is reached by its name
extern "C" __device__ int scamd_llvm_writelane(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");
// List keys of the register-list kernel: the float32 score with its 5 low mantissa bits replaced by the number of the
// slot (lane of idx[r], within the query's half) that holds the candidate's row id -- the keys stay sorted along the
// lanes, the row ids never move.  A key differs from its score by < 32 ulp (KEY_SLACK_ULPS, part of the certificate's
// error bound); empty entries are KEY_BIG | slot.
constexpr int KEY_SLOT_MASK = 31;
constexpr float KEY_BIG = 3.0e38f;
// float32 bit patterns -> integers whose signed order is the float order (scalar unit: gfx950 has no SALU float compare)
__device__ __forceinline__ int key_order(int bits) { return bits ^ ((bits >> 31) & 0x7fffffff); }

// B3 = the "3 x bf16" scoring engine (d <= 50 only).  The f32-input MFMA of gfx950 runs at the f32 VECTOR rate (157 TFLOP/s),
// the bf16 one sixteen times faster.  A float32 coordinate is written as hi + lo with hi = bf16(x), lo = bf16(x - hi)
// (residual <= 2^-16 |x|) and q.c ~ qh.ch + qh.cl + ql.ch: three bf16 products instead of one f32 product, 12
// v_mfma_f32_32x32x16_bf16 (32 cycles each) per 32 x 32 sub-tile instead of 26 v_mfma_f32_32x32x2_f32 (64 cycles each):
// 4.3x less matrix time for scores that are off by <= 3 * 2^-16 * 2 ||q|| ||c|| -- which the float64 certificate of pass 2
// prices in (CERT_K_B3, CERT_K2_B3), so the result is the exact kNN as before.  Image row = 68 dwords: 64 bf16 hi | 64 bf16 lo
// | ||c||^2 (f32, for the stop rule) | pad.  Dims 50..55 of the hi part carry the threshold subtraction and the norm:
// candidate rows hold [1, 1, 1, n1, n2, n3] (||c||^2 = n1 + n2 + n3 exactly: three bf16 pieces hold 24 bits), the query
// operand [-t1, -t2, -t3, 1, 1, 1] -- the accumulator comes out as score - threshold, with no extra instruction.
template <int H, int TC_ = 64, bool B3 = false>
struct RegCfg {
  static constexpr int HP = (H + 1 + 3) / 4 * 4;  // dims of one half + the extra k slot, rounded up to 4
  static constexpr int DPL = B3 ? 68 : ((2 * HP) % 8 == 4 ? 2 * HP : 2 * HP + 4);
  static constexpr int TC = TC_, SUBS = TC_ / 32, NW = 4, QB = 128, NT = 256, KP = 32;
  static constexpr int TILE_BYTES = TC * DPL * 4;
  static constexpr int TILE_KB = TILE_BYTES / 1024;
  // tile buffers in LDS: the float32 engine stages through registers into two; the bf16 engine's tiles arrive by LDS-DMA
  // into a ring of three, two requests in flight behind the tile being scored (round 4: see `sweep`)
  static constexpr bool GLDS = B3 && TC_ == 64;  // (the 128-candidate tiles of SCAMD_KNN_BIG_TILES keep the register staging)
  static constexpr int NBUF = GLDS ? 3 : 2;
  static constexpr size_t LDS_BYTES = NBUF * (size_t)TILE_BYTES;
  static_assert(TILE_BYTES % 1024 == 0, "tile must be a whole number of 1 KiB pieces");
  static_assert(!B3 || H == 25, "the bf16 engine is built for d <= 50");
};
constexpr int B3_DPL = 68;
// certificate factors (units of u = 2^-24), see knn_rerank_kernel.  bf16 engine: 198 accumulated terms instead of 52
// (+146 on both terms), and on the 2 q.c term the split's own error: bf16 carries 8 significant bits (unit roundoff 2^-8), so
// |x - hi - lo| <= 2^-16 |x| and the three dropped pieces (ql.cl, q's residual, c's residual) sum to <= 3 * 2^-16 = 768 u.
constexpr double CERT_K_F32 = 138.0, CERT_K_B3 = 284.0, CERT_K2_B3 = 768.0;
// ||c||^2 of padding rows: finite (0 * inf = NaN in the cross products) but ABOVE KEY_BIG, the "threshold" of a list that is
// not filled yet: the largest bf16 value (0x7F7F = 3.3895e38; one bf16 piece, the other two are 0).  With 1e38 (round 3) a pad
// row scored below KEY_BIG, passed the sign test of an unfilled list and was inserted -- harmless (its row id is filtered in
// pass 2) but insertion work in small cells, and exactness hung on 1e38 - 3e38 + 3e38 rounding the right way.
constexpr float B3_PAD_NORM = 3.3895313892515355e38f;

// float32 -> bf16 bits, round to nearest even (finite inputs)
__device__ __forceinline__ unsigned int bf16_rn(float v) {
  const unsigned int b = __float_as_uint(v);
  return (b + 0x7fffu + ((b >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf16_f32(unsigned int h) { return __uint_as_float(h << 16); }
// three bf16 pieces (by truncation) whose sum is v exactly
__device__ __forceinline__ void split3_bf16(float v, unsigned int& p1, unsigned int& p2, unsigned int& p3) {
  const float a = __uint_as_float(__float_as_uint(v) & 0xffff0000u);
  const float r1 = v - a;
  const float b = __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
  const float r2 = r1 - b;
  p1 = __float_as_uint(a) >> 16;
  p2 = __float_as_uint(b) >> 16;
  p3 = __float_as_uint(r2) >> 16;
}
// dword c (0..67) of the bf16 image row of a point with centred coordinates vc(dim) and squared norm nf
template <typename F>
__device__ __forceinline__ unsigned int b3_row_dword(int c, int d, float nf, F vc) {
  if (c >= 64) return c == 64 ? __float_as_uint(nf) : 0u;
  const bool lo = c >= 32;
  const int d0 = 2 * (c & 31);
  unsigned int w[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int dim = d0 + t;
    unsigned int v = 0u;
    if (dim < 50) {
      if (dim < d) {
        const float x = vc(dim);
        const unsigned int h = bf16_rn(x);
        v = lo ? bf16_rn(x - bf16_f32(h)) : h;
      }
    } else if (dim < 56 && !lo) {
      unsigned int n1, n2, n3;
      split3_bf16(nf, n1, n2, n3);
      v = dim < 53 ? 0x3F80u : (dim == 53 ? n1 : (dim == 54 ? n2 : n3));
    }
    w[t] = v;
  }
  return w[0] | (w[1] << 16);
}

// packed image of x for the register-list kernel: [n_pad][DPL] float32 (layout above); rows >= n get
// ||c||^2 = +inf so that they can never be selected.
__global__ void knn_pack_image_kernel(const float* __restrict__ x, const float* __restrict__ mu, int64_t n, int d,
                                      int64_t ld, int H, int HP, int DPL, int64_t n_pad, float* __restrict__ xp,
                                      unsigned int* __restrict__ cmax_bits, int b3) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  float wmax = 0.f;
  for (int64_t r = wave; r < n_pad; r += nwaves) {
    double s = 0.0;
    for (int c = lane; c < d; c += 64) {
      double v = (r < n) ? (double)__fsub_rn(x[r * ld + c], mu[c]) : 0.0;
      s += v * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float nf = (r < n) ? (float)s : INFINITY;
    if (b3) {  // bf16 hi / lo image (RegCfg: B3)
      const float nb = (r < n) ? nf : B3_PAD_NORM;
      unsigned int* xu = reinterpret_cast<unsigned int*>(xp);
      for (int c = lane; c < B3_DPL; c += 64)
        xu[r * B3_DPL + c] = b3_row_dword(c, d, nb, [&](int dim) { return r < n ? __fsub_rn(x[r * ld + dim], mu[dim]) : 0.f; });
      if (r < n) wmax = fmaxf(wmax, nf);
      continue;
    }
    for (int c = lane; c < DPL; c += 64) {
      const int hh = c / HP, cc = c - hh * HP;  // hh == 2: trailing pad
      float v = 0.f;
      if (hh < 2) {
        const int dim = hh * H + cc;
        if (cc < H) v = (r < n && dim < d) ? __fsub_rn(x[r * ld + dim], mu[dim]) : 0.f;
        else if (cc == H) v = (hh == 0) ? 1.0f : nf;  // B operand of the extra k-pair: [1 | ||c||^2]
      }
      xp[r * DPL + c] = v;
    }
    if (r < n) wmax = fmaxf(wmax, nf);
  }
  if (lane == 0 && wmax > 0.f) atomicMax(cmax_bits, __float_as_uint(wmax));
}

// Cell-pruned ("exact IVF") mode: rows are grouped by a coarse quantiser into cells, every cell padded to whole
// tiles in the candidate image and to whole 128-query blocks in the query list.  A block (all its queries in one
// cell a) visits the cells in order of increasing lower bound LB(a,b) = |c_a - c_b| - r_a - r_b on any distance
// between a member of a and a member of b, and stops at the first cell whose bound exceeds every query's current
// threshold distance: all remaining cells are farther still.  The answer is the exact kNN (pass 2 certifies it).
// More cells than this did not cut the evaluated pairs at 10M x 50 (isotropic 50-d blobs: the ball bound prunes at the
// granularity of a blob, not of a cell) and the quantiser grows with the cell count; the order tables are n_cells^2 x 8 B
constexpr int IVF_MAX_CELLS = 1024;
// entries of a block's cell order whose tile range and lower bound are copied to LDS when the block starts (the sweep loop
// then depends on no load but its tiles'; later entries are read from the tables)
#ifdef SCAMD_EMU
constexpr int IVF_META_CELLS = 4;   // (the host emulation of the tests refills often: its blocks sweep 5 - 16 cells)
#else
constexpr int IVF_META_CELLS = 64;
#endif
constexpr size_t IVF_META_BYTES = (size_t)IVF_META_CELLS * 3 * 4;

struct IvfArgs {
  const int* qpos;          // [n_blocks * 128] image row of every query slot (-1 = padding)
  const int* block_cell;    // [n_blocks]
  const int* cell_tile0;    // [n_cells] first tile of the cell in the image
  const int* cell_ntiles;   // [n_cells]
  const float* centers;     // [n_cells][dc]
  const float* radius;      // [n_cells] (already inflated for rounding)
  const int* order;         // [n_cells][n_cells] cells by ascending lower bound from cell a (own cell first)
  const float* order_lb2;   // [n_cells][n_cells] the squared lower bounds in that order (INF = empty cell)
  const int* perm;          // [n_image_rows] original row id of an image row (-1 = padding)
  unsigned long long* pairs; // (query, candidate) pairs evaluated, for the roofline figure
  const int* block_perm;     // [n_blocks] launch slot -> block: cells with the longest expected sweep first (LPT)
  int* qorder;               // [n_blocks * 128] out (may be null): query number (row - q_begin) of every query slot, -1 = padding
  int prepass_tiles;         // tiles of the own cell the threshold pre-pass scores (SCAMD_KNN_PREPASS_TILES, default 32 bf16 / 16 float32)
  int prepass_cells;         // cells (own cell first, then by ascending lower bound) the pre-pass covers (SCAMD_KNN_PREPASS_CELLS, default 1)
  int prepass_min2;          // 1: the starting threshold is taken from the two smallest scores per lane (SCAMD_KNN_PREPASS_MIN2, default 1)
  const unsigned int* cmax_bits;  // float bits of the largest ||c||^2 of the image (the COARSE sweep's slack, see knn_select_reg_block)
  int debug_no_insert;       // debug (SCAMD_KNN_DEBUG_NO_INSERT=1): survivors are dropped -- WRONG results, MFMA-side ceiling
  int cell_preload;          // 1: (bf16 engine) the tile requests run on into the next cell of the block's order while the
                             // current cell's last tiles are scored (SCAMD_KNN_CELL_PRELOAD, default 1; 0: every cell starts cold)
  unsigned long long* trace; // debug (SCAMD_KNN_TRACE=<file>): per block {start, end (100 MHz clock), tiles swept, hw id,
                             // end of the prologue, end of the pre-pass, ticks inside the sweeps of the cells, cells swept}
  int n_cells, dc;          // dc = stride of `centers` (>= d)
  int d;
  int* queue_ctr;           // persistent launch (null: one workgroup per launch slot): [8] next position of the queue of XCD x --
  int n_slots;              // launch slots x, x + 8, x + 16, ... of block_perm[0, n_slots); see knn_select_reg_kernel
};

// TC_ = candidates per LDS tile, WPS = resident blocks per CU (= waves per SIMD) the register budget is cut for
template <int HP>
struct BFragF32 {
  float v[HP];
};
struct BFragBf16 {
  i32x4 h[4], l[4];  // k-steps of 16 dims: 8 bf16 of the hi part / of the lo part per lane
};

// The bf16 engine's arithmetic, shared by the select kernel and by the debug entry that returns raw scores
// (scamd_knn_debug_b3_scores_f32: the test of the certificate's error bound measures THESE instructions).
// Query operand of lane (half, l31): k-step s holds dims 16 s + 8 half .. + 8 of -2 q as four bf16 pairs, hi part in qh,
// lo part in ql (the scaling by -2 is exact); dims 50..55 of the query side are [-t1, -t2, -t3, 1, 1, 1] (threshold 0 here).
__device__ __forceinline__ void b3_query_operand(const float* __restrict__ xp, int64_t qrow, int half, i32x4 (&qh)[4],
                                                 i32x4 (&ql)[4]) {
  const i32x4* qp = reinterpret_cast<const i32x4*>(xp + qrow * B3_DPL);
  auto neg2 = [](int w) {
    const float a = -2.0f * __uint_as_float((unsigned int)w << 16), b = -2.0f * __uint_as_float((unsigned int)w & 0xffff0000u);
    return (int)((__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xffff0000u));
  };
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const i32x4 h4 = qp[2 * s + half], l4 = qp[8 + 2 * s + half];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      qh[s][j] = neg2(h4[j]);
      ql[s][j] = neg2(l4[j]);
    }
  }
  if (half == 0) {
    qh[3][1] = 0;                      // dims 50, 51: -t1, -t2 (threshold 0 until the lists are filled)
    qh[3][2] = 0x3F800000;             // dims 52, 53: -t3, 1
    qh[3][3] = 0x3F803F80;             // dims 54, 55: 1, 1
  }
}
// lane exchanges of the pre-pass's sorting network that stay in the VALU (DPP control words: quad_perm 0x00-0xFF,
// row_ror:8 0x128, row_mirror 0x140, row_half_mirror 0x141) or cross a row of 16 lanes (ds_swizzle, bit mode: and 0x1F,
// xor << 10)
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// lane i ^ 4: lanes 0-3 / 8-11 of a row (banks 0, 2) read four lanes up (row_shl:4), lanes 4-7 / 12-15 four lanes down
__device__ __forceinline__ float dpp_xor4_f32(float v) {
  int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x104, 0xf, 0x5, false);
  t = __builtin_amdgcn_update_dpp(t, __float_as_int(v), 0x114, 0xf, 0xa, false);
  return __int_as_float(t);
}
template <int PATTERN>
__device__ __forceinline__ float swizzle_f32(float v) {
  return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), PATTERN));
}
// scores of one 32 x 32 sub-tile minus the thresholds: qh.ch, then qh.cl, then ql.ch, 4 k-steps of 16 dims each
__device__ __forceinline__ f32x16 b3_chain(const i32x4 (&qh)[4], const i32x4 (&ql)[4], const BFragBf16& b) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, qh[s]), __builtin_bit_cast(bf16x8, b.h[s]), acc, 0, 0, 0);
#pragma unroll
  for (int s = 0; s < 4; ++s)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, qh[s]), __builtin_bit_cast(bf16x8, b.l[s]), acc, 0, 0, 0);
#pragma unroll
  for (int s = 0; s < 4; ++s)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ql[s]), __builtin_bit_cast(bf16x8, b.h[s]), acc, 0, 0, 0);
  return acc;
}

// the two stages of the COARSE sweep (round 6): hi.hi first, then -- only for sub-tiles in which some coarse score passed the
// widened threshold -- hi.lo and lo.hi on top of it: the same twelve instructions in the same order as b3_chain
__device__ __forceinline__ f32x16 b3_chain_hh(const i32x4 (&qh)[4], const BFragBf16& b) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, qh[s]), __builtin_bit_cast(bf16x8, b.h[s]), acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ f32x16 b3_chain_rest(const i32x4 (&qh)[4], const i32x4 (&ql)[4], const BFragBf16& b, f32x16 acc) {
#pragma unroll
  for (int s = 0; s < 4; ++s)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, qh[s]), __builtin_bit_cast(bf16x8, b.l[s]), acc, 0, 0, 0);
#pragma unroll
  for (int s = 0; s < 4; ++s)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ql[s]), __builtin_bit_cast(bf16x8, b.h[s]), acc, 0, 0, 0);
  return acc;
}

// one block of the search: 128 queries (brute force: queries blk * 128 ..; pruned sweep: the query slots of block blk)
// COARSE (bf16 engine, pruned-sweep kernel; chosen by the host when the cell bounds prune little, i.e. a query meets >= 1e5
// candidates): a sub-tile is first scored with the hi.hi product alone -- 4 of the 12 MFMAs -- against the threshold WIDENED
// by `slack` >= |hi.lo + lo.hi| (bf16 round-to-nearest: |x_lo| <= 2^-8 |x| per coordinate, the query operand is -2 q, so the two
// dropped products are bounded by 2 * 2 * 2^-8 (1 + 2^-8) ||q|| ||c|| <= 2^-6 * 1.02 ||q||_max(wave) ||c||_max(image)); only a
// sub-tile with a coarse survivor gets the other 8 MFMAs, the slack is taken off again and the exact sign test / insertion
// follow as in the plain kernel.  A candidate whose full score is below the threshold has a coarse score below threshold +
// slack: the lists are the plain kernel's lists.  With 1M candidates per query ~10 % of the sub-tiles are refined.
template <int H, int TC_, int WPS, bool IVF, bool B3, bool COARSE = false>
__device__ __forceinline__ void knn_select_reg_block(const int blk, const float* __restrict__ xp, int n_tiles_all,
                                                     int64_t n_pad, int64_t q_begin, int thr_rank,
                                                     int* __restrict__ cand_idx, float* __restrict__ cand_tau,
                                                     const IvfArgs& iv) {
  static_assert(!COARSE || (B3 && IVF), "the coarse first stage belongs to the bf16 engine's pruned sweep");
  using C = RegCfg<H, TC_, B3>;
  constexpr int HP = C::HP, DPL = C::DPL, TC = C::TC, SUBS = C::SUBS;
  using BFrag = std::conditional_t<B3, BFragBf16, BFragF32<HP>>;
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][TC][DPL] (+ IVF: wmax[4])
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int thr_lane = thr_rank - 1;

  // A operand: lane l holds query (l&31), dims [half*H, half*H+H), pre-scaled by -2
  // (B3: k-step s, lane half h: dims 16 s + 8 h .. + 8 as four bf16 pairs, hi part in qh, lo part in ql)
  float aq[B3 ? 1 : H];
  i32x4 qh[B3 ? 4 : 1], ql[B3 ? 4 : 1];
  int64_t qrow;          // image row of this lane's query
  bool qvalid = true;
  if constexpr (IVF) {
    const int p = iv.qpos[(int64_t)blk * C::QB + wave * 32 + l31];
    qvalid = p >= 0;
    qrow = qvalid ? p : 0;
  } else {
    qrow = q_begin + (int64_t)blk * C::QB + wave * 32 + l31;
    if (qrow > n_pad - 1) qrow = n_pad - 1;  // padded query slot: results are never read
  }
  if constexpr (!B3) {
    const float* qp = xp + qrow * DPL + half * HP;
#pragma unroll
    for (int s = 0; s < H; ++s) aq[s] = -2.0f * qp[s];
  } else {
    b3_query_operand(xp, qrow, half, qh, ql);
  }

  // A operand of the extra k-pair: lanes 0..31 hold -thr of query (l&31), lanes 32..63 hold 1.0.
  // Until the lists are filled (sub-tile 0) the "threshold" is 0, i.e. the accumulator is the plain score.
  float athr = half ? 1.0f : 0.0f;
  float slack = 0.f;  // COARSE: what the operand's threshold is widened by (0 during the pre-pass: plain scores)
  // B3: the threshold lives in dims 50..52 of the query operand as three bf16 pieces (their sum is -thr exactly); `athr`
  // stays the float32 master copy, this re-derives the pieces of all 32 queries of the wave after it changed
  auto sync_thr = [&]() {
    if constexpr (B3) {
      unsigned int t1, t2, t3;
      split3_bf16(COARSE ? athr - slack : athr, t1, t2, t3);
      if (half == 0) {
        qh[3][1] = (int)(t1 | (t2 << 16));
        qh[3][2] = (int)(t3 | 0x3F800000u);
      }
    }
  };
  float key[16];
  int idx[16];
  // empty list: KEY_BIG with ascending slot numbers (ascending keys along the lanes of each half)
  const float key_empty = __int_as_float((__float_as_int(KEY_BIG) & ~KEY_SLOT_MASK) | l31);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    key[r] = key_empty;
    idx[r] = -1;
  }

  int n_sub = 0;  // sub-tiles of the current sweep
  // cell-pruned mode: `minima` = the pre-pass over the own cell (key[r] collects, per lane, the smallest score of the
  // candidates j = lane (mod 32): 32 distinct candidates per query); the threshold derived from it seeds the list,
  // which the list threshold can only tighten (same lane layout as athr)
  bool minima = false;
  // LDS buffer of tile t of the current sweep: the two buffers alternate / the ring of three goes round across sweeps
  constexpr bool GLDS = C::GLDS;
  int ring = 0;
  auto buf_of = [&](int t) -> int { return GLDS ? (ring + t) % 3 : (t & 1); };
  // B operand of sub-tile g of the current sweep: lane l holds candidate (l&31), the same dim slice as A, then
  // the extra k slot
  auto load_b = [&](int g, BFrag& b) {
    const float* tb = smem + buf_of(g / SUBS) * TC * DPL;
    if constexpr (B3) {
      // row = 17 16-byte units: hi part units 0..7, lo part 8..15 (row stride 68 dwords = 4 mod 64: conflict free)
      const i32x4* p = reinterpret_cast<const i32x4*>(tb + ((g % SUBS) * 32 + l31) * DPL);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        b.h[s] = p[2 * s + half];
        b.l[s] = p[8 + 2 * s + half];
      }
    } else {
      const float4* p = reinterpret_cast<const float4*>(tb + ((g % SUBS) * 32 + l31) * DPL + half * HP);
#pragma unroll
      for (int s4 = 0; s4 < HP / 4; ++s4) {
        const float4 v = p[s4];
        b.v[4 * s4 + 0] = v.x;
        b.v[4 * s4 + 1] = v.y;
        b.v[4 * s4 + 2] = v.z;
        b.v[4 * s4 + 3] = v.w;
      }
    }
  };
  // scores of one sub-tile minus the thresholds (the MFMA chain)
  auto chain = [&](const BFrag& b, float athr_op) -> f32x16 {
    if constexpr (B3) {
      (void)athr_op;
#ifdef SCAMD_KNN_PROBE_HH
      // PROBE BUILD ONLY (tools/knn_coarse_probe.sh compiles a second library with this macro; WRONG results): the hi.hi product
      // alone -- what a coarse first stage of the sweep would cost.  (As a RUN-TIME switch on `iv` this branch made the
      // production kernel 4.3 x slower -- 50.1 instead of 11.7 ms: DESIGN.md section 8, round 6.)
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, qh[s]), __builtin_bit_cast(bf16x8, b.h[s]), acc, 0, 0, 0);
      return acc;
#else
      return b3_chain(qh, ql, b);
#endif
    } else {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[0], b.v[0], acc, 0, 0, 0);
#pragma unroll
      for (int s = 1; s < H; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[s], b.v[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(athr_op, b.v[H], acc, 0, 0, 0);
      return acc;
    }
  };
  // Insert the survivors of a sub-tile.  acc[r] = score - (threshold its chain used); that threshold is lane
  // i(r,h) of `athr_used` (negated).  all = true (very first sub-tile): every finite score is inserted.
  // The two halves of a register (two queries) are handled one after the other, so every operand of an insertion is a
  // scalar: per survivor 2 readlanes (score, largest key), 1 writelane (row id into the evicted entry's slot) and
  // DPP shift + lane-0 fix + v_med3 on the keys -- new_key[l] = med3(key[l-1], key[l], v) IS the sorted insertion
  // (key[l-1] <= key[l]): v below both -> the left neighbour moves up, between -> v lands here, above -> unchanged.
  // The row ids stay where they are (slot = low 5 key bits).  On gfx950 the f32 MFMA shares the VALU lanes, and a
  // streaming top-k list takes ~100 insertions per query at 1M rows: this path, not the filter, is what the MFMA
  // stream competes with (round 1: 16 VALU per survivor + 15 per register with a hit, 147 VALU per sub-tile in all).
  auto insert_half = [&](const f32x16& acc, float athr_used, int cbase, const int r, const int h, unsigned int bits) {
    const int ih = (r & 3) + 8 * (r >> 2) + 4 * h;      // this half's query = its lane of the threshold operand
    const float tu = -readlane_f32(athr_used, ih);
    const float sc = acc[r] + tu;                        // the float32 score again (+- 1 ulp)
    if (lane == 0) SCAMD_EMU_COUNT(2, 1);                // [2] (register, half) groups with a survivor
    do {
      const int s = __builtin_ctz(bits);
      if (lane == 0) SCAMD_EMU_COUNT(0, 1);              // [0] survivors of the sign test
      const int vb = __builtin_amdgcn_readlane(__float_as_int(sc), 32 * h + s);
      const int lastb = __builtin_amdgcn_readlane(__float_as_int(key[r]), 32 * h + 31);
      // (a survivor of a threshold one sub-tile old may no longer beat the list's largest entry)
      if (__builtin_expect(key_order(vb) < key_order(lastb), 1)) {  // (the likely path laid out in line: a taken branch costs as much as four of these instructions)
        if (lane == 0) SCAMD_EMU_COUNT(1, 1);            // [1] insertions
        const int slot = lastb & KEY_SLOT_MASK;          // the evicted entry's slot is reused
        const float kv = __int_as_float((vb & ~KEY_SLOT_MASK) | slot);
        idx[r] = scamd_llvm_writelane(cbase + s, 32 * h + slot, idx[r]);
        // lane l-1's key by DPP wave_shr:1; lanes 0 and 32 have no left neighbour
        float upk = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(key[r]), 0x138, 0xf, 0xf, true));
        upk = (l31 == 0) ? -INFINITY : upk;
        const float nk = __builtin_amdgcn_fmed3f(upk, key[r], kv);
        key[r] = (half == h) ? nk : key[r];
      }
      bits &= bits - 1;
    } while (bits);
    // new threshold of this query -> its lane of the A operand of the extra k-pair
    const float t = readlane_f32(key[r], 32 * h + thr_lane);
    athr = __int_as_float(scamd_llvm_writelane(__float_as_int(-t), ih, __float_as_int(athr)));
  };
  auto insert = [&](const f32x16& acc, float athr_used, int cbase, bool all) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const unsigned long long m = all ? __ballot(acc[r] < KEY_BIG) : __ballot(acc[r] < 0.f);  // (all: plain scores; pad rows score >= KEY_BIG)
      if (m) {
        const unsigned int lo = (unsigned int)m, hi = (unsigned int)(m >> 32);
        if (lo) insert_half(acc, athr_used, cbase, r, 0, lo);
        if (hi) insert_half(acc, athr_used, cbase, r, 1, hi);
      }
    }
    sync_thr();
  };
  int row0 = 0;  // image row of the current sweep's first candidate
  // One pipeline step = ONE scheduling region: chain of sub-tile g into acc_cur (with the thresholds in athr),
  // fragment reads of sub-tile g+1, sign test of the previous sub-tile's accumulator.
  // COARSE: the second stage of a sub-tile whose coarse scores (in `acc`, taken against thr + slack) hold a survivor: the other
  // eight MFMAs on its fragments `bf`, the slack off again, the exact sign test and the insertions
  auto refine = [&](f32x16& acc, const BFrag& bf, float athr_used, int cbase) {
    if constexpr (COARSE) {
      acc = b3_chain_rest(qh, ql, bf, acc);
      bool neg = false;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[r] += slack;  // score - thr again
        neg |= acc[r] < 0.f;
      }
      if (__any(neg)) insert(acc, athr_used, cbase, false);
    }
  };
  auto coarse_hit = [&](const f32x16& acc) -> bool {
    const int c0 = __float_as_int(acc[0]) | __float_as_int(acc[1]) | __float_as_int(acc[2]);
    const int c1 = __float_as_int(acc[3]) | __float_as_int(acc[4]) | __float_as_int(acc[5]);
    const int c2 = __float_as_int(acc[6]) | __float_as_int(acc[7]) | __float_as_int(acc[8]);
    const int c3 = __float_as_int(acc[9]) | __float_as_int(acc[10]) | __float_as_int(acc[11]);
    const int c4 = __float_as_int(acc[12]) | __float_as_int(acc[13]) | __float_as_int(acc[14]);
    const int c5 = (c0 | c1 | c2) | (c3 | c4 | __float_as_int(acc[15]));
    return __any(c5 < 0);
  };
  auto step = [&](int g, BFrag& b_cur, f32x16& acc_cur, float& athr_cur, const f32x16& acc_prev,
                  float athr_prev, BFrag& b_nxt) {
    load_b(min(g + 1, n_sub - 1), b_nxt);  // (the clamp re-reads the last sub-tile: no branch in the region)
    athr_cur = athr;
    if constexpr (COARSE) {
      if (!minima) {
        // no deferral here: the coarse chain is short, the other two waves of the SIMD fill its latency.  The step works IN
        // acc_cur (no second accumulator: the kernel's register budget has none to spare; the PIPELINED form -- coarse chain of
        // sub-tile g issued, then sub-tile g - 1 tested and refined from the fragments b_nxt still holds -- keeps two accumulators
        // and two fragment sets live across the refinement: 1408 B of scratch per lane and 2.2 s instead of 0.26 s, measured);
        // the deferred tests of the plain pipeline are skipped in this mode (here, and at the end of the sweep)
        acc_cur = b3_chain_hh(qh, b_cur);
        if (coarse_hit(acc_cur)) {
          if (iv.debug_no_insert) return;
          refine(acc_cur, b_cur, athr_cur, row0 + g * 32);
        }
        return;
      }
    }
    acc_cur = chain(b_cur, athr_cur);
    // sign test: OR of the 16 accumulators of the previous sub-tile (3-input ORs)
    const int o0 = __float_as_int(acc_prev[0]) | __float_as_int(acc_prev[1]) | __float_as_int(acc_prev[2]);
    const int o1 = __float_as_int(acc_prev[3]) | __float_as_int(acc_prev[4]) | __float_as_int(acc_prev[5]);
    const int o2 = __float_as_int(acc_prev[6]) | __float_as_int(acc_prev[7]) | __float_as_int(acc_prev[8]);
    const int o3 = __float_as_int(acc_prev[9]) | __float_as_int(acc_prev[10]) | __float_as_int(acc_prev[11]);
    const int o4 = __float_as_int(acc_prev[12]) | __float_as_int(acc_prev[13]) | __float_as_int(acc_prev[14]);
    const int o5 = (o0 | o1 | o2) | (o3 | o4 | __float_as_int(acc_prev[15]));
    const bool hit = __any(o5 < 0);
    // pin the interleave: 1 MFMA, then 1 LDS read + up to 2 VALU in its shadow
#pragma unroll
    for (int s = 0; s < (B3 ? 12 : H + 1); ++s) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
    }
    if constexpr (IVF) {
      if (minima) {
        if (lane == 0) SCAMD_EMU_COUNT(5, 1);  // [5] sub-tiles of the threshold pre-pass
#pragma unroll
        for (int r = 0; r < 16; ++r) {  // the two smallest scores of the lane's candidate class: (key[r], idx[r] as float bits)
          idx[r] = __float_as_int(__builtin_amdgcn_fmed3f(key[r], __int_as_float(idx[r]), acc_prev[r]));
          key[r] = fminf(key[r], acc_prev[r]);
        }
        return;
      }
    }
    if constexpr (IVF) {
      if (iv.debug_no_insert) return;
    }
    if (lane == 0) {
      SCAMD_EMU_COUNT(3, 1);              // [3] sub-tiles tested by a wave (32 queries x 32 candidates)
      if (hit) SCAMD_EMU_COUNT(4, 1);     // [4] ... with at least one survivor
    }
    if (hit) insert(acc_prev, athr_prev, row0 + (g - 1) * 32, false);
  };

  // staging registers: every wave moves NP 1-KiB pieces per tile; out-of-range piece ids are clamped (a duplicate
  // copy of the last piece) so that no load sits behind a branch (hipcc waits vmcnt(0) after a conditional load)
  constexpr int NP = (C::TILE_KB + C::NW - 1) / C::NW;
  f32x4 st[GLDS ? 1 : NP];
  int tile0 = 0;  // first tile of the current sweep
  auto gload = [&](int t) {
    if constexpr (!GLDS) {
      const f32x4* src = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(xp) + (int64_t)(tile0 + t) * C::TILE_BYTES) + lane;
#pragma unroll
      for (int j = 0; j < NP; ++j) st[j] = src[min(wave + C::NW * j, C::TILE_KB - 1) * 64];
    }
  };
  auto lstore = [&](int buf) {
    if constexpr (!GLDS) {
      f32x4* dst = reinterpret_cast<f32x4*>(reinterpret_cast<char*>(smem) + buf * C::TILE_BYTES) + lane;
#pragma unroll
      for (int j = 0; j < NP; ++j) dst[min(wave + C::NW * j, C::TILE_KB - 1) * 64] = st[j];
    }
  };
  // ---- bf16 engine: tiles by LDS-DMA into a ring of three buffers (round 4) ----
  // The per-block trace of the pruned sweep (tools/knn_trace.py, profiles/r04r_*) put a block at 1.4-1.5 us per tile with the
  // insertions dropped, against 0.8 us in the pre-pass over its own cell: with register staging a tile is requested at the
  // barrier before the one it is stored at -- ONE request in flight per block, one tile per memory round trip when the tile
  // comes from beyond L2 (other cells), the matrix pipe idle for the difference.  Here the image of a tile (global and LDS
  // layouts are the same bytes) is requested by `global_load_lds_dwordx4` (1 KiB per wave instruction, NP per wave and tile)
  // THREE tiles ahead of the one being scored: requested after barrier t - 3, waited for (this wave's NP youngest requests
  // may stay outstanding) before barrier t - 1, read after it.  The stream does not stop at the end of a cell: the caller
  // announces the next cell of the block's order (`next_tile0`, `next_n`) and its first tiles are requested behind this
  // cell's last ones -- speculatively, the stopping rule may end the block first (the requests are drained before the
  // block writes its results).  No staging registers, no ds_write pass.
  int n_ahead = 0;                  // leading tiles of the sweep about to start that the previous sweep requested
  int next_tile0 = -1, next_n = 0;  // the cell the caller sweeps next (-1: none announced)
  auto request = [&](int tile, int buf) {
    if constexpr (GLDS) {
      static_assert(!GLDS || 2 * NP < 16, "the counted waits use the low vmcnt field only");
      const char* src = reinterpret_cast<const char*>(xp) + (int64_t)tile * C::TILE_BYTES + lane * 16;
      char* dst = reinterpret_cast<char*>(smem) + buf * C::TILE_BYTES;
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const int piece = min(wave + C::NW * j, C::TILE_KB - 1);  // (the clamp repeats the last piece: no branch)
        __builtin_amdgcn_global_load_lds(src + piece * 1024, SCAMD_LDS_PTR(dst + piece * 1024), 16, 0, 0);
      }
    }
  };

  // Sweep over `n_tiles` consecutive tiles of the image starting at tile `t0`.  first = the block's very first
  // sweep: its sub-tile 0 fills the lists (every finite score is inserted).  All four waves call it together; on
  // entry nobody reads the LDS tiles any more (the caller's barrier / kernel start guarantees it).
  auto sweep = [&](int t0, int n_tiles, bool first) {
    tile0 = t0;
    row0 = t0 * TC;
    n_sub = n_tiles * SUBS;
    // (GLDS) stream position s of this sweep: its own tile s, or tile s - n_tiles of the announced next cell
    auto exists = [&](int sp) -> bool { return sp < n_tiles || (next_tile0 >= 0 && sp - n_tiles < next_n); };
    auto req = [&](int sp) {
      if (exists(sp)) request(sp < n_tiles ? tile0 + sp : next_tile0 + (sp - n_tiles), (ring + sp) % 3);
    };
    if constexpr (GLDS) {
      static_assert(!GLDS || SUBS == 2, "one barrier per tile");
      if (n_ahead == 0) {
        // nothing of this sweep is under way (the buffers are free: the caller's barrier): positions 0 .. 2, then tile 0
        req(0);
        req(1);
        req(2);
        if (exists(2)) SCAMD_BARRIER_VM(2 * NP);
        else if (exists(1)) SCAMD_BARRIER_VM(NP);
        else SCAMD_BARRIER_VM(0);
      } else {
        // tile 0 landed before the previous sweep's last barrier; what it did not request of positions 1 and 2 goes out now
        if (n_ahead < 2) req(1);
        if (n_ahead < 3) req(2);
      }
    } else {
      gload(0);
      lstore(0);
      if (n_tiles > 1) gload(1);
      __syncthreads();
    }
    BFrag bA, bB;
    f32x16 accA, accB;
    float athrA = athr, athrB = athr;
    load_b(0, bA);
    if (!IVF && first) {
      // sub-tile 0: plain scores (threshold 0), every finite one is inserted; afterwards all thresholds are
      // finite whenever the sub-tile holds at least thr_rank real rows
      accB = chain(bA, athr);
      insert(accB, athr, row0, true);
      load_b(1 < n_sub ? 1 : 0, bB);  // sub-tile 0 is done: prefetch sub-tile 1
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) accB[r] = INFINITY;  // "previous" scores of the first pipeline step: no hit
    accA = accB;
    athrA = athrB;
    // half-tile loop (two sub-tiles per iteration, roles of the A/B register sets fixed: the insertion code exists
    // twice only); the barrier of tile t sits before its last sub-tile, whose fragment is already in registers
    constexpr int HPT = SUBS / 2;
    for (int h = 0; h < HPT * n_tiles; ++h) {
      const int g = 2 * h;
      if (IVF || h > 0 || !first) step(g, bA, accA, athrA, accB, athrB, bB);
      if ((h % HPT) == HPT - 1) {
        const int t = h / HPT;
        if constexpr (GLDS) {
          // position t + 1 has landed once this wave's requests for position t + 2 are the only ones outstanding; every wave
          // has completed its reads of tile t except the last sub-tile (in bB): behind the barrier position t + 1 is
          // visible and tile t's buffer takes position t + 3
          if (exists(t + 2)) SCAMD_BARRIER_VM(NP);
          else SCAMD_BARRIER_VM(0);
          req(t + 3);
        } else {
          if (t + 1 < n_tiles) lstore((t + 1) & 1);  // buffer of tile t-1: free since the previous barrier
          // every wave has completed its reads of tile t except the last sub-tile (already in bB): after the barrier
          // tile t+1 is visible and the staging registers are free for tile t+2
          __syncthreads();
          if (t + 2 < n_tiles) gload(t + 2);
        }
      }
      step(g + 1, bB, accB, athrB, accA, athrA, bA);
    }
    if (!(COARSE && !minima)) {  // (the coarse sweep has tested and inserted inside its steps)
      bool neg = false;
#pragma unroll
      for (int r = 0; r < 16; ++r) neg |= accB[r] < 0.f;
      if (IVF && minima) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          idx[r] = __float_as_int(__builtin_amdgcn_fmed3f(key[r], __int_as_float(idx[r]), accB[r]));
          key[r] = fminf(key[r], accB[r]);
        }
      } else if (__any(neg)) {
        insert(accB, athrB, row0 + (n_sub - 1) * 32, false);
      }
    }
    if constexpr (GLDS) {  // the ring goes on where this sweep ends; positions n_tiles .. n_tiles + 2 are under way
      ring = (ring + n_tiles) % 3;
      n_ahead = next_tile0 >= 0 ? min(3, next_n) : 0;
    }
  };
  // barrier between sweeps (LDS traffic only in the bf16 engine: requests for the next cell's tiles stay in flight)
  auto block_sync = [&]() {
    if constexpr (GLDS) SCAMD_BARRIER_LDS();
    else __syncthreads();
  };

  if constexpr (!IVF) {
    sweep(0, n_tiles_all, true);
    // the lists leave in SORTED order (entry l31 = the l31-th smallest key; its row id sits in the slot named by the key's
    // low bits): pass 2 re-ranks the entries below the threshold only (knn_rerank_rows_kernel)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t q = (int64_t)blk * C::QB + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      const int sid = __shfl(idx[r], 32 * half + (__float_as_int(key[r]) & KEY_SLOT_MASK));
      cand_idx[q * C::KP + l31] = sid;
      if (l31 == thr_lane) cand_tau[q] = key[r];
    }
  } else {
    // ---- cell order of this block: row a of the per-cell tables (ascending lower bound, own cell first), built once
    // per cell by ivf_cell_order_kernel instead of once per block ----
    float* wmax = smem + C::NBUF * TC * DPL;  // [4]
    const unsigned long long trace_t0 = iv.trace ? wall_clock64() : 0ull;
    unsigned long long trace_tiles = 0;
    const int a = iv.block_cell[blk];
    const int* order = iv.order + (int64_t)a * iv.n_cells;
    const float* lb2 = iv.order_lb2 + (int64_t)a * iv.n_cells;
    // the head of the order with its tile ranges -> LDS (one gather by wave 0, under the latency of the query operand's)
    int* s_t0 = reinterpret_cast<int*>(wmax + 16);
    int* s_nt = s_t0 + IVF_META_CELLS;
    float* s_lb = reinterpret_cast<float*>(s_nt + IVF_META_CELLS);
    // (entries c0 .. c0 + 63; the rare block that sweeps more than 64 cells refills the table -- the loop itself holds no
    // pointer into the tables: they cost 20 scalar registers, spilled, when the loop could fall back to them)
    auto meta_fill = [&](int c0) {
      if (tid < IVF_META_CELLS) {
        const int ci = c0 + tid;
        const bool in = ci < iv.n_cells;
        const int cb = in ? iv.order[(int64_t)a * iv.n_cells + ci] : 0;
        s_lb[tid] = in ? iv.order_lb2[(int64_t)a * iv.n_cells + ci] : INFINITY;
        s_t0[tid] = iv.cell_tile0[cb];
        s_nt[tid] = iv.cell_ntiles[cb];
      }
      block_sync();
    };
    meta_fill(0);
    auto meta_lb = [&](int ci) -> float { return s_lb[ci & (IVF_META_CELLS - 1)]; };
    auto meta_t0 = [&](int ci) -> int { return s_t0[ci & (IVF_META_CELLS - 1)]; };
    auto meta_nt = [&](int ci) -> int { return s_nt[ci & (IVF_META_CELLS - 1)]; };
    // (the trace's intermediate stamps go straight to memory: no registers held across the sweeps for a debug mode)
    if (iv.trace && tid == 0) iv.trace[(size_t)blk * 8 + 4] = wall_clock64();
    // exact threshold distance^2 of a query = thr (score space) + ||q||^2; per wave the max over its real queries
    // ||q||^2 sits in the extra k slot of the row's second half (B3: in the row's tail)
    const float qn = B3 ? xp[qrow * DPL + 64] : xp[qrow * DPL + HP + H];
    // ---- pre-pass: a tight starting threshold from the own cell ----
    // A streaming top-k list with threshold "current thr_rank-th best" inserts ~thr_rank*(1 + ln(N/thr_rank))
    // candidates per query, most of them while the list warms up; each insertion costs ~30 VALU on the lanes the
    // MFMA chain needs.  The pre-pass sweeps the own cell once with threshold 0 (acc = plain score) and only
    // keeps lane-wise minima (16 v_min per sub-tile, no branches): 32 distinct candidates per query, whose
    // thr_rank-th smallest bounds the final threshold from above (~ the 34th nearest of the cell for rank 21).
    // The real sweep then starts with that threshold: about half as many insertions in total.
    {
      minima = true;
      // round 4: the pre-pass keeps the TWO smallest scores per lane (64 distinct candidates per query instead of 32; the
      // second one in idx[r], unused until the lists start): the thr_rank-th smallest of 64 is about the 27th best
      // candidate of the cell, that of 32 lane minima about the 49th -- and every candidate below the starting threshold
      // costs an insertion when the real sweep meets it
#pragma unroll
      for (int r = 0; r < 16; ++r) idx[r] = __float_as_int(KEY_BIG);
      const int pre_tiles = min(iv.cell_ntiles[a], iv.prepass_tiles);
      if (tid == 0) atomicAdd(iv.pairs + 1, (unsigned long long)pre_tiles * TC * C::QB);  // counted apart: not useful work
      // (the real sweep starts with the own cell's first tile again: requested under the pre-pass's last tile)
      next_tile0 = (iv.cell_preload && iv.prepass_cells <= 1) ? meta_t0(0) : -1;
      next_n = meta_nt(0);
      sweep(iv.cell_tile0[a], pre_tiles, false);
      next_tile0 = -1;
      // round 4: the pre-pass may go on over the next nearest cells (iv.prepass_cells - 1 of them, whole cells).  The sweep
      // is bound by the instructions of the list insertions, not by the matrix pipe (counters: profiles/r04a_knn_pmc*.csv),
      // and an insertion-free pass over more candidates starts the lists nearer their final thresholds.
      for (int ci = 1; ci < iv.prepass_cells && ci < iv.n_cells; ++ci) {
        if (!(lb2[ci] < INFINITY)) break;
        const int pb = order[ci];
        block_sync();
        if (tid == 0) atomicAdd(iv.pairs + 1, (unsigned long long)iv.cell_ntiles[pb] * TC * C::QB);
        sweep(iv.cell_tile0[pb], iv.cell_ntiles[pb], false);
      }
      minima = false;
      block_sync();
      // compare-exchange constants: a lane whose bit b (of l31) is clear keeps the SMALLER value of its pair --
      // v = med3(v, partner, lim[b]) with lim = -inf there and +inf on the partner's side (negated: the larger value)
      float lim[5];
#pragma unroll
      for (int b = 0; b < 5; ++b) lim[b] = (l31 & (1 << b)) ? INFINITY : -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        // bitonic sort of the 64 values of each half's query: element e = 32 j + l31, j = 0 in x (smallest per lane), j = 1
        // in y (second smallest).  Stages 2 .. 32 sort both registers along the 32 lanes, x ascending and y descending, in
        // the "mirror" form of the network (the first step of a merge of blocks of B pairs lane i with lane B - 1 - i, the
        // rest with i ^ j: every block ascends, no direction flags); then the 64-merge: its first step compares x with y in
        // the lane and leaves the 32 smallest of the 64 in x, which its remaining five steps put in ascending order.
        // Round 5: the partners come through the VALU (DPP quad_perm / row_half_mirror / row_mirror / row_ror:8, two
        // bank-masked row shifts for i ^ 4) except the three that cross a row of 16 (ds_swizzle): 3 LDS-pipe exchanges per
        // register instead of 35 `ds_bpermute`, each of which the compiler had put a full `s_waitcnt lgkmcnt(0)` behind.
        float x = key[r], y = iv.prepass_min2 ? __int_as_float(idx[r]) : KEY_BIG;  // (SCAMD_KNN_PREPASS_MIN2=0: lane minima only, rounds 1-3)
        idx[r] = -1;
        auto cx2 = [&](float xp_, float yp_, int b) __attribute__((always_inline)) {
          x = __builtin_amdgcn_fmed3f(x, xp_, lim[b]);
          y = __builtin_amdgcn_fmed3f(y, yp_, -lim[b]);
        };
        cx2(dpp_f32<0xB1>(x), dpp_f32<0xB1>(y), 0);                    // blocks of 2
        cx2(dpp_f32<0x1B>(x), dpp_f32<0x1B>(y), 1);                    // blocks of 4: mirror, then i ^ 1
        cx2(dpp_f32<0xB1>(x), dpp_f32<0xB1>(y), 0);
        cx2(dpp_f32<0x141>(x), dpp_f32<0x141>(y), 2);                  // blocks of 8: row_half_mirror, i ^ 2, i ^ 1
        cx2(dpp_f32<0x4E>(x), dpp_f32<0x4E>(y), 1);
        cx2(dpp_f32<0xB1>(x), dpp_f32<0xB1>(y), 0);
        cx2(dpp_f32<0x140>(x), dpp_f32<0x140>(y), 3);                  // blocks of 16: row_mirror, i ^ 4, i ^ 2, i ^ 1
        cx2(dpp_xor4_f32(x), dpp_xor4_f32(y), 2);
        cx2(dpp_f32<0x4E>(x), dpp_f32<0x4E>(y), 1);
        cx2(dpp_f32<0xB1>(x), dpp_f32<0xB1>(y), 0);
        cx2(swizzle_f32<0x7C1F>(x), swizzle_f32<0x7C1F>(y), 4);        // blocks of 32: i ^ 31 (swizzle), row_ror:8, i ^ 4, ...
        cx2(dpp_f32<0x128>(x), dpp_f32<0x128>(y), 3);
        cx2(dpp_xor4_f32(x), dpp_xor4_f32(y), 2);
        cx2(dpp_f32<0x4E>(x), dpp_f32<0x4E>(y), 1);
        cx2(dpp_f32<0xB1>(x), dpp_f32<0xB1>(y), 0);
        x = __builtin_amdgcn_fmed3f(x, y, -INFINITY);                  // min: the 32 smallest, a bitonic sequence along the lanes
        x = __builtin_amdgcn_fmed3f(x, swizzle_f32<0x401F>(x), lim[4]);
        x = __builtin_amdgcn_fmed3f(x, dpp_f32<0x128>(x), lim[3]);
        x = __builtin_amdgcn_fmed3f(x, dpp_xor4_f32(x), lim[2]);
        x = __builtin_amdgcn_fmed3f(x, dpp_f32<0x4E>(x), lim[1]);
        x = __builtin_amdgcn_fmed3f(x, dpp_f32<0xB1>(x), lim[0]);
        const int i0 = (r & 3) + 8 * (r >> 2), i1 = i0 + 4;
        const float t0 = readlane_f32(x, thr_lane), t1 = readlane_f32(x, 32 + thr_lane);
        athr = (lane == i0) ? -t0 : ((lane == i1) ? -t1 : athr);
        // the list starts out as 32 placeholders AT the pre-pass threshold (row id -1): the list's thr_rank-th entry
        // can then never exceed that proven bound, and real entries (all below it) displace the placeholders from
        // the top.  Slot numbers ascend with the key: l31 for a positive threshold, 31 - l31 for a negative one.
        const int tb = __float_as_int(fminf(half ? t1 : t0, KEY_BIG));  // (never +inf: its mantissa holds the slot)
        key[r] = __int_as_float((tb & ~KEY_SLOT_MASK) | (tb < 0 ? 31 - l31 : l31));
      }
      // Padding slots (the last block of a cell is half empty on average: 3 % of all slots) are "queries" at image row 0, a
      // row of some far cell: the cells come in the order of THIS cell's bounds, which for that row is no order at all, so
      // its list kept improving through the whole sweep -- the last block of a cell was the slowest of the cell in 390 of 484
      // cells, by a factor 1.5 (profiles/r05r_knn_last_block_of_a_cell.log), and those blocks end their queues.  Their
      // threshold becomes -1e30: no score is below it, nothing is inserted, nothing of theirs is read.
      if (half == 0 && !qvalid) athr = 1e30f;
      if constexpr (COARSE) {
        float qmax2 = qvalid ? qn : 0.f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) qmax2 = fmaxf(qmax2, __shfl_xor(qmax2, o));
        slack = 0.015625f * 1.02f * sqrtf(qmax2 * __uint_as_float(iv.cmax_bits[0]));
      }
      sync_thr();
    }
    if (iv.trace && tid == 0) iv.trace[(size_t)blk * 8 + 5] = wall_clock64();
    bool first = true;
    for (int ci = 0; ci < iv.n_cells; ++ci) {
      if (ci > 0 && (ci & (IVF_META_CELLS - 1)) == 0) meta_fill(ci);  // (the sweep's closing barrier: nobody reads the old entries)
      const float lb = meta_lb(ci);
      if (!(lb < INFINITY)) break;  // empty cells sort last
      if (ci > 0) {
        // athr = -thr on lanes 0..31.  thr lives in score space (||c||^2 - 2 q.c), where float32 carries an absolute
        // error of ~(d + 14) 2^-24 ||q||^2 (the rounding of ||q||^2 itself, of the sum below and of the scores the
        // threshold was taken from); far from the origin (||q||^2 >> d^2) that exceeds the 1e-3 relative slack of
        // the test below, so it is added per query: 1e-5 >= 142 * 2^-24 covers d <= 128
        float dthr = (half == 0 && qvalid) ? (qn - athr) + 1e-5f * qn : -INFINITY;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dthr = fmaxf(dthr, __shfl_xor(dthr, o));
        if (lane == 0) wmax[wave] = dthr;
        block_sync();
        const float tmax = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        block_sync();  // wmax is rewritten at the next cell
        // every remaining cell is at least this far: done once the bound clears every threshold (with slack for
        // the float32 rounding of thresholds and bounds)
        if (lb * (1.0f - 1e-3f) > tmax + 1e-3f * fabsf(tmax)) break;
      }
      const int c_t0 = meta_t0(ci), c_nt = meta_nt(ci);
      trace_tiles += c_nt;  // (also the block's count of evaluated pairs: one atomic when it is done)
      // (no request across a refill of the table, nor past the last cell: the table's entries beyond n_cells are INF)
      next_tile0 = (iv.cell_preload && ((ci + 1) & (IVF_META_CELLS - 1)) != 0 && meta_lb(ci + 1) < INFINITY) ? meta_t0(ci + 1) : -1;
      next_n = meta_nt((ci + 1) & (IVF_META_CELLS - 1));
      const unsigned long long ts = iv.trace ? wall_clock64() : 0ull;
      sweep(c_t0, c_nt, first);
      first = false;
      block_sync();  // all fragment reads of this cell are done before the next sweep restages the tiles
      if (iv.trace && tid == 0) {
        iv.trace[(size_t)blk * 8 + 6] += wall_clock64() - ts;
        iv.trace[(size_t)blk * 8 + 7] += 1ull;
      }
    }
    if constexpr (GLDS) SCAMD_WAIT_VM0();  // (requests for a cell the stopping rule skipped)
    if (tid == 0) atomicAdd(iv.pairs, trace_tiles * TC * C::QB);
    if (iv.trace && tid == 0) {
      unsigned int hw;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      unsigned int xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      unsigned long long* tr = iv.trace + (size_t)blk * 8;
      tr[0] = trace_t0;
      tr[1] = wall_clock64();
      tr[2] = trace_tiles;
      tr[3] = ((unsigned long long)xcc << 32) | hw;
    }
    // results go to the ORIGINAL query / row ids
    // (sorted order: see the brute-force branch; the shuffle sits outside the branch on qp -- every lane takes part)
    const int qbase = (int64_t)blk * C::QB + wave * 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qslot = qbase + (r & 3) + 8 * (r >> 2) + 4 * half;
      const int qp = iv.qpos[qslot];
      const int sid = __shfl(idx[r], 32 * half + (__float_as_int(key[r]) & KEY_SLOT_MASK));
      // final threshold = min(list entry, pre-pass threshold): everything below it is in the list
      // (cross-lane reads before the branch on qp: the two halves of the wave may part there)
      const int i0 = (r & 3) + 8 * (r >> 2), i1 = i0 + 4;
      const float tf0 = -readlane_f32(athr, i0), tf1 = -readlane_f32(athr, i1);
      // pass 2 walks the queries in slot order (= cell order: neighbouring queries share their candidates' rows)
      if (iv.qorder && l31 == 0) iv.qorder[qslot] = qp >= 0 ? iv.perm[qp] - (int)q_begin : -1;
      if (qp >= 0) {
        const int64_t qi = (int64_t)iv.perm[qp] - q_begin;
        cand_idx[qi * C::KP + l31] = sid >= 0 ? iv.perm[sid] : -1;
        if (l31 == thr_lane) cand_tau[qi] = half ? tf1 : tf0;
      }
    }
  }
}

// The launch.  Brute force: workgroup b = block b.  Pruned sweep: launch slot s -> block block_perm[s] (longest expected
// sweep first, the blocks of a cell on one XCD: slots x, x + 8, ... are the queue of XCD x, ivf_block_order_kernel), either
// one workgroup per slot, or -- round 5, iv.queue_ctr -- PERSISTENT: as many workgroups as the chip holds, each taking the
// next block off the queue of its XCD (workgroup b runs on XCD b mod 8, like slot b would) and, once that queue is empty,
// off the other XCDs' queues.  The per-block timeline (profiles/r05p_knn_timeline.log) had the queues finish 0.9 ms apart
// (12.30 .. 13.18 ms: the work estimates behind the static deal are a ranking, not a measurement) and 4 % of the block
// slots empty in the steady state, waiting for the dispatcher.
template <int H, int TC_, int WPS, bool IVF, bool B3 = false, bool COARSE = false>
__global__ __launch_bounds__(256, WPS) void knn_select_reg_kernel(const float* __restrict__ xp, int n_tiles_all,
                                                                  int64_t n_pad, int64_t q_begin,
                                                                  int thr_rank, int* __restrict__ cand_idx,
                                                                  float* __restrict__ cand_tau, IvfArgs iv) {
  if constexpr (!IVF) {
    knn_select_reg_block<H, TC_, WPS, IVF, B3, COARSE>((int)blockIdx.x, xp, n_tiles_all, n_pad, q_begin, thr_rank, cand_idx, cand_tau, iv);
  } else {
    using C = RegCfg<H, TC_, B3>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int* s_next = reinterpret_cast<int*>(smem + C::NBUF * C::TC * C::DPL + 8);  // (wmax[4 .. 15] are unused)
    for (int pass = 0;; ++pass) {
      int blk = -1;
      if (iv.queue_ctr == nullptr) {
        if (pass == 0) blk = iv.block_perm[blockIdx.x];  // (-1: launch slot without a block)
      } else {
        __syncthreads();  // nobody reads the previous block's tiles / tables / s_next any more
        if (threadIdx.x == 0) {
          int got = -1;
          const int x0 = (int)(blockIdx.x & 7);
          for (int t = 0; t < 8 && got < 0; ++t) {
            const int x = (x0 + t) & 7;
            // (a queue's blocks sit at its positions 0 .. len - 1, -1 behind them: a queue that ran dry is not taken from again)
            const int64_t peek = 8 * (int64_t)__hip_atomic_load(&iv.queue_ctr[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + x;
            if (peek >= iv.n_slots || iv.block_perm[peek] < 0) continue;
            const int64_t slot = 8 * (int64_t)atomicAdd(&iv.queue_ctr[x], 1) + x;
            if (slot < iv.n_slots) got = iv.block_perm[slot];  // (-1: somebody else took the queue's last block in between)
          }
          *s_next = got;
        }
        __syncthreads();
        blk = *s_next;
      }
      if (blk < 0) return;
      knn_select_reg_block<H, TC_, WPS, IVF, B3, COARSE>(blk, xp, n_tiles_all, n_pad, q_begin, thr_rank, cand_idx, cand_tau, iv);
    }
  }
}

// Per-cell sweep order of the cell-pruned search: block a sorts all cells by the lower bound
// LB(a,b) = max(0, |c_a - c_b| (1 - 1e-4) - r_a - r_b) on the distance between any member of a and any member of b
// (own cell first, empty cells last with LB = INF) and writes row a of order / order_lb2 (LB squared).
// nprobe > 0 -- the APPROXIMATE mode (scamd_knn_l2_ivf_f32): the cells are sorted by the distance between the centroids
// instead (the ball bound says nothing when the balls overlap, which on data without separated clusters is every pair
// of cells), the first nprobe non-empty cells get the bound 0 (always swept) and every later one INF -- the sweep
// kernel stops at the first INF, so it needs no change: a query block sees the rows of its cell's nprobe nearest cells.
__global__ __launch_bounds__(256) void ivf_cell_order_kernel(const float* __restrict__ centers, int d,
                                                             const float* __restrict__ radius,
                                                             const int* __restrict__ cell_ntiles, int n_cells,
                                                             int* __restrict__ order_out, float* __restrict__ lb2_out,
                                                             int* __restrict__ work_out, int nprobe) {
  extern __shared__ __attribute__((aligned(16))) float co_smem[];
  int npow = 1;
  while (npow < n_cells) npow <<= 1;
  float* lb2 = co_smem;                                 // [npow]
  int* order = reinterpret_cast<int*>(co_smem + npow);  // [npow]
  const int a = blockIdx.x, tid = threadIdx.x;
  const float ra = radius[a];
  for (int b = tid; b < npow; b += 256) {
    float v = INFINITY;
    if (b < n_cells && cell_ntiles[b] > 0) {
      float d2 = 0.f;
      for (int c = 0; c < d; ++c) {
        const float df = centers[a * d + c] - centers[b * d + c];
        d2 += df * df;
      }
      const float lb = fmaxf(0.f, sqrtf(d2) * (1.0f - 1e-4f) - ra - radius[b]);
      v = (b == a) ? -1.0f : (nprobe > 0 ? d2 : lb * lb);
    }
    lb2[b] = v;
    order[b] = b;
  }
  __syncthreads();
  for (int kk = 2; kk <= npow; kk <<= 1) {  // bitonic sort by (lb2, cell)
    for (int j = kk >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < npow; i += 256) {
        const int p = i ^ j;
        if (p > i) {
          const bool up = (i & kk) == 0;
          const float vi = lb2[i], vp = lb2[p];
          const int oi = order[i], op = order[p];
          const bool gt = vi > vp || (vi == vp && oi > op);
          if (gt == up) {
            lb2[i] = vp;
            lb2[p] = vi;
            order[i] = op;
            order[p] = oi;
          }
        }
      }
      __syncthreads();
    }
  }
  if (nprobe > 0) {  // (every thread rewrites its own entries; the own cell keeps -1)
    for (int i = tid; i < npow; i += 256)
      if (i > 0) lb2[i] = (i < nprobe && lb2[i] < INFINITY) ? 0.f : INFINITY;
    __syncthreads();
  }
  for (int i = tid; i < n_cells; i += 256) {
    order_out[(int64_t)a * n_cells + i] = order[i];
    lb2_out[(int64_t)a * n_cells + i] = lb2[i];
  }
  // expected sweep length of a block of this cell (a ranking, not a bound: only the launch order uses it): the tiles
  // of the cells whose lower bound is within the cell's own radius -- in a clustered embedding, the cells of the same
  // cluster.  Neighbour distances are of the order of the radius of a ~2000-row cell in 50 dimensions.
  // (approximate mode: the tiles of the nprobe cells, exactly)
  __shared__ int s_work;
  if (tid == 0) s_work = 0;
  __syncthreads();
  int wsum = 0;
  for (int i = tid; i < n_cells; i += 256)
    if (lb2[i] <= ra * ra) wsum += cell_ntiles[order[i]];
  atomicAdd(&s_work, wsum);
  __syncthreads();
  if (tid == 0) work_out[a] = cell_ntiles[a] > 0 ? s_work : 0;
}

// Launch order of the pruned sweep: cells by decreasing expected work (ties: cell id), every cell's blocks together.
// One workgroup; n_cells <= IVF_MAX_CELLS = 1024.  Blocks differ 10x in duration (cluster sizes); in block-id order the
// last ~12 % of a launch ran with most CUs idle (SCAMD_KNN_TRACE timeline, round 2).
__global__ __launch_bounds__(1024) void ivf_block_order_kernel(const int* __restrict__ work, const int* __restrict__ blk_off,
                                                               const int* __restrict__ blk_cnt, int n_cells,
                                                               int* __restrict__ block_perm, int n_slots, int xcd_mode,
                                                               int* __restrict__ err) {
  __shared__ long long key[IVF_MAX_CELLS];
  __shared__ int start[IVF_MAX_CELLS];
  const int tid = threadIdx.x;
  start[tid] = -1;  // (a cell no queue had room for keeps -1: reported through *err, never a silent skip)
  // descending work, ascending cell: key = (~work << 32) | cell, sorted ascending
  key[tid] = tid < n_cells ? (((long long)(0x7fffffff - work[tid])) << 32) | (unsigned int)tid : 0x7fffffffffffffffll;
  for (int t = tid; t < n_slots; t += 1024) block_perm[t] = -1;  // launch slots without a block (xcd_mode) exit at once
  __syncthreads();
  for (int kk = 2; kk <= IVF_MAX_CELLS; kk <<= 1) {
    for (int j = kk >> 1; j > 0; j >>= 1) {
      const int p = tid ^ j;
      if (p > tid) {
        const bool up = (tid & kk) == 0;
        const long long a = key[tid], b = key[p];
        if ((a > b) == up) {
          key[tid] = b;
          key[p] = a;
        }
      }
      __syncthreads();
    }
  }
  // xcd_mode: workgroup s of a launch runs on XCD s mod 8 (eight L2 caches that do not share).  The blocks of ONE cell
  // sweep the same cells in the same order, so they go to ONE XCD -- launch slots x, x + 8, x + 16, ... -- and pull each
  // tile through that L2 once instead of through all eight (round 4 counters of the block-id order: 35 GB fetched from
  // the memory side for 59 GB staged into LDS per launch).  The cells, longest expected sweep first, are dealt round
  // robin: cell i of the sorted order joins the queue of XCD i mod 8 -- every queue is still longest-first and holds
  // every eighth cell of the order (thread x builds queue x).  Otherwise: exclusive scan of the block counts in sorted
  // order (serial: 1024 entries)
  // (block counts and work estimates in sorted order, staged in LDS: the serial loops below then wait for LDS, not for a
  // dependent global load per cell -- 175 -> ~50 us for 512 cells)
  __shared__ int s_cnt[IVF_MAX_CELLS];
  __shared__ int s_work[IVF_MAX_CELLS];
  if (tid < n_cells) {
    const int c = (int)(key[tid] & 0xffffffffll);
    s_cnt[tid] = blk_cnt[c];
    s_work[tid] = max(work[c], 1);
  }
  __syncthreads();
  if (xcd_mode) {
    // cell i of the sorted order goes to the XCD whose queue holds the least work so far (blocks x expected sweep per
    // block; dealing the cells round robin left the queues up to 10 % apart: 18.99 vs 17.42 ms, profiles/r04d).  The
    // eight running totals live in lanes 0..7 of the first wave, the minimum is found with three shuffles.
    if (tid < 64) {
      long long load = 0ll;
      int len = 0;
      for (int i = 0; i < n_cells; ++i) {
        const int cnt = s_cnt[i];
        const long long wk = (long long)cnt * (long long)s_work[i];
        // (a queue may not outgrow its share of the launch slots; the slots are sized so that some queue always has room)
        const bool room = tid < 8 && (len + cnt) * 8 <= n_slots;
        long long best = room ? (load << 3) | (long long)(tid & 7) : 0x7fffffffffffffffll;  // (totals stay far below 2^60)
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) {
          const long long other = __shfl_xor(best, o);
          best = other < best ? other : best;
        }
        const int xb = (int)(best & 7);
        if (tid == xb && best != 0x7fffffffffffffffll) {
          start[i] = len * 8 + xb;
          len += cnt;
          load += wk;
        }
      }
    }
  } else if (tid == 0) {
    int run = 0;
    for (int i = 0; i < n_cells; ++i) {
      start[i] = run;
      run += s_cnt[i];
    }
  }
  __syncthreads();
  if (tid < n_cells) {
    const int c = (int)(key[tid] & 0xffffffffll);
    const int o = blk_off[c], cnt = blk_cnt[c], st = start[tid];
    // a block without a launch slot would leave its queries unanswered: the host sizes the slots so that this cannot
    // happen (run_ivf_select) and turns the flag into an error if it ever does
    if (cnt > 0 && st < 0) atomicOr(err, 1);
    for (int t = 0; t < cnt && st >= 0; ++t) {
      const int slot = xcd_mode ? st + 8 * t : st + t;
      if (slot < n_slots) block_perm[slot] = o + t;
      else atomicOr(err, 1);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// pass 2: exact float64 re-rank + certificate.  One wave per query, 4 queries per block.
// ------------------------------------------------------------------------------------------------
__device__ inline bool key_less(double da, int ia, double db, int ib) {
  return da < db || (da == db && ia < ib);
}

template <int KP>
__global__ __launch_bounds__(256) void knn_rerank_kernel(
    const float* __restrict__ x, const float* __restrict__ mu, int64_t n, int d, int64_t ld, int64_t q_begin,
    int64_t n_query, int k, const int* __restrict__ cand_idx, const float* __restrict__ cand_tau,
    const unsigned int* __restrict__ cmax_bits, double cert_scale, double cert_k, double cert_k2,
    int32_t* __restrict__ out_idx,
    double* __restrict__ out_dist, double* __restrict__ kth_d2, int* __restrict__ flag_list,
    int* __restrict__ n_flag, const int* __restrict__ qlist, int n_list) {
  constexpr int PER = (KP + 63) / 64;
  __shared__ float qs[4][KNN_MAX_D];
  __shared__ double sd[4][KP];
  __shared__ int si[4][KP];
  __shared__ double skth[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // (qlist: only these queries -- the second tier of the bf16 engine re-ranks what its first certificate rejected)
  const int64_t qj = (int64_t)blockIdx.x * 4 + w;
  if (qj >= (qlist ? (int64_t)n_list : n_query)) return;  // whole wave exits together (no block-level sync below)
  const int64_t qi = qlist ? (int64_t)qlist[qj] : qj;
  const int64_t q = q_begin + qi;
  for (int c = lane; c < d; c += 64) qs[w][c] = x[q * ld + c];
  if (lane == 0) skth[w] = INFINITY;
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0);

  double qn = 0.0;  // ||fl(q - mu)||^2: the norm of the query's image row (the thresholds live in that frame)
  for (int c = 0; c < d; ++c) {
    const double qc = (double)__fsub_rn(qs[w][c], mu[c]);
    qn += qc * qc;
  }

  double myd[PER];
  int myi[PER];
#pragma unroll
  for (int p = 0; p < PER; ++p) {
    int u = lane + p * 64;
    myd[p] = INFINITY;
    myi[p] = -1;
    if (u < KP) {
      int idx = cand_idx[qi * KP + u];
      if (idx >= 0 && idx < n && idx != q) {
        const float* cp = x + (int64_t)idx * ld;
        double s = 0.0;
        // RB coordinates of the candidate's row requested at a time, summed in the same order as one by one (every lane
        // walks its own row: one load in flight per lane made this kernel 50 serial cache round trips per query; eight at
        // a time plus a one-by-one tail were still 8 trips for d = 50 in a kernel of a wave per query -- a million waves,
        // latency x occupancy bound).  The tail is a partial batch: clamped addresses, products of the pad not added.
        constexpr int RB = 12;
        // two batches in flight: the loads of batch b + 1 are issued before batch b is consumed
        float ta[RB], tb[RB];
#pragma unroll
        for (int u = 0; u < RB; ++u) ta[u] = cp[min(u, d - 1)];
        for (int c = 0; c < d; c += 2 * RB) {
#pragma unroll
          for (int u = 0; u < RB; ++u) tb[u] = cp[min(c + RB + u, d - 1)];
#pragma unroll
          for (int u = 0; u < RB; ++u) {
            // (no branch: a pad coordinate contributes fma(0, 0, s) = s; with `if (c + u < d)` the compiler loaded one
            // element of the batch inside the branch and waited for ALL of them there)
            const double df = c + u < d ? (double)qs[w][min(c + u, d - 1)] - (double)ta[u] : 0.0;
            s = fma(df, df, s);
          }
#pragma unroll
          for (int u = 0; u < RB; ++u) ta[u] = cp[min(c + 2 * RB + u, d - 1)];
#pragma unroll
          for (int u = 0; u < RB; ++u) {
            const double df = c + RB + u < d ? (double)qs[w][min(c + RB + u, d - 1)] - (double)tb[u] : 0.0;
            s = fma(df, df, s);
          }
        }
        myd[p] = s;
        myi[p] = idx;
      }
      // NaN distances sort last and are reported as missing
      if (!(myd[p] == myd[p])) {
        myd[p] = INFINITY;
        myi[p] = -1;
      }
      sd[w][u] = myd[p];
      si[w][u] = myi[p];
    }
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0);

  const int kk = k - 1;  // neighbours besides self
  if (lane == 0) {
    out_idx[qi * k] = (int32_t)q;
    out_dist[qi * k] = 0.0;
  }
#pragma unroll
  for (int p = 0; p < PER; ++p) {
    int u = lane + p * 64;
    if (u < KP) {
      int rank = 0;
      for (int v = 0; v < KP; ++v) {
        double dv = sd[w][v];
        int iv = si[w][v];
        // missing entries (idx -1, +inf) order after everything, ties among them by slot
        bool less = (iv >= 0) ? (myi[p] < 0 || key_less(dv, iv, myd[p], myi[p]))
                              : (myi[p] < 0 && v < u);
        rank += less ? 1 : 0;
      }
      if (rank < kk) {
        out_idx[qi * k + 1 + rank] = myi[p];
        out_dist[qi * k + 1 + rank] = (myi[p] >= 0) ? sqrt(myd[p]) : INFINITY;
        if (rank == kk - 1) skth[w] = myd[p];
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0);
  if (lane == 0) {
    double dk = (kk > 0) ? skth[w] : 0.0;
    double tau = (double)cand_tau[qi];
    double cmax = (double)__uint_as_float(*cmax_bits);
    // |s_float32 - s_exact| <= (2H+2) u (||c||^2 + 2 ||q|| ||c||), u = 2^-24; 2H+2 <= 130.
    // Centring: the image holds fl(x - mu), each coordinate off by <= u |x - mu|, so a squared distance between image
    // rows differs from the true one by <= 4 u (qn + cmax) <= 8 u (cmax + 2 ||q|| ||c||) -> 138.
    // Slot keys (register-list kernel): the threshold and the keys it was compared with carry a slot number in their
    // 5 low mantissa bits, i.e. each is off by < 32 ulp OF ITS OWN MAGNITUDE -- the threshold's, not cmax's (keys far
    // above the threshold cannot be confused with it): 128 u |tau| with a factor 2 for a binade boundary.  (Charging it
    // against cmax as well, factor 202, sent 4317 instead of 379 queries of the 10M x 50 run to the float64 scan: +3 s.)
    // 3 x bf16 engine: cert_k = 138 + 146 (198 accumulated terms instead of 2H + 2 = 52) and cert_k2 = 768 on the 2 q.c
    // term alone (the hi + lo split's dropped pieces, 3 * 2^-16: the norm and the threshold are split EXACTLY).
    // Which candidates does the bound have to hold for?  Only for rows that could be a missed neighbour, i.e. rows within
    // sqrt(dk) of the query: their norm is at most ||q|| + sqrt(dk) (triangle inequality; 1e-3 relative slack for the
    // centring perturbation of the image rows).  The global maximum norm is only the cap: one far outlier row no longer
    // inflates the bound of every query (round 4; before, a single row with a huge coordinate sent ALL queries to the
    // float64 scan).
    {
      const double rq = sqrt(qn) + sqrt(dk);
      const double cq = rq * rq * (1.0 + 1e-3);
      if (cq < cmax) cmax = cq;
    }
    double eps = cert_scale * 5.9604644775390625e-08 *
                 (cert_k * (cmax + 2.0 * sqrt(qn * cmax)) + cert_k2 * 2.0 * sqrt(qn * cmax) + 128.0 * fabs(tau));
    bool certified = (tau >= 1e38f) || ((dk - qn) + eps < tau);
    kth_d2[qi] = dk;
    if (!certified) {
      int slot = atomicAdd(n_flag, 1);
      flag_list[slot] = (int)qi;
    }
  }
}

// Pass 2 for the register-list kernels (lists of 32 in sorted order, d <= 64): two queries per wave, and the candidates'
// rows travel ROW-WISE.  The kernel above lets every lane walk its own candidate's row: each of its load instructions
// touches 32 different cache lines, 50 instructions per query -- 1600 line look-ups per query in the vector L1, which
// serves about one per cycle: 1M queries = 6.2M cycles per CU = the measured 2.0 ms (and 3.5 ms for the variant that
// issued 72 clamped loads instead of 50: profiles/r04a_bench_kernel_stats.csv).  Here one load instruction fetches one
// candidate's whole row (lanes = coordinates: 2-3 lines), straight into LDS (global_load_lds_dword, no staging
// registers, all rows of both queries in flight at once); then lane j reads row j back from LDS (odd row stride: no bank
// conflict) and sums (q - c)^2 in float64 in coordinate order -- the same operations in the same order as above, so the
// distances are bit for bit the same.  Only the entries BELOW the threshold entry are candidates (n_rank = thr_rank - 1:
// a certified query has its k - 1 neighbours there; an uncertified one is redone by the float64 scan whatever this
// kernel reports), and the queries come in the select kernel's slot order (qlist = IvfArgs::qorder: queries of one
// cell after another, whose candidate rows are the same few thousand rows of x).
// LDS per wave: [2 n_rank + 2][RS] floats (RS = d | 1), then n_rank-entry (double, int) tables per half.
__global__ __launch_bounds__(256) void knn_rerank_rows_kernel(
    const float* __restrict__ x, const float* __restrict__ mu, int64_t n, int d, int64_t ld, int64_t q_begin,
    int64_t n_query, int k, int n_rank, const int* __restrict__ cand_idx, const float* __restrict__ cand_tau,
    const unsigned int* __restrict__ cmax_bits, double cert_scale, double cert_k, double cert_k2,
    int32_t* __restrict__ out_idx, double* __restrict__ out_dist, double* __restrict__ kth_d2,
    int* __restrict__ flag_list, int* __restrict__ n_flag, const int* __restrict__ qlist, int64_t n_list) {
  extern __shared__ __attribute__((aligned(16))) float rsm[];
  const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int RS = d | 1;
  const int wave_floats = (2 * n_rank + 2) * RS + 2 + 2 * n_rank * 3 + 4;  // rows, query rows, (pad,) sd (double) + si, skth (double)
  float* rows = rsm + (size_t)w * ((wave_floats + 3) & ~3);
  float* qrows = rows + 2 * n_rank * RS;
  double* sd = reinterpret_cast<double*>(rows + (((2 * n_rank + 2) * RS + 1) & ~1));  // [2][n_rank]
  double* skth = sd + 2 * n_rank;                                                    // [2]
  int* si = reinterpret_cast<int*>(skth + 2);                                          // [2][n_rank]
  const int64_t nq_list = qlist ? n_list : n_query;
  const int64_t pair = (int64_t)blockIdx.x * 4 + w;
  if (2 * pair >= nq_list) return;  // whole wave (no block-level sync below)
  const int64_t qj = 2 * pair + half;
  int qi = -1;
  if (qj < nq_list) qi = qlist ? qlist[qj] : (int)qj;
  const bool qv = qi >= 0;
  const int q = (int)q_begin + (qv ? qi : 0);
  int ci = -1;
  if (qv && l31 < n_rank) {
    ci = cand_idx[(int64_t)qi * 32 + l31];
    if (ci < 0 || ci >= n || ci == q) ci = -1;
  }
  // the two query rows, then every candidate row: one instruction per row, lanes = coordinates
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int qh = __builtin_amdgcn_readlane(q, 32 * h);
    const int vh = __builtin_amdgcn_readlane(qv ? 1 : 0, 32 * h);
    if (vh && lane < d)  // (d <= 64: one instruction per row)
      __builtin_amdgcn_global_load_lds(x + (int64_t)qh * ld + lane, SCAMD_LDS_PTR(qrows + h * RS), 4, 0, 0);
  }
  for (int j = 0; j < 2 * n_rank; ++j) {
    const int sl = j >= n_rank ? 32 + (j - n_rank) : j;
    const int cj = __builtin_amdgcn_readlane(ci, sl);
    if (cj >= 0 && lane < d)
      __builtin_amdgcn_global_load_lds(x + (int64_t)cj * ld + lane, SCAMD_LDS_PTR(rows + j * RS), 4, 0, 0);
  }
  if (l31 < 1) skth[half] = INFINITY;
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();

  const float* myq = qrows + half * RS;
  // ||fl(q - mu)||^2: the norm of the query's image row (the thresholds live in that frame).  The 32 lanes of the query's half
  // share the coordinates and meet in a fixed butterfly (every lane used to walk all d of them: a quarter of the kernel's
  // instructions for one number per query)
  double qn = 0.0;
  for (int c = l31; c < d; c += 32) {
    const double qc = (double)__fsub_rn(myq[c], mu[c]);
    qn += qc * qc;
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) qn += __shfl_xor(qn, o);
  double myd = INFINITY;
  int myi = -1;
  if (ci >= 0) {
    const float* myrow = rows + (half * n_rank + l31) * RS;
    double s = 0.0;
    for (int c = 0; c < d; ++c) {
      const double df = (double)myq[c] - (double)myrow[c];
      s = fma(df, df, s);
    }
    myd = s;
    myi = ci;
    if (!(myd == myd)) {  // NaN distances sort last and are reported as missing
      myd = INFINITY;
      myi = -1;
    }
  }
  if (l31 < n_rank) {
    sd[half * n_rank + l31] = myd;
    si[half * n_rank + l31] = myi;
  }
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();

  const int kk = k - 1;  // neighbours besides self
  if (qv && l31 == 0) {
    out_idx[(int64_t)qi * k] = (int32_t)q;
    out_dist[(int64_t)qi * k] = 0.0;
  }
  if (qv && l31 < n_rank) {
    int rank = 0;
    for (int v = 0; v < n_rank; ++v) {
      const double dv = sd[half * n_rank + v];
      const int iv = si[half * n_rank + v];
      // missing entries (idx -1, +inf) order after everything, ties among them by position
      const bool less = (iv >= 0) ? (myi < 0 || key_less(dv, iv, myd, myi)) : (myi < 0 && v < l31);
      rank += less ? 1 : 0;
    }
    if (rank < kk) {
      out_idx[(int64_t)qi * k + 1 + rank] = myi;
      out_dist[(int64_t)qi * k + 1 + rank] = (myi >= 0) ? sqrt(myd) : INFINITY;
      if (rank == kk - 1) skth[half] = myd;
    }
  }
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  if (qv && l31 == 0) {
    // the certificate: see knn_rerank_kernel (same expressions)
    const double dk = (kk > 0) ? skth[half] : 0.0;
    const double tau = (double)cand_tau[qi];
    double cmax = (double)__uint_as_float(*cmax_bits);
    {
      const double rq = sqrt(qn) + sqrt(dk);
      const double cq = rq * rq * (1.0 + 1e-3);
      if (cq < cmax) cmax = cq;
    }
    const double eps = cert_scale * 5.9604644775390625e-08 *
                       (cert_k * (cmax + 2.0 * sqrt(qn * cmax)) + cert_k2 * 2.0 * sqrt(qn * cmax) + 128.0 * fabs(tau));
    const bool certified = (tau >= 1e38f) || ((dk - qn) + eps < tau);
    kth_d2[qi] = dk;
    if (!certified) {
      const int slot = atomicAdd(n_flag, 1);
      flag_list[slot] = qi;
    }
  }
}
static size_t rerank_rows_lds_bytes(int d, int n_rank) {
  const int RS = d | 1;
  const int wave_floats = (2 * n_rank + 2) * RS + 2 + 2 * n_rank * 3 + 4;
  return (size_t)4 * ((wave_floats + 3) & ~3) * sizeof(float);
}

// ------------------------------------------------------------------------------------------------
// pass 3: float64 scan for uncertified queries, two kernels.
//   scan: grid (row chunks, queries): every block scans one chunk of the rows and appends each row whose exact d^2
//         is <= bound (the k-th exact distance among the candidates, an upper bound of the true k-th distance)
//         to the query's collection (one global counter per query; order irrelevant, the rank step sorts);
//   rank: one block per query rank-sorts the collection by (distance, index) and writes the k-1 best.
// ------------------------------------------------------------------------------------------------
constexpr int FALLBACK_ROW_CHUNKS = 128;

__global__ __launch_bounds__(256) void knn_fallback_scan_kernel(
    const float* __restrict__ x, int64_t n, int d, int64_t ld, int64_t q_begin,
    const int* __restrict__ flag_list, int flag_begin, const double* __restrict__ kth_d2,
    double* __restrict__ scratch_d, int* __restrict__ scratch_i, int* __restrict__ counts) {
  __shared__ float qs[KNN_MAX_D];
  const int fb = blockIdx.y;
  const int64_t qi = flag_list[flag_begin + fb];
  const int64_t q = q_begin + qi;
  double* bd = scratch_d + (int64_t)fb * FALLBACK_CAP;
  int* bi = scratch_i + (int64_t)fb * FALLBACK_CAP;
  for (int c = threadIdx.x; c < d; c += blockDim.x) qs[c] = x[q * ld + c];
  __syncthreads();
  const double bound = kth_d2[qi];
  const int64_t rows_per_chunk = (n + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_chunk, r1 = std::min<int64_t>(n, r0 + rows_per_chunk);
  for (int64_t r = r0 + threadIdx.x; r < r1; r += blockDim.x) {
    if (r == q) continue;
    const float* cp = x + r * ld;
    double s = 0.0;
    for (int c = 0; c < d; ++c) {
      double df = (double)qs[c] - (double)cp[c];
      s = fma(df, df, s);
    }
    if (s <= bound) {
      int slot = atomicAdd(&counts[fb], 1);
      if (slot < FALLBACK_CAP) {
        bd[slot] = s;
        bi[slot] = (int)r;
      }
    }
  }
}

// The same scan restricted to the cells that can hold a row within the bound (cell-pruned mode): a cell whose ball
// lies farther from the query than sqrt(bound) is skipped, |q - x| >= |q - mu_c| - r_c for every member x.  The float64
// scan over ALL rows cost 0.7 ms per query at 10M rows (1.1 s for the 1499 uncertified queries of the 10M x 50 run);
// with ~2.5 % of the rows in reach it is noise again.  grid (cell chunks, queries).
__global__ __launch_bounds__(256) void knn_fallback_scan_cells_kernel(
    const float* __restrict__ x, int d, int64_t ld, int64_t q_begin, const int* __restrict__ flag_list, int flag_begin,
    const double* __restrict__ kth_d2, double* __restrict__ scratch_d, int* __restrict__ scratch_i, int* __restrict__ counts,
    const float* __restrict__ cent, const float* __restrict__ radius, const int* __restrict__ cell_tile0,
    const int* __restrict__ cell_ntiles, const int* __restrict__ perm, int n_cells) {
  __shared__ float qs[KNN_MAX_D];
  const int fb = blockIdx.y;
  const int64_t qi = flag_list[flag_begin + fb];
  const int64_t q = q_begin + qi;
  double* bd = scratch_d + (int64_t)fb * FALLBACK_CAP;
  int* bi = scratch_i + (int64_t)fb * FALLBACK_CAP;
  for (int c = threadIdx.x; c < d; c += blockDim.x) qs[c] = x[q * ld + c];
  __syncthreads();
  const double bound = kth_d2[qi];
  const int per = (n_cells + gridDim.x - 1) / gridDim.x;
  const int c0 = blockIdx.x * per, c1 = min(n_cells, c0 + per);
  for (int c = c0; c < c1; ++c) {
    const int nt = cell_ntiles[c];
    if (nt == 0) continue;
    double dc2 = 0.0;  // (every thread: uniform, 50 fused multiply-adds)
    for (int t = 0; t < d; ++t) {
      const double df = (double)qs[t] - (double)cent[c * d + t];
      dc2 = fma(df, df, dc2);
    }
    const double lb = sqrt(dc2) * (1.0 - 1e-6) - (double)radius[c];  // (the radius is stored inflated by 1e-4)
    if (lb > 0.0 && lb * lb > bound * (1.0 + 1e-9)) continue;
    const int64_t r0 = (int64_t)cell_tile0[c] * 64, r1 = r0 + (int64_t)nt * 64;
    for (int64_t row = r0 + threadIdx.x; row < r1; row += blockDim.x) {
      const int orig = perm[row];
      if (orig < 0 || orig == q) continue;
      const float* cp = x + (int64_t)orig * ld;
      double s = 0.0;
      // ten coordinates requested at a time, summed in the same order as one by one (one load in flight per thread made a
      // cell of 2048 rows 400 serial round trips per thread: 0.9 ms for the 58 queries of the 1M bench)
      int t = 0;
      for (; t + 10 <= d; t += 10) {
        float v[10];
#pragma unroll
        for (int u = 0; u < 10; ++u) v[u] = cp[t + u];
#pragma unroll
        for (int u = 0; u < 10; ++u) {
          const double df = (double)qs[t + u] - (double)v[u];
          s = fma(df, df, s);
        }
      }
      for (; t < d; ++t) {
        const double df = (double)qs[t] - (double)cp[t];
        s = fma(df, df, s);
      }
      if (s <= bound) {
        const int slot = atomicAdd(&counts[fb], 1);
        if (slot < FALLBACK_CAP) {
          bd[slot] = s;
          bi[slot] = orig;
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void knn_fallback_rank_kernel(
    int k, const int* __restrict__ flag_list, int flag_begin, const double* __restrict__ scratch_d,
    const int* __restrict__ scratch_i, const int* __restrict__ counts, int32_t* __restrict__ out_idx,
    double* __restrict__ out_dist, double* __restrict__ kth_d2, int* __restrict__ retry_list, int* __restrict__ n_retry) {
  const int fb = blockIdx.x;
  const int64_t qi = flag_list[flag_begin + fb];
  const double* bd = scratch_d + (int64_t)fb * FALLBACK_CAP;
  const int* bi = scratch_i + (int64_t)fb * FALLBACK_CAP;
  int m = counts[fb];
  // More rows within the bound than the table holds: the bound came from a list the scoring engine could not order
  // (norms far larger than the neighbour distances: clusters at +-3000 with unit spread -- tests/test_gpu_knn_certificate.py).
  // The k-1 smallest of ANY FALLBACK_CAP rows bound the true (k-1)-th distance from above: tighten the bound to that and scan
  // again (each round keeps about (k - 1) / FALLBACK_CAP of the rows; only more than FALLBACK_CAP rows tied AT the k-th
  // distance cannot be resolved, which the host reports after a few rounds).
  const bool overflow = m > FALLBACK_CAP;
  if (overflow) m = FALLBACK_CAP;
  const int kk = k - 1;
  for (int u = threadIdx.x; u < m; u += blockDim.x) {
    double du = bd[u];
    int iu = bi[u];
    int rank = 0;
    for (int v = 0; v < m; ++v) rank += key_less(bd[v], bi[v], du, iu) ? 1 : 0;
    if (overflow) {
      if (rank == kk - 1) kth_d2[qi] = du;
    } else if (rank < kk) {
      out_idx[qi * k + 1 + rank] = iu;
      out_dist[qi * k + 1 + rank] = sqrt(du);
    }
  }
  if (overflow && threadIdx.x == 0) retry_list[atomicAdd(n_retry, 1)] = (int)qi;
}

// ------------------------------------------------------------------------------------------------
// coarse quantiser of the cell-pruned search: a few Lloyd iterations on a sample, then one assignment of all rows.
// Quality only affects how much is pruned, never the result.
// ------------------------------------------------------------------------------------------------
constexpr double IVF_FIX = 1048576.0;  // 2^20 fixed point of the centroid sums (order-independent atomics)

__global__ void ivf_init_kernel(const float* __restrict__ x, int64_t n, int d, int64_t ld, int n_cells,
                                float* __restrict__ cent) {
  const int c = blockIdx.x;
  const int64_t row = (int64_t)c * (n / n_cells) + (n / n_cells) / 2;
  for (int j = threadIdx.x; j < d; j += blockDim.x) cent[c * d + j] = x[row * ld + j];
}

// centroid table of the assignment kernel: [n_cells][DP] rows of float4 = the centroid, zeros, |c|^2 / 2 in the last column
__global__ void ivf_centpad_kernel(const float* __restrict__ cent, int n_cells, int d, int DP, float* __restrict__ centp) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_cells * DP) return;
  const int c = e / DP, t = e - c * DP;
  float v = 0.f;
  if (t < d) {
    v = cent[c * d + t];
  } else if (t == DP - 1) {
    float hn = 0.f;
    for (int u = 0; u < d; ++u) hn = fmaf(cent[c * d + u], cent[c * d + u], hn);
    v = 0.5f * hn;
  }
  centp[e] = v;
}

// One thread per sampled row (row = start + j * step, j < count), the row in registers.
// nearest centroid = argmax_c (x . c - |c|^2 / 2): ONE fused multiply-add per dimension, two dimensions per
// v_pk_fma_f32 with the centroid as the SCALAR operand (the table is the same for every lane: s_load through the
// constant cache, no LDS).  History: (x - c)^2 from an LDS-staged table was a subtract and an fma per dimension,
// unpacked -- 2.4 ms of VALU issue at 1M x 1024 x 50; the packed dot product from LDS 1.7 ms, bound by the broadcast
// ds_read_b128 (13 per centroid and wave).  The half norm rides in the last column against a -1 in the row.  A
// different rounding than the difference form can move a row between two almost equidistant cells; cells only steer
// the pruning, never the result.
// accumulate != 0: fixed-point sums / counts of the Lloyd update.  qcounts != NULL: additionally counts the rows of the
// query range [q0, q1).
template <int H>
__global__ __launch_bounds__(256) void ivf_assign_kernel(const float* __restrict__ x, int d, int64_t ld, int64_t start,
                                                         int64_t step, int64_t count, const float* __restrict__ centp,
                                                         int n_cells, int* __restrict__ labels, int accumulate,
                                                         long long* __restrict__ sums, int* __restrict__ counts,
                                                         int64_t q0, int64_t q1, int* __restrict__ qcounts) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  constexpr int DP = (2 * H + 1 + 3) / 4 * 4;  // dims + the norm column, padded to float4
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool valid = j < count;
  const int64_t row = valid ? start + j * step : start;
  float xr[DP];
#pragma unroll
  for (int c = 0; c < DP; ++c) xr[c] = (c < d) ? x[row * ld + c] : 0.f;
  xr[DP - 1] = -1.0f;
  float best = -INFINITY;
  int bi = 0;
  for (int c = 0; c < n_cells; ++c) {
    const float4* cp = reinterpret_cast<const float4*>(centp + (int64_t)c * DP);
    f2 s = {0.f, 0.f};
#pragma unroll
    for (int t4 = 0; t4 < DP / 4; ++t4) {
      const float4 v = cp[t4];
      const f2 xa = {xr[4 * t4 + 0], xr[4 * t4 + 1]}, xb = {xr[4 * t4 + 2], xr[4 * t4 + 3]};
      const f2 va = {v.x, v.y}, vb = {v.z, v.w};
      s = __builtin_elementwise_fma(xa, va, s);
      s = __builtin_elementwise_fma(xb, vb, s);
    }
    const float score = s.x + s.y;
    if (score > best) {
      best = score;
      bi = c;
    }
  }
  if (!valid) return;
  labels[j] = bi;
  if (accumulate) {
#pragma unroll
    for (int t = 0; t < DP; ++t)
      if (t < d)
        atomicAdd(reinterpret_cast<unsigned long long*>(&sums[(int64_t)bi * d + t]),
                  (unsigned long long)llrint((double)xr[t] * IVF_FIX));
  }
  if (counts) atomicAdd(&counts[bi], 1);
  if (qcounts && row >= q0 && row < q1) atomicAdd(&qcounts[bi], 1);
}

// ------------------------------------------------------------------------------------------------
// The assignment of ALL rows on the bf16 matrix cores (round 4; d <= 50).  Any assignment of rows to cells gives a correct
// search -- cells only steer the pruning, the radii are computed from the rows a cell actually received -- so the nearest
// centroid may be taken from bf16 products: score(x, c) = bf16(x) . bf16(c) - |c|^2 / 2 (the half norm exact: three bf16
// pieces in the padding dimensions, as the thresholds of the select kernel).  A wave owns 32 rows as the A operand, the
// centroids are the B operand from an LDS table: 4 MFMAs + 48 VALU per 32 x 32 scores instead of 26 packed fmas per score
// (ivf_assign_kernel: 1.5 ms at 1M x 512; kept for the sampled Lloyd iterations, which also accumulate the sums).
// ------------------------------------------------------------------------------------------------
constexpr int CENT_DPL = 36;  // dwords per centroid row in LDS: 32 of bf16 pairs + 4 (stride 36 = 4 mod 32: conflict-free b128 reads)
__global__ void ivf_centpad_bf16_kernel(const float* __restrict__ cent, int n_cells, int n_cells_pad, int d,
                                        unsigned int* __restrict__ centb /* [n_cells_pad][32] */) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_cells_pad * 32) return;
  const int c = e >> 5, w = e & 31;
  if (c >= n_cells) {  // padding centroid: can never win (score -> -1.7e38)
    centb[e] = w == 26 ? (0xFF00u << 16) : (w == 25 ? 0x3F803F80u : 0u);  // dims 50..52 = 1, dim 53 = most negative bf16 exponent below inf
    return;
  }
  float hn = 0.f;
  for (int u = 0; u < d; ++u) hn = fmaf(cent[c * d + u], cent[c * d + u], hn);
  centb[e] = b3_row_dword(w, d, -0.5f * hn, [&](int dim) { return cent[c * d + dim]; });
}

// rows j = 0 .. count-1 are x[start + j * step]; labels[j] = the row's cell.  A workgroup stages the centroid table once
// and assigns ASSIGN_ROWS rows (a first version staged it per 128 rows: 0.73 ms at 1M x 512, slower than the float32
// kernel's 0.67 -- the 74 KB table load was the kernel).
constexpr int ASSIGN_ROWS = 512;
__global__ __launch_bounds__(256) void ivf_assign_mfma_kernel(const float* __restrict__ x, int d, int64_t ld, int64_t start,
                                                              int64_t step, int64_t count,
                                                              const unsigned int* __restrict__ centb, int n_cells_pad,
                                                              int n_cells, int* __restrict__ labels, int* __restrict__ counts,
                                                              int64_t q0, int64_t q1, int* __restrict__ qcounts) {
  extern __shared__ __attribute__((aligned(16))) unsigned int ctab[];  // [n_cells_pad][CENT_DPL]
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = tid >> 6;
  {
    const uint4* src = reinterpret_cast<const uint4*>(centb);
    uint4* dst = reinterpret_cast<uint4*>(ctab);
    for (int e = tid; e < n_cells_pad * 8; e += 256) dst[(e >> 3) * (CENT_DPL / 4) + (e & 7)] = src[e];
  }
  __syncthreads();
  const int n_sub = n_cells_pad / 32;
  for (int grp = 0; grp < ASSIGN_ROWS / 128; ++grp) {
    // A operand: row (l31) of this wave, dims 16 s + 8 half .. + 8 of k-step s as four bf16 pairs (hi part only)
    const int64_t j0 = (int64_t)blockIdx.x * ASSIGN_ROWS + grp * 128 + wave * 32;
    if (j0 >= count) break;  // (whole wave)
    const int64_t jr = std::min<int64_t>(j0 + l31, count - 1);
    const float* xr = x + (start + jr * step) * ld;
    i32x4 qa[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int dim = 16 * s4 + 8 * half + 2 * j;
        const float v0 = dim < d ? xr[dim] : 0.f, v1 = dim + 1 < d ? xr[dim + 1] : 0.f;
        qa[s4][j] = (int)(bf16_rn(v0) | (bf16_rn(v1) << 16));
      }
    }
    if (half == 0) {  // dims 50..52 (the centroid side holds 1, 1, 1): 0; dims 53..55 (the half norm's three pieces): 1
      qa[3][1] = 0;
      qa[3][2] = 0x3F800000;
      qa[3][3] = 0x3F803F80;
    }
    float best[16];
    int bg[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      best[r] = -INFINITY;
      bg[r] = 0;
    }
    for (int g = 0; g < n_sub; ++g) {
      const i32x4* p = reinterpret_cast<const i32x4*>(ctab + (g * 32 + l31) * CENT_DPL);
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, qa[s4]), __builtin_bit_cast(bf16x8, p[2 * s4 + half]), acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool better = acc[r] > best[r];  // (strict: the first sub-tile wins a tie)
        best[r] = better ? acc[r] : best[r];
        bg[r] = better ? g : bg[r];
      }
    }
    // per row (register r, half): the best of the 32 lanes' centroids (ties: the smallest centroid id)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = best[r];
      int id = bg[r] * 32 + l31;
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o);
        const int oid = __shfl_xor(id, o);
        const bool take = ov > v || (ov == v && oid < id);
        v = take ? ov : v;
        id = take ? oid : id;
      }
      const int64_t jj = j0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      // (padding centroids score a finite -1.7e38: rows whose scores against every real centroid overflow to -inf --
      // magnitudes around 1e19 -- would elect one; any real cell gives a correct search, an id past the tables does not)
      id = min(id, n_cells - 1);
      if (l31 == 0 && jj < count) {
        const int64_t row = start + jj * step;
        labels[jj] = id;
        if (counts) {  // (the sampled Lloyd iterations; the assignment of every row counts afterwards: ivf_count_kernel)
          atomicAdd(&counts[id], 1);
          if (qcounts && row >= q0 && row < q1) atomicAdd(&qcounts[id], 1);
        }
      }
    }
  }
}

// fixed-point coordinate sums of the Lloyd update from the labels of the sampled rows (order-independent atomics)
__global__ void ivf_sums_kernel(const float* __restrict__ x, int d, int64_t ld, int64_t start, int64_t step, int64_t count,
                                const int* __restrict__ labels, long long* __restrict__ sums) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= count * d) return;
  const int64_t j = e / d;
  const int t = (int)(e - j * d);
  atomicAdd(reinterpret_cast<unsigned long long*>(&sums[(int64_t)labels[j] * d + t]),
            (unsigned long long)llrint((double)x[(start + j * step) * ld + t] * IVF_FIX));
}

__global__ void ivf_update_kernel(const long long* __restrict__ sums, const int* __restrict__ counts, int n_cells, int d,
                                  float* __restrict__ cent) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_cells * d) return;
  const int c = e / d;
  if (counts[c] > 0) cent[e] = (float)((double)sums[e] / IVF_FIX / (double)counts[c]);
}

// position of every row in the cell-sorted image (+ of the query rows in the query list); order inside a cell is
// arbitrary
// rows (and query rows) per cell of the final assignment: per-workgroup counters in LDS, one flush per cell and workgroup
// (the assignment kernel's own two atomics per row on ~500 words were most of its 0.69 ms at 1M rows)
__global__ __launch_bounds__(1024) void ivf_count_kernel(const int* __restrict__ labels, int64_t n, int n_cells, int64_t q0, int64_t q1,
                                                        int* __restrict__ counts, int* __restrict__ qcounts) {
  __shared__ int l_row[IVF_MAX_CELLS], l_q[IVF_MAX_CELLS];
  for (int c = threadIdx.x; c < n_cells; c += 1024) l_row[c] = 0, l_q[c] = 0;
  __syncthreads();
  const int64_t i0 = (int64_t)blockIdx.x * 4096 + threadIdx.x;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int64_t i = i0 + (int64_t)t * 1024;
    if (i < n) {
      const int c = labels[i];
      atomicAdd(&l_row[c], 1);
      if (i >= q0 && i < q1) atomicAdd(&l_q[c], 1);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < n_cells; c += 1024) {
    if (l_row[c]) atomicAdd(&counts[c], l_row[c]);
    if (l_q[c]) atomicAdd(&qcounts[c], l_q[c]);
  }
}
// Round 6: positions through per-workgroup counters in LDS (a workgroup ranks 4096 rows per cell, then takes ONE range per cell
// from the global cursors): 8 x fewer returning atomics on the ~500 cursor words every row used to hit (0.43 ms at 1M rows).
// The order of the rows inside a cell is arbitrary either way (the search is exact for any order).
constexpr int SCATTER_ROWS = 4;  // rows per thread
__global__ __launch_bounds__(1024) void ivf_scatter_kernel(const int* __restrict__ labels, int64_t n, const int* __restrict__ cell_map,
                                                          const int* __restrict__ row_off, int* __restrict__ row_cur, int64_t q0, int64_t q1,
                                                          const int* __restrict__ slot_off, int* __restrict__ slot_cur, int* __restrict__ perm,
                                                          int* __restrict__ qpos, int* __restrict__ qrow, int n_cells) {
  __shared__ int l_row[IVF_MAX_CELLS], l_slot[IVF_MAX_CELLS];
  for (int c = threadIdx.x; c < n_cells; c += 1024) l_row[c] = 0, l_slot[c] = 0;
  __syncthreads();
  const int64_t i0 = (int64_t)blockIdx.x * (1024 * SCATTER_ROWS) + threadIdx.x;
  int cell[SCATTER_ROWS], rk[SCATTER_ROWS], qk[SCATTER_ROWS];
#pragma unroll
  for (int t = 0; t < SCATTER_ROWS; ++t) {
    const int64_t i = i0 + (int64_t)t * 1024;
    cell[t] = -1, rk[t] = 0, qk[t] = -1;
    if (i < n) {
      cell[t] = cell_map[labels[i]];
      rk[t] = atomicAdd(&l_row[cell[t]], 1);
      if (i >= q0 && i < q1) qk[t] = atomicAdd(&l_slot[cell[t]], 1);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < n_cells; c += 1024) {
    const int nr = l_row[c], ns = l_slot[c];
    l_row[c] = nr ? atomicAdd(&row_cur[c], nr) : 0;
    l_slot[c] = ns ? atomicAdd(&slot_cur[c], ns) : 0;
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < SCATTER_ROWS; ++t) {
    const int64_t i = i0 + (int64_t)t * 1024;
    if (cell[t] < 0) continue;
    const int c = cell[t];
    const int pos = row_off[c] + l_row[c] + rk[t];
    perm[pos] = (int)i;
    if (qk[t] >= 0) {
      qpos[slot_off[c] + l_slot[c] + qk[t]] = pos;
      if (qrow) qrow[i - q0] = pos;  // image row of the query (second tier of the bf16 engine)
    }
  }
}

// second tier: the flagged queries, counted per (merged) cell ...
__global__ void ivf_t2_count_kernel(const int* __restrict__ flag_list, int n_flag, int64_t q0, const int* __restrict__ labels,
                                    const int* __restrict__ cell_map, int* __restrict__ cnt, int* __restrict__ cell_out,
                                    int* __restrict__ pos_out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_flag) return;
  const int c = cell_map[labels[q0 + flag_list[j]]];
  cell_out[j] = c;
  pos_out[j] = atomicAdd(&cnt[c], 1);
}
// ... and written into their cell's query blocks (slot -> image row; untouched slots stay -1)
__global__ void ivf_t2_fill_kernel(const int* __restrict__ flag_list, int n_flag, const int* __restrict__ cell_in,
                                   const int* __restrict__ pos_in, const int* __restrict__ slot_off,
                                   const int* __restrict__ qrow, int* __restrict__ qpos) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_flag) return;
  qpos[slot_off[cell_in[j]] + pos_in[j]] = qrow[flag_list[j]];
}
__global__ void knn_iota_kernel(int* __restrict__ a, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = i;
}

// image of the cell-sorted rows (layout of knn_pack_image_kernel); padding rows get ||c||^2 = +inf.  Also the
// cell radii: max distance of a member to its cell's centre (float32, as uint bits for atomicMax).
// 16 lanes per row, four rows per wave: the chain perm -> labels -> cell_map -> x row is four dependent gathers, and a
// wave that walks its rows one at a time has one of them in flight (1.43 ms at 1M x 50 for 0.4 GB of traffic).
__global__ void ivf_pack_image_kernel(const float* __restrict__ x, const float* __restrict__ mu, int d, int64_t ld,
                                      int H, int HP, int DPL,
                                      int64_t n_img, const int* __restrict__ perm, const int* __restrict__ labels,
                                      const int* __restrict__ cell_map, const float* __restrict__ cent,
                                      float* __restrict__ xp, unsigned int* __restrict__ cmax_bits,
                                      unsigned int* __restrict__ radius_bits, int b3) {
  const int lane = threadIdx.x & 63, sub = lane & 15;
  const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int64_t ngrp = ((int64_t)gridDim.x * blockDim.x) >> 4;
  float wmax = 0.f;
  // a group of 16 lanes packs a CONTIGUOUS run of image rows (sorted by cell: one or two cells per run) and keeps the run's
  // largest centre distance per cell in registers -- one atomic per (group, cell) instead of a read and often an atomic per
  // row on the same word as every other group working in that cell (round 6: the kernel got SLOWER with more workgroups,
  // 0.81 / 1.01 / 2.90 ms at 4096 / 16384 / 65536)
  const int64_t rpg = (n_img + ngrp - 1) / ngrp;
  const int64_t r_lo = grp * rpg, r_hi = r_lo + rpg < n_img ? r_lo + rpg : n_img;
  int run_cell = -1;
  unsigned int run_rb = 0u;
  auto flush_radius = [&]() {
    if (sub == 0 && run_cell >= 0 && run_rb > 0u) atomicMax(&radius_bits[run_cell], run_rb);
  };
  int src_next = r_lo < r_hi ? perm[r_lo] : -1;
  for (int64_t r = r_lo; r < r_hi; ++r) {
    const int src = src_next;
    src_next = r + 1 < r_hi ? perm[r + 1] : -1;
    double s = 0.0;
    float dc2 = 0.f;
    int cell = 0;
    if (src >= 0) {
      cell = cell_map[labels[src]];
      for (int c = sub; c < d; c += 16) {
        const float v = x[(int64_t)src * ld + c];
        const float vc = __fsub_rn(v, mu[c]);  // the image row (centred); the cell geometry stays in x's own frame
        s += (double)vc * (double)vc;
        const float df = v - cent[cell * d + c];
        dc2 = fmaf(df, df, dc2);
      }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      s += __shfl_xor(s, o);
      dc2 += __shfl_xor(dc2, o);
    }
    const float nf = (src >= 0) ? (float)s : INFINITY;
    if (b3) {  // bf16 hi / lo image (RegCfg: B3)
      const float nb = (src >= 0) ? nf : B3_PAD_NORM;
      unsigned int* xu = reinterpret_cast<unsigned int*>(xp);
      for (int c = sub; c < B3_DPL; c += 16)
        xu[r * B3_DPL + c] =
            b3_row_dword(c, d, nb, [&](int dim) { return src >= 0 ? __fsub_rn(x[(int64_t)src * ld + dim], mu[dim]) : 0.f; });
    } else
    for (int c = sub; c < DPL; c += 16) {
      const int hh = c / HP, cc = c - hh * HP;
      float v = 0.f;
      if (hh < 2) {
        const int dim = hh * H + cc;
        if (cc < H) v = (src >= 0 && dim < d) ? __fsub_rn(x[(int64_t)src * ld + dim], mu[dim]) : 0.f;
        else if (cc == H) v = (hh == 0) ? 1.0f : nf;
      }
      xp[r * DPL + c] = v;
    }
    if (src >= 0) {
      wmax = fmaxf(wmax, nf);
      const unsigned int rb = __float_as_uint(sqrtf(dc2) * 1.0001f + 1e-6f);
      if (cell != run_cell) {
        flush_radius();
        run_cell = cell;
        run_rb = 0u;
      }
      run_rb = max(run_rb, rb);
    }
  }
  flush_radius();
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o));
  // (read before the atomic: the maximum only grows, a stale read costs an atomic and never loses one -- 16k same-word
  // atomics, one per wave, were a third of this kernel)
  if (lane == 0 && wmax > 0.f && __float_as_uint(wmax) > __hip_atomic_load(cmax_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    atomicMax(cmax_bits, __float_as_uint(wmax));
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct KnnPlan {
  int H, TC, NW, KP;
  bool reg;       // register-list kernel (knn_select_reg_kernel) instead of the LDS-list kernel
  bool b3;        // ... with the 3 x bf16 scoring engine (RegCfg: B3)
  bool thr_margin_env;
  int thr_rank;   // register-list kernel: rank (1..32) of the list entry used as the filter threshold
  int row_dwords; // row stride of the packed copy
  int64_t n_pad, nq_pad;
  bool ivf;       // cell-pruned exact search (register-list kernel only)
  int nprobe;     // > 0: approximate mode, every query block sweeps its cell's nprobe nearest cells only
  int n_cells;    // coarse cells
  int64_t n_img_max, n_slot_max;  // upper bounds of the padded image rows / query slots
};

template <int H>
static void reg_plan(KnnPlan* p) {
  p->row_dwords = RegCfg<H>::DPL;
}

static bool knn_plan(int64_t n, int d, int64_t n_query, int k, KnnPlan* p, int nprobe = 0) {
  p->nprobe = 0;
  if (d <= 16) p->H = 8;
  else if (d <= 32) p->H = 16;
  else if (d <= 50) p->H = 25;
  else if (d <= 64) p->H = 32;
  else if (d <= 128) p->H = 64;
  else if (d <= 256) p->H = 128;
  else return false;
  p->TC = (p->H == 128) ? 32 : ((p->H == 64) ? 64 : 128);
  if (k <= 24) { p->KP = 32; p->NW = 8; }
  else if (k <= 56) { p->KP = 64; p->NW = 4; }
  else if (k <= 120) { p->KP = 128; p->NW = 2; }
  else if (k <= 256) { p->KP = 288; p->NW = 1; }  // (one wave per block: the lists of 32 queries take 72 KB of LDS)
  else return false;
  // d > 128: the query operand and one fragment take 256 registers -- blocks of at most two waves (512 VGPRs per lane)
  if (p->H == 128 && p->NW > 2) {
    p->NW = 2;
    p->KP = std::max(p->KP, 128);
  }
  static const bool legacy = [] {
    const char* e = getenv("SCAMD_KNN_LEGACY");
    return e && e[0] == '1';
  }();
  p->reg = (p->KP == 32 && p->H <= 32 && !legacy);
  {
    // k columns = self + k-1 others must be certified below the threshold: keep a margin of 6 ranks
    static const int margin = [] {
      const char* e = getenv("SCAMD_KNN_THR_MARGIN");
      return e ? atoi(e) : 6;
    }();
    p->thr_rank = std::min(32, std::max(1, k + margin));
    p->thr_margin_env = getenv("SCAMD_KNN_THR_MARGIN") != nullptr;
  }
  p->row_dwords = 2 * p->H;
  if (p->reg) {
    p->NW = 4;
    p->TC = 128;
    switch (p->H) {
      case 8: reg_plan<8>(p); break;
      case 16: reg_plan<16>(p); break;
      case 25: reg_plan<25>(p); break;
      default: reg_plan<32>(p); break;
    }
  }
  // 3 x bf16 engine: 32 < d <= 50 (the PCA embedding); SCAMD_KNN_B3=0 keeps the float32 MFMA engine (read per call)
  p->b3 = false;
  if (p->reg && p->H == 25) {
    const char* e = getenv("SCAMD_KNN_B3");
    p->b3 = !(e && e[0] == '0');
    if (p->b3) {
      p->row_dwords = B3_DPL;
      // The engine's error bound is ~4x the float32 engine's (relative to ||q|| ||c||, which on clustered data is far
      // larger than the neighbour distances): with the margin of 6 ranks, 508 queries of the planted 1M run failed the
      // certificate under a bound that was still 2x too small, and the float64 scan of those cost 4.2 ms.  The failure
      // probability falls like (bound / gap)^margin.  Measured with the correct bound, planted 1M: margin 8 -> select
      // 16.5 ms + 200 fallbacks, 10 -> 17.0 ms + 5, 12 -> 18.0 ms + 1.  Round 4: the bound prices the norms a missed
      // neighbour can have instead of the largest norm of the data set (knn_rerank_rows_kernel), which rejects fewer
      // queries at every margin -- one box, planted 1M: margin 10 -> 16.97 ms + 1 float64 scan, 8 -> 16.30 ms + 58,
      // 6 -> 15.89 ms + 1423 queries through the second tier (which costs what the margin saved), 4 -> 15.41 ms + 25764
      // (profiles/r04e_knn_knobs.log): 8 ranks (threshold = the 23rd list entry at k = 15).
      if (!p->thr_margin_env) p->thr_rank = std::min(32, std::max(1, k + 8));
    }
  }
  const int QB = p->NW * 32;
  p->nq_pad = (n_query + QB - 1) / QB * QB;
  p->n_pad = (n + 255) / 256 * 256;
  // cell pruning pays once a sweep is long compared with the per-cell restarts; SCAMD_KNN_IVF=0 forces brute force
  const int ivf_env = [] {  // read per call: the tests flip it inside one process
    const char* e = getenv("SCAMD_KNN_IVF");
    return e ? atoi(e) : -1;
  }();
  p->ivf = p->reg && ivf_env != 0 && (n >= 65536 || ivf_env == 1) && n >= 4096;
  // approximate mode: always through the cell tables (register-list kernel: k <= 24, d <= 64, and n >= 4096 rows -- below
  // that, and for the other shapes, the call is answered exactly)
  if (nprobe > 0 && p->reg && n >= 4096) {
    p->ivf = true;
    p->nprobe = nprobe;
  }
  p->n_cells = 0;
  p->n_img_max = p->n_slot_max = 0;
  if (p->ivf) {
    static const int cell_rows = [] {
      const char* e = getenv("SCAMD_KNN_CELL_ROWS");
      return e ? std::max(256, atoi(e)) : 2048;
    }();
    int c = 16;
    while (c < IVF_MAX_CELLS && (int64_t)c * cell_rows < n) c <<= 1;  // ~cell_rows rows per cell
    while (c > 1 && (int64_t)c * 256 > n) c >>= 1;
    p->n_cells = c;
    p->n_img_max = (n + (int64_t)64 * c + 255) / 256 * 256;
    p->n_slot_max = n_query + (int64_t)128 * c;
    p->n_pad = p->n_img_max;
  }
  return true;
}

struct KnnBuffers {
  float* xp; float* cn; float* mu; double* mean_partial; unsigned int* cmax; int* cand_idx; float* cand_tau; double* kth_d2;
  int* flag_list; int* counters; double* scratch_d; int* scratch_i; int* fb_counts;
  int* fb_retry[2];  // queries whose float64 scan overflowed its table: scanned again with a tighter bound
  // cell-pruned search
  int* labels; int* perm; int* qpos; int* block_cell; float* cent; float* centp; long long* sums; int* cell_ints;
  unsigned int* radius_bits; int* cell_order; float* cell_lb2; int* cell_aux; int* block_perm;
  int* qorder;  // [n_slot_max] query number of every query slot of the pruned sweep (-1 = padding): pass 2's order
  // second tier of the bf16 engine (float32 engine on the queries its certificate rejected)
  int* qrow; float* xp2; int* flag_list2; int* t2_cell; int* t2_pos; int* t2_ints;
};

static void knn_carve(Workspace& ws, const KnnPlan& p, int64_t n_query, KnnBuffers* b) {
  b->xp = ws.take<float>((size_t)p.n_pad * p.row_dwords);
  b->cn = ws.take<float>((size_t)p.n_pad);
  b->mu = ws.take<float>(KNN_MAX_D);
  b->mean_partial = ws.take<double>((size_t)MEAN_BLOCKS * KNN_MAX_D);
  b->cmax = ws.take<unsigned int>(4);
  b->cand_idx = ws.take<int>((size_t)p.nq_pad * p.KP);
  b->cand_tau = ws.take<float>((size_t)p.nq_pad);
  b->kth_d2 = ws.take<double>((size_t)n_query);
  b->flag_list = ws.take<int>((size_t)n_query);
  b->counters = ws.take<int>(16);  // [0] uncertified, [1] overflow, [2..3] swept pairs, [4..5] pre-pass pairs (u64), [6] launch-order error, [8..15] XCD queue positions
  b->scratch_d = ws.take<double>((size_t)FALLBACK_CHUNK * FALLBACK_CAP);
  b->scratch_i = ws.take<int>((size_t)FALLBACK_CHUNK * FALLBACK_CAP);
  b->fb_counts = ws.take<int>((size_t)FALLBACK_CHUNK);
  b->fb_retry[0] = ws.take<int>((size_t)n_query);
  b->fb_retry[1] = ws.take<int>((size_t)n_query);
  b->labels = b->perm = b->qpos = b->block_cell = b->cell_ints = nullptr;
  b->cent = b->centp = nullptr;
  b->sums = nullptr;
  b->radius_bits = nullptr;
  b->cell_order = nullptr;
  b->cell_lb2 = nullptr;
  b->cell_aux = b->block_perm = b->qorder = nullptr;
  if (p.ivf) {
    b->labels = ws.take<int>((size_t)p.n_pad);  // sample labels, then labels of all rows
    b->perm = ws.take<int>((size_t)p.n_img_max);
    b->qpos = ws.take<int>((size_t)p.n_slot_max);
    b->block_cell = ws.take<int>((size_t)(p.n_slot_max / 128 + 1));
    b->cent = ws.take<float>((size_t)p.n_cells * 128);
    b->centp = ws.take<float>((size_t)p.n_cells * 136);  // [n_cells][2H + norm column, padded to float4]
    b->sums = ws.take<long long>((size_t)p.n_cells * 128);
    b->cell_ints = ws.take<int>((size_t)p.n_cells * 8);  // counts, qcounts, map, row_off, row_cur, slot_off, slot_cur, tile0/ntiles reuse
    b->radius_bits = ws.take<unsigned int>((size_t)p.n_cells);
    b->cell_order = ws.take<int>((size_t)p.n_cells * p.n_cells);
    b->cell_lb2 = ws.take<float>((size_t)p.n_cells * p.n_cells);
    b->cell_aux = ws.take<int>((size_t)p.n_cells * 3);  // expected work, first block, block count of every cell
    b->block_perm = ws.take<int>((size_t)(p.n_slot_max / 128 + 1) * 2 + 64);  // (XCD-aware order: up to 8 x the longest queue)
    b->qorder = ws.take<int>((size_t)p.n_slot_max);
  }
  b->qrow = b->flag_list2 = b->t2_cell = b->t2_pos = b->t2_ints = nullptr;
  b->xp2 = nullptr;
  if (p.ivf && p.b3) {
    b->qrow = ws.take<int>((size_t)n_query);
    b->xp2 = ws.take<float>((size_t)p.n_img_max * RegCfg<25>::DPL);
    b->flag_list2 = ws.take<int>((size_t)n_query);
    b->t2_cell = ws.take<int>((size_t)n_query);
    b->t2_pos = ws.take<int>((size_t)n_query);
    b->t2_ints = ws.take<int>((size_t)p.n_cells * 2 + 8);  // per-cell counts, slot offsets; [2 nc ..] counters
  }
}

template <int H, int TC, int NW, int KP>
static int launch_select(const KnnPlan& p, const KnnBuffers& b, int64_t q_begin, hipStream_t s) {
  using C = SelectCfg<H, TC, NW, KP>;
  auto kern = knn_select_kernel<H, TC, NW, KP>;
  SCAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
  const int n_tiles = (int)(p.n_pad / TC);
  const int grid = (int)(p.nq_pad / C::QB);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), C::LDS_BYTES, s, b.xp, b.cn, n_tiles, p.n_pad,
                     q_begin, b.cand_idx, b.cand_tau);
  SCAMD_LAUNCH_CHECK();
  return SCAMD_OK;
}

template <int H, int TC>
static int dispatch_kp(const KnnPlan& p, const KnnBuffers& b, int64_t q_begin, hipStream_t s) {
  switch (p.KP) {
    case 32: return launch_select<H, TC, 8, 32>(p, b, q_begin, s);
    case 64: return launch_select<H, TC, 4, 64>(p, b, q_begin, s);
    case 128: return launch_select<H, TC, 2, 128>(p, b, q_begin, s);
    default: return launch_select<H, (TC < 64 ? TC : 64), 1, 288>(p, b, q_begin, s);  // (one wave stages at most 64 norms per tile)
  }
}
// d > 128: two shapes only (see knn_plan)
template <>
int dispatch_kp<128, 32>(const KnnPlan& p, const KnnBuffers& b, int64_t q_begin, hipStream_t s) {
  if (p.KP == 128) return launch_select<128, 32, 2, 128>(p, b, q_begin, s);
  return launch_select<128, 32, 1, 288>(p, b, q_begin, s);
}

template <int H, int TC_, int WPS, bool B3 = false>
static int launch_select_reg_mode(const KnnPlan& p, const KnnBuffers& b, int64_t q_begin, hipStream_t s) {
  using C = RegCfg<H, TC_, B3>;
  auto kern = knn_select_reg_kernel<H, TC_, WPS, false, B3>;
  SCAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
  const int n_tiles = (int)(p.n_pad / C::TC);
  const int grid = (int)(p.nq_pad / C::QB);
  IvfArgs none{};
  hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), C::LDS_BYTES, s, b.xp, n_tiles, p.n_pad, q_begin,
                     p.thr_rank, b.cand_idx, b.cand_tau, none);
  SCAMD_LAUNCH_CHECK();
  return SCAMD_OK;
}

template <int H>
static int launch_select_reg(const KnnPlan& p, const KnnBuffers& b, int64_t q_begin, hipStream_t s) {
  // 64-candidate tiles (30 KB of LDS per block) and <= 168 VGPRs: three 4-wave blocks share a CU and cover each
  // other's barrier / insertion stalls (measured 812 vs 831 ms at 1M against 128-candidate tiles, two blocks)
  static const bool big_tiles = [] {
    const char* e = getenv("SCAMD_KNN_BIG_TILES");
    return e && e[0] == '1';
  }();
  if constexpr (H == 25) {
    if (p.b3) return big_tiles ? launch_select_reg_mode<25, 128, 2, true>(p, b, q_begin, s)
                               : launch_select_reg_mode<25, 64, 2, true>(p, b, q_begin, s);
  }
  if (big_tiles) return launch_select_reg_mode<H, 128, 2>(p, b, q_begin, s);
  return launch_select_reg_mode<H, 64, 3>(p, b, q_begin, s);
}

static int dispatch_select(const KnnPlan& p, const KnnBuffers& b, int64_t q_begin, hipStream_t s) {
  if (p.reg) {
    switch (p.H) {
      case 8: return launch_select_reg<8>(p, b, q_begin, s);
      case 16: return launch_select_reg<16>(p, b, q_begin, s);
      case 25: return launch_select_reg<25>(p, b, q_begin, s);
      default: return launch_select_reg<32>(p, b, q_begin, s);
    }
  }
  switch (p.H) {
    case 8: return dispatch_kp<8, 128>(p, b, q_begin, s);
    case 16: return dispatch_kp<16, 128>(p, b, q_begin, s);
    case 25: return dispatch_kp<25, 128>(p, b, q_begin, s);
    case 32: return dispatch_kp<32, 128>(p, b, q_begin, s);
    case 64: return dispatch_kp<64, 64>(p, b, q_begin, s);
    default: return dispatch_kp<128, 32>(p, b, q_begin, s);
  }
}

// ---- cell-pruned search: quantiser, cell-sorted image, launch -------------------------------------------
template <int H, bool B3 = false>
static int run_ivf_select(const KnnPlan& p, const KnnBuffers& b, const float* x, int64_t n, int d, int64_t ld,
                          int64_t q_begin, int64_t n_query, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1,
                          int64_t* rows_out) {
  using C = RegCfg<H, 64, B3>;
  constexpr int MIN_CELL = 128;
  const int nc = p.n_cells;
  int* counts = b.cell_ints;
  int* qcounts = counts + nc;
  int* cell_map = qcounts + nc;
  int* row_off = cell_map + nc;
  int* row_cur = row_off + nc;
  int* slot_off = row_cur + nc;
  int* slot_cur = slot_off + nc;
  int* tile0 = slot_cur + nc;                     // cell_ints holds 8 * nc ints
  int* ntiles = reinterpret_cast<int*>(b.sums);  // the sums buffer is free once the quantiser is done
  // 1. quantiser: Lloyd on a strided sample
  auto assign = ivf_assign_kernel<H>;
  constexpr int DPA = (2 * H + 1 + 3) / 4 * 4;  // row length of the padded centroid table (ivf_assign_kernel)
  // d <= 50: the assignments run on the bf16 matrix cores (any assignment gives a correct search; SCAMD_KNN_ASSIGN_MFMA=0
  // keeps the float32 kernel, which at 32768 sampled rows x 512 centroids was 0.34 ms per Lloyd iteration -- a grid of
  // 128 workgroups each looping over all centroids)
  const int ncp = (nc + 31) / 32 * 32;
  const size_t lds_a = (size_t)ncp * CENT_DPL * sizeof(unsigned int);
  bool mfma_assign = false;
  {
    const char* e = getenv("SCAMD_KNN_ASSIGN_MFMA");
    mfma_assign = H == 25 && !(e && e[0] == '0') && lds_a <= 150 * 1024;
  }
  unsigned int* centb = reinterpret_cast<unsigned int*>(b.centp);  // [ncp][32] dwords <= the float table's nc x 136 floats
  if (mfma_assign)
    SCAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ivf_assign_mfma_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a));
  auto run_assign = [&](int64_t a_start, int64_t a_step, int64_t a_count, int accumulate, int64_t aq0, int64_t aq1,
                        int* a_qcounts, bool count_in_kernel = true) -> int {
    if (mfma_assign) {
      hipLaunchKernelGGL(ivf_centpad_bf16_kernel, dim3((unsigned)ceil_div((int64_t)ncp * 32, 256)), dim3(256), 0, s, b.cent, nc, ncp, d,
                         centb);
      SCAMD_LAUNCH_CHECK();
      hipLaunchKernelGGL(ivf_assign_mfma_kernel, dim3((unsigned)ceil_div(a_count, ASSIGN_ROWS)), dim3(256), lds_a, s, x, d, ld, a_start,
                         a_step, a_count, centb, ncp, nc, b.labels, count_in_kernel ? counts : (int*)nullptr, aq0, aq1, a_qcounts);
      SCAMD_LAUNCH_CHECK();
      if (accumulate) {
        hipLaunchKernelGGL(ivf_sums_kernel, dim3((unsigned)ceil_div(a_count * d, 256)), dim3(256), 0, s, x, d, ld, a_start, a_step,
                           a_count, b.labels, b.sums);
        SCAMD_LAUNCH_CHECK();
      }
    } else {
      hipLaunchKernelGGL(ivf_centpad_kernel, dim3((unsigned)ceil_div((int64_t)nc * DPA, 256)), dim3(256), 0, s, b.cent, nc, d,
                         DPA, b.centp);
      SCAMD_LAUNCH_CHECK();
      hipLaunchKernelGGL(assign, dim3((unsigned)ceil_div(a_count, 256)), dim3(256), 0, s, x, d, ld, a_start, a_step, a_count,
                         b.centp, nc, b.labels, accumulate, accumulate ? b.sums : (long long*)nullptr, counts, aq0, aq1,
                         a_qcounts);
      SCAMD_LAUNCH_CHECK();
    }
    return SCAMD_OK;
  };
  hipLaunchKernelGGL(ivf_init_kernel, dim3(nc), dim3(64), 0, s, x, n, d, ld, nc, b.cent);
  SCAMD_LAUNCH_CHECK();
  const int64_t n_sample = std::min<int64_t>(n, (int64_t)64 * nc);
  const int64_t step = n / n_sample;
  static const int lloyd_iters = [] {  // (A/B knob; 3 since round 1)
    const char* e = getenv("SCAMD_KNN_LLOYD_ITERS");
    return e ? std::max(0, std::min(16, atoi(e))) : 3;
  }();
  for (int it = 0; it < lloyd_iters; ++it) {
    SCAMD_HIP_CHECK(hipMemsetAsync(b.sums, 0, sizeof(long long) * nc * d, s));
    SCAMD_HIP_CHECK(hipMemsetAsync(counts, 0, sizeof(int) * nc, s));
    const int rca = run_assign(0, step, n_sample, 1, 0, 0, nullptr);
    if (rca != SCAMD_OK) return rca;
    hipLaunchKernelGGL(ivf_update_kernel, dim3((unsigned)ceil_div((int64_t)nc * d, 256)), dim3(256), 0, s, b.sums,
                       counts, nc, d, b.cent);
    SCAMD_LAUNCH_CHECK();
  }
  // 2. every row to its cell
  SCAMD_HIP_CHECK(hipMemsetAsync(counts, 0, sizeof(int) * 2 * nc, s));
  {
    const int rca = run_assign(0, 1, n, 0, q_begin, q_begin + n_query, qcounts, !mfma_assign);
    if (rca != SCAMD_OK) return rca;
    if (mfma_assign) {
      hipLaunchKernelGGL(ivf_count_kernel, dim3((unsigned)ceil_div(n, 4096)), dim3(1024), 0, s, b.labels, n, nc, q_begin, q_begin + n_query,
                         counts, qcounts);
      SCAMD_LAUNCH_CHECK();
    }
  }
  std::vector<int> h_cnt(2 * nc);
  std::vector<float> h_cent((size_t)nc * d);
  SCAMD_HIP_CHECK(hipMemcpyAsync(h_cnt.data(), counts, sizeof(int) * 2 * nc, hipMemcpyDeviceToHost, s));
  SCAMD_HIP_CHECK(hipMemcpyAsync(h_cent.data(), b.cent, sizeof(float) * nc * d, hipMemcpyDeviceToHost, s));
  SCAMD_HIP_CHECK(hipStreamSynchronize(s));
  // 3. host: fold small cells into their nearest big cell; padded layout of the image and of the query list
  std::vector<int> h_map(nc), mc(nc, 0), mq(nc, 0);
  int biggest = 0;
  for (int c = 1; c < nc; ++c)
    if (h_cnt[c] > h_cnt[biggest]) biggest = c;
  for (int c = 0; c < nc; ++c) {
    int tgt = c;
    if (h_cnt[c] < MIN_CELL) {
      tgt = biggest;
      double bestd = INFINITY;
      for (int e = 0; e < nc; ++e) {
        if (h_cnt[e] < MIN_CELL) continue;
        double dd = 0.0;
        for (int t = 0; t < d; ++t) {
          const double df = (double)h_cent[(size_t)c * d + t] - (double)h_cent[(size_t)e * d + t];
          dd += df * df;
        }
        if (dd < bestd) {
          bestd = dd;
          tgt = e;
        }
      }
    }
    h_map[c] = tgt;
    mc[tgt] += h_cnt[c];
    mq[tgt] += h_cnt[nc + c];
  }
  std::vector<int> h_row_off(nc), h_slot_off(nc), h_tile0(nc), h_ntiles(nc), h_block_cell, h_blk(2 * nc);
  int64_t rows = 0, slots = 0;
  for (int c = 0; c < nc; ++c) {
    h_row_off[c] = (int)rows;
    h_tile0[c] = (int)(rows / 64);
    h_ntiles[c] = (mc[c] + 63) / 64;
    rows += (int64_t)h_ntiles[c] * 64;
    h_slot_off[c] = (int)slots;
    const int nb = (mq[c] + 127) / 128;
    h_blk[c] = (int)h_block_cell.size();  // first block of the cell
    h_blk[nc + c] = nb;
    for (int t = 0; t < nb; ++t) h_block_cell.push_back(c);
    slots += (int64_t)nb * 128;
  }
  const int n_blocks = (int)h_block_cell.size();
  SCAMD_REQUIRE(rows <= p.n_img_max && slots <= p.n_slot_max, SCAMD_EWORKSPACE, "knn: cell layout exceeds its bound");
  // launch slots: block-id order = n_blocks; XCD-aware order (SCAMD_KNN_XCD_ORDER=0 disables it) = 8 queues of at most
  // ceil(n_blocks / 8) + (blocks of the largest cell) slots, if the table has room
  int n_launch = n_blocks, xcd_mode = 0;
  {
    // (the queues are built on the device from the device's work estimates; the host sizes the launch for queues of twice
    // the mean length plus the largest cell -- slots beyond a queue's end hold -1 and exit at once, a queue that would
    // not fit is cut off by `slot < n_slots` in ivf_block_order_kernel ... which must not happen: checked there)
    const char* e = getenv("SCAMD_KNN_XCD_ORDER");
    int maxb = 0;
    for (int c = 0; c < nc; ++c) maxb = std::max(maxb, h_blk[nc + c]);
    const int64_t want = (int64_t)8 * (2 * ((n_blocks + 7) / 8) + maxb);
    const int64_t cap = (int64_t)(p.n_slot_max / 128 + 1) * 2 + 64;
    if (!(e && e[0] == '0') && n_blocks >= 64 && want <= cap) {
      xcd_mode = 1;
      n_launch = (int)want;
    }
  }
  rows_out[0] = rows;
  rows_out[1] = slots;
  if (n_blocks == 0) return SCAMD_OK;
  SCAMD_HIP_CHECK(hipMemcpyAsync(cell_map, h_map.data(), sizeof(int) * nc, hipMemcpyHostToDevice, s));
  SCAMD_HIP_CHECK(hipMemcpyAsync(row_off, h_row_off.data(), sizeof(int) * nc, hipMemcpyHostToDevice, s));
  SCAMD_HIP_CHECK(hipMemcpyAsync(slot_off, h_slot_off.data(), sizeof(int) * nc, hipMemcpyHostToDevice, s));
  SCAMD_HIP_CHECK(hipMemcpyAsync(tile0, h_tile0.data(), sizeof(int) * nc, hipMemcpyHostToDevice, s));
  SCAMD_HIP_CHECK(hipMemcpyAsync(ntiles, h_ntiles.data(), sizeof(int) * nc, hipMemcpyHostToDevice, s));
  SCAMD_HIP_CHECK(hipMemcpyAsync(b.block_cell, h_block_cell.data(), sizeof(int) * n_blocks, hipMemcpyHostToDevice, s));
  SCAMD_HIP_CHECK(hipMemcpyAsync(b.cell_aux + nc, h_blk.data(), sizeof(int) * 2 * nc, hipMemcpyHostToDevice, s));
  SCAMD_HIP_CHECK(hipMemsetAsync(row_cur, 0, sizeof(int) * nc, s));
  SCAMD_HIP_CHECK(hipMemsetAsync(slot_cur, 0, sizeof(int) * nc, s));
  SCAMD_HIP_CHECK(hipMemsetAsync(b.perm, 0xff, sizeof(int) * rows, s));
  SCAMD_HIP_CHECK(hipMemsetAsync(b.qpos, 0xff, sizeof(int) * slots, s));
  SCAMD_HIP_CHECK(hipMemsetAsync(b.radius_bits, 0, sizeof(unsigned int) * nc, s));
  SCAMD_HIP_CHECK(hipMemsetAsync(b.counters + 2, 0, 16, s));
  SCAMD_HIP_CHECK(hipStreamSynchronize(s));  // the host vectors above must outlive their copies
  // 4. cell-sorted image
  hipLaunchKernelGGL(ivf_scatter_kernel, dim3((unsigned)ceil_div(n, 1024 * SCATTER_ROWS)), dim3(1024), 0, s, b.labels, n, cell_map, row_off,
                     row_cur, q_begin, q_begin + n_query, slot_off, slot_cur, b.perm, b.qpos, b.qrow, nc);
  SCAMD_LAUNCH_CHECK();
  {
    // (a group of 16 lanes per image row, 16 groups per workgroup; SCAMD_KNN_PACK_BLOCKS: A/B of the grid)
    static const int pack_blocks_env = [] {
      const char* e = getenv("SCAMD_KNN_PACK_BLOCKS");
      return e ? atoi(e) : 0;
    }();
    const int blocks = (int)std::min<int64_t>((rows + 15) / 16, pack_blocks_env > 0 ? pack_blocks_env : 256 * 16);
    hipLaunchKernelGGL(ivf_pack_image_kernel, dim3(blocks), dim3(256), 0, s, x, b.mu, d, ld, H, C::HP, C::DPL, rows, b.perm,
                       b.labels, cell_map, b.cent, b.xp, b.cmax, b.radius_bits, B3 ? 1 : 0);
    SCAMD_LAUNCH_CHECK();
  }
  // 5. sweep order of every cell (needs the radii the pack kernel just produced)
  {
    int npow = 1;
    while (npow < nc) npow <<= 1;
    const size_t olds = (size_t)npow * 8;
    SCAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ivf_cell_order_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)olds));
    hipLaunchKernelGGL(ivf_cell_order_kernel, dim3(nc), dim3(256), olds, s, b.cent, d,
                       reinterpret_cast<const float*>(b.radius_bits), ntiles, nc, b.cell_order, b.cell_lb2, b.cell_aux,
                       p.nprobe);
    SCAMD_LAUNCH_CHECK();
    // launch order: longest expected sweeps first
    hipLaunchKernelGGL(ivf_block_order_kernel, dim3(1), dim3(1024), 0, s, b.cell_aux, b.cell_aux + nc, b.cell_aux + 2 * nc,
                       nc, b.block_perm, n_launch, xcd_mode, b.counters + 6);
    SCAMD_LAUNCH_CHECK();
  }
  // 6. pruned sweep
  // register budget: cut for 3 resident blocks per CU by default (168 VGPRs; the H = 25 / 32 instantiations then spill
  // 96 / 236 bytes per lane to scratch, `-Rpass-analysis=kernel-resource-usage`); SCAMD_KNN_IVF_WPS=2 selects the
  // build cut for 2 blocks per CU (no spills) -- an A/B switch until both have been measured
  const char* wps_env = getenv("SCAMD_KNN_IVF_WPS");
  // (bf16 engine, round 4: with its tiles arriving by LDS-DMA the staging registers are gone -- 175 VGPRs uncut, 5 of them
  // spilled in the build cut for 3 blocks per CU (was 47), and 3 x 53 KB of LDS just fit the CU's 160: 13.31 ms against
  // 16.11 with 2 blocks on one box, profiles/r04s_knn_lds_dma_ring_ab.log -- a third block covers the other two's insertion
  // stalls.  SCAMD_KNN_IVF_WPS=2 selects the uncut build)
  auto kern = B3 ? ((wps_env && atoi(wps_env) == 2) ? knn_select_reg_kernel<H, 64, 2, true, B3> : knn_select_reg_kernel<H, 64, 3, true, B3>)
                 : ((wps_env && atoi(wps_env) == 2) ? knn_select_reg_kernel<H, 64, 2, true, false>
                                                     : knn_select_reg_kernel<H, 64, 3, true, false>);
  // COARSE first stage (bf16 engine; knn_select_reg_block): pays when a query meets so many candidates that few 32 x 32
  // sub-tiles hold one below its threshold -- the sweeps in which the cell bounds prune little.  Decided from the device's own
  // work estimates (tiles within a cell's radius, ivf_cell_order_kernel), weighted by the cells' query blocks: expected
  // candidates per query >= 5e5 (1M cells: the planted matrix ~3e4 -> plain kernel, the weak / structure-less ones 1e6 -> coarse,
  // 317 -> 260 ms; 10M planted cells ~2.5e5 -> plain: 795 ms against 835 with the coarse stage, profiles/r06y2_ab.log).
  // SCAMD_KNN_COARSE=0 / 1 forces the choice (A/B, tests).
  bool coarse = false;
  if constexpr (B3) {
    const char* ce = getenv("SCAMD_KNN_COARSE");
    if (ce && (ce[0] == '0' || ce[0] == '1')) {
      coarse = ce[0] == '1';
    } else {
      std::vector<int> h_work(nc);
      SCAMD_HIP_CHECK(hipMemcpyAsync(h_work.data(), b.cell_aux, sizeof(int) * nc, hipMemcpyDeviceToHost, s));
      SCAMD_HIP_CHECK(hipStreamSynchronize(s));
      double num = 0.0, den = 0.0;
      for (int c = 0; c < nc; ++c) {
        num += (double)h_blk[nc + c] * (double)h_work[c] * 64.0;
        den += (double)h_blk[nc + c];
      }
      coarse = den > 0.0 && num / den >= 5.0e5;
    }
    if (coarse) kern = knn_select_reg_kernel<H, 64, 3, true, B3, B3>;  // (COARSE = B3: the float32 engine has no such instantiation)
  }
  g_last_coarse = coarse ? 1 : 0;
  const size_t lds = C::LDS_BYTES + 64 + IVF_META_BYTES;
  SCAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds));
  IvfArgs iv;
  iv.qpos = b.qpos;
  iv.block_cell = b.block_cell;
  iv.cell_tile0 = tile0;
  iv.cell_ntiles = ntiles;
  iv.centers = b.cent;
  iv.radius = reinterpret_cast<const float*>(b.radius_bits);
  iv.order = b.cell_order;
  iv.order_lb2 = b.cell_lb2;
  iv.perm = b.perm;
  iv.cmax_bits = b.cmax;
  iv.pairs = reinterpret_cast<unsigned long long*>(b.counters + 2);
  iv.n_cells = nc;
  iv.dc = d;
  iv.d = d;
  iv.block_perm = b.block_perm;
  iv.qorder = b.qorder;
  {
    const char* e = getenv("SCAMD_KNN_PREPASS_TILES");
    // float32 engine, measured at 1M: 48 -> 29.3 ms, 24 / 12 -> 29.0, 6 -> 29.6, 2 -> 30.2; bf16 engine (the pre-pass
    // costs a quarter): 8 -> 17.1, 16 -> 16.8, 32 -> 16.15, 64 -> 16.1
    iv.prepass_tiles = e ? std::max(1, atoi(e)) : (B3 ? 32 : 16);
    const char* e2 = getenv("SCAMD_KNN_PREPASS_CELLS");
    iv.prepass_cells = e2 ? std::max(1, atoi(e2)) : 1;
    const char* e3 = getenv("SCAMD_KNN_PREPASS_MIN2");
    iv.prepass_min2 = (e3 && e3[0] == '0') ? 0 : 1;
  }
  {
    const char* e = getenv("SCAMD_KNN_DEBUG_NO_INSERT");
    iv.debug_no_insert = (e && e[0] == '1') ? 1 : 0;
  }
  {
    const char* e = getenv("SCAMD_KNN_CELL_PRELOAD");  // A/B switch: 0 = every cell's sweep requests its first tile itself
    iv.cell_preload = (e && e[0] == '0') ? 0 : 1;
  }
  iv.trace = nullptr;
  const char* trace_path = getenv("SCAMD_KNN_TRACE");  // debug: per-block timeline of the sweep, dumped to this file
  if (trace_path && trace_path[0]) {
    SCAMD_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&iv.trace), sizeof(unsigned long long) * 8 * n_blocks));
    SCAMD_HIP_CHECK(hipMemsetAsync(iv.trace, 0, sizeof(unsigned long long) * 8 * n_blocks, s));
  }
  // persistent launch (see knn_select_reg_kernel): 8 XCDs x 32 CUs x the resident blocks per CU; SCAMD_KNN_PERSISTENT=0 is the
  // launch of one workgroup per slot, = N sets the number of workgroups (rounded up to a multiple of 8)
  int n_groups = n_launch;
  iv.queue_ctr = nullptr;
  iv.n_slots = n_launch;
  {
    const char* e = getenv("SCAMD_KNN_PERSISTENT");
    const int want = e ? atoi(e) : 256 * ((wps_env && atoi(wps_env) == 2) ? 2 : 3);
    if (want > 0 && n_launch > want) {
      n_groups = (want + 7) / 8 * 8;
      iv.queue_ctr = b.counters + 8;
      SCAMD_HIP_CHECK(hipMemsetAsync(b.counters + 8, 0, sizeof(int) * 8, s));
    }
  }
  SCAMD_HIP_CHECK(hipEventRecord(ev0, s));
  hipLaunchKernelGGL(kern, dim3(n_groups), dim3(C::NT), lds, s, b.xp, (int)(rows / 64), rows, q_begin, p.thr_rank,
                     b.cand_idx, b.cand_tau, iv);
  SCAMD_LAUNCH_CHECK();
  SCAMD_HIP_CHECK(hipEventRecord(ev1, s));
  if (iv.trace) {
    std::vector<unsigned long long> h((size_t)8 * n_blocks);
    SCAMD_HIP_CHECK(hipMemcpyAsync(h.data(), iv.trace, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost, s));
    SCAMD_HIP_CHECK(hipStreamSynchronize(s));
    SCAMD_HIP_CHECK(hipFree(iv.trace));
    if (FILE* f = fopen(trace_path, "wb")) {
      fwrite(h.data(), sizeof(unsigned long long), h.size(), f);
      fwrite(h_block_cell.data(), sizeof(int), h_block_cell.size(), f);
      fclose(f);
    }
  }
  return SCAMD_OK;
}

static int dispatch_ivf(const KnnPlan& p, const KnnBuffers& b, const float* x, int64_t n, int d, int64_t ld,
                        int64_t q_begin, int64_t n_query, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1,
                        int64_t* rows_out) {
  switch (p.H) {
    case 8: return run_ivf_select<8>(p, b, x, n, d, ld, q_begin, n_query, s, ev0, ev1, rows_out);
    case 16: return run_ivf_select<16>(p, b, x, n, d, ld, q_begin, n_query, s, ev0, ev1, rows_out);
    case 25:
      if (p.b3) return run_ivf_select<25, true>(p, b, x, n, d, ld, q_begin, n_query, s, ev0, ev1, rows_out);
      return run_ivf_select<25>(p, b, x, n, d, ld, q_begin, n_query, s, ev0, ev1, rows_out);
    default: return run_ivf_select<32>(p, b, x, n, d, ld, q_begin, n_query, s, ev0, ev1, rows_out);
  }
}

// Second tier of the bf16 engine (pruned mode).  Its certificate is ~4x looser than the float32 engine's relative to
// ||q|| ||c||; how many queries it rejects depends on the data (508 of 1M planted cells at a margin of 6 ranks, 76543 of
// 10M x 4k at a margin of 10: 7 s of float64 cell scans).  The rejected queries alone -- grouped by cell into query
// blocks of their own -- are swept again by the FLOAT32 engine over a float32 image of the same cell-sorted rows (same
// cell tables, same sweep orders), re-ranked and certified with the float32 bound; what fails that too goes to the
// float64 scan as before.  Cost: one image (0.7 ms per million rows) + the float32 sweep of < 1 % of the queries.
#define T2_DBG(line)                                                                 \
  do {                                                                               \
    if (getenv("SCAMD_KNN_T2_DEBUG")) {                                              \
      hipError_t e_ = hipStreamSynchronize(s);                                       \
      fprintf(stderr, "[knn tier2] line %d done (%s)\n", (int)(line), hipGetErrorString(e_)); \
      fflush(stderr);                                                                \
    }                                                                                \
  } while (0)
static int run_ivf_tier2(const KnnPlan& p, const KnnBuffers& b, const float* x, int64_t n, int d, int64_t ld,
                         int64_t q_begin, int64_t n_query, int k, double cert_scale, int n_flag, int64_t rows,
                         int32_t* out_idx, double* out_dist, hipStream_t s, int* n_flag2_host) {
  using C = RegCfg<25, 64, false>;
  const int nc = p.n_cells;
  int* cell_map = b.cell_ints + 2 * nc;
  int* tile0 = b.cell_ints + 7 * nc;
  int* ntiles = reinterpret_cast<int*>(b.sums);
  int* cnt2 = b.t2_ints;
  int* slot_off2 = b.t2_ints + nc;
  int* ctr2 = b.t2_ints + 2 * nc;  // [0] still uncertified, [2..3] / [4..5] pair counters of the second sweep (not reported)
  *n_flag2_host = 0;
  SCAMD_HIP_CHECK(hipMemsetAsync(cnt2, 0, sizeof(int) * nc, s));
  SCAMD_HIP_CHECK(hipMemsetAsync(ctr2, 0, sizeof(int) * 8, s));
  hipLaunchKernelGGL(ivf_t2_count_kernel, dim3((unsigned)ceil_div(n_flag, 256)), dim3(256), 0, s, b.flag_list, n_flag, q_begin,
                     b.labels, cell_map, cnt2, b.t2_cell, b.t2_pos);
  SCAMD_LAUNCH_CHECK();
  T2_DBG(__LINE__);
  std::vector<int> h_cnt(nc), h_slot_off(nc), h_block_cell;
  SCAMD_READBACK_NOW(h_cnt.data(), cnt2, sizeof(int) * nc, s);
  int64_t slots = 0;
  for (int c = 0; c < nc; ++c) {
    h_slot_off[c] = (int)slots;
    const int nb = (h_cnt[c] + 127) / 128;
    for (int t = 0; t < nb; ++t) h_block_cell.push_back(c);
    slots += (int64_t)nb * 128;
  }
  const int n_blocks = (int)h_block_cell.size();
  if (n_blocks == 0) return SCAMD_OK;
  SCAMD_REQUIRE(slots <= p.n_slot_max, SCAMD_EWORKSPACE, "knn: second-tier query layout exceeds its bound");
  SCAMD_HIP_CHECK(hipMemcpyAsync(slot_off2, h_slot_off.data(), sizeof(int) * nc, hipMemcpyHostToDevice, s));
  SCAMD_HIP_CHECK(hipMemcpyAsync(b.block_cell, h_block_cell.data(), sizeof(int) * n_blocks, hipMemcpyHostToDevice, s));
  SCAMD_HIP_CHECK(hipMemsetAsync(b.qpos, 0xff, sizeof(int) * slots, s));
  SCAMD_HIP_CHECK(hipStreamSynchronize(s));  // the host vectors must outlive their copies
  hipLaunchKernelGGL(ivf_t2_fill_kernel, dim3((unsigned)ceil_div(n_flag, 256)), dim3(256), 0, s, b.flag_list, n_flag, b.t2_cell,
                     b.t2_pos, slot_off2, b.qrow, b.qpos);
  SCAMD_LAUNCH_CHECK();
  T2_DBG(__LINE__);
  hipLaunchKernelGGL(knn_iota_kernel, dim3((unsigned)ceil_div(n_blocks, 256)), dim3(256), 0, s, b.block_perm, n_blocks);
  SCAMD_LAUNCH_CHECK();
  T2_DBG(__LINE__);
  {
    const int blocks = (int)std::min<int64_t>((rows + 3) / 4, 256 * 16);
    hipLaunchKernelGGL(ivf_pack_image_kernel, dim3(blocks), dim3(256), 0, s, x, b.mu, d, ld, 25, C::HP, C::DPL, rows, b.perm,
                       b.labels, cell_map, b.cent, b.xp2, b.cmax, b.radius_bits, 0);
    SCAMD_LAUNCH_CHECK();
  T2_DBG(__LINE__);
  }
  auto kern = knn_select_reg_kernel<25, 64, 3, true, false>;
  const size_t lds = C::LDS_BYTES + 64 + IVF_META_BYTES;
  SCAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds));
  IvfArgs iv;
  iv.qpos = b.qpos;
  iv.block_cell = b.block_cell;
  iv.cell_tile0 = tile0;
  iv.cell_ntiles = ntiles;
  iv.centers = b.cent;
  iv.radius = reinterpret_cast<const float*>(b.radius_bits);
  iv.order = b.cell_order;
  iv.order_lb2 = b.cell_lb2;
  iv.perm = b.perm;
  iv.cmax_bits = b.cmax;
  iv.pairs = reinterpret_cast<unsigned long long*>(ctr2 + 2);
  iv.n_cells = nc;
  iv.dc = d;
  iv.d = d;
  iv.block_perm = b.block_perm;
  iv.qorder = nullptr;
  iv.prepass_tiles = 16;
  iv.prepass_cells = 1;
  iv.prepass_min2 = 1;
  iv.debug_no_insert = 0;
  iv.cell_preload = 1;
  iv.trace = nullptr;
  iv.queue_ctr = nullptr;  // (a few hundred queries: one workgroup per block)
  iv.n_slots = n_blocks;
  const int thr_rank = std::min(32, std::max(1, k + 6));
  hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(C::NT), lds, s, b.xp2, (int)(rows / 64), rows, q_begin, thr_rank, b.cand_idx,
                     b.cand_tau, iv);
  SCAMD_LAUNCH_CHECK();
  T2_DBG(__LINE__);
  {
    const size_t rlds = rerank_rows_lds_bytes(d, thr_rank - 1);
    SCAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(knn_rerank_rows_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)rlds));
    hipLaunchKernelGGL(knn_rerank_rows_kernel, dim3((unsigned)ceil_div(n_flag, 8)), dim3(256), rlds, s, x, b.mu, n, d, ld,
                       q_begin, n_query, k, std::max(1, thr_rank - 1), b.cand_idx, b.cand_tau, b.cmax, cert_scale, CERT_K_F32,
                       0.0, out_idx, out_dist, b.kth_d2, b.flag_list2, ctr2, (const int*)b.flag_list, (int64_t)n_flag);
  }
  SCAMD_LAUNCH_CHECK();
  T2_DBG(__LINE__);
  SCAMD_READBACK_NOW(n_flag2_host, ctr2, sizeof(int), s);
  return SCAMD_OK;
}

}  // namespace scamd

using namespace scamd;

extern "C" float scamd_knn_last_select_ms(void) { return g_last_select_ms; }
extern "C" double scamd_knn_last_select_pairs(void) { return g_last_select_pairs; }
extern "C" double scamd_knn_last_select_prepass_pairs(void) { return g_last_select_prepass_pairs; }
extern "C" int scamd_knn_last_select_engine(void) { return g_last_select_engine; }
extern "C" int scamd_knn_last_second_tier_queries(void) { return g_last_second_tier; }
extern "C" int scamd_knn_last_nprobe(void) { return g_last_nprobe; }
extern "C" int scamd_knn_last_coarse(void) { return g_last_coarse; }

extern "C" size_t scamd_knn_workspace_bytes(int64_t n, int d, int64_t n_query, int k) {
  // one figure for the exact and the approximate entry point: the approximate plan goes through the cell tables at
  // sizes where the exact one sweeps by brute force (4096 <= n < 65536)
  size_t need = 0;
  for (int nprobe = 0; nprobe <= 1; ++nprobe) {
    KnnPlan p;
    if (!knn_plan(n, d, n_query, k, &p, nprobe)) return 0;
    Workspace ws(nullptr, 0);
    KnnBuffers b;
    knn_carve(ws, p, n_query, &b);
    need = std::max(need, ws.used());
  }
  return need;
}

static int knn_l2_impl(const float* x, int64_t n, int d, int64_t ld_x, int64_t q_begin, int64_t n_query, int k,
                       int32_t* out_idx, double* out_dist, double cert_scale, int64_t* n_fallback_host, void* workspace,
                       size_t workspace_bytes, scamd_stream_t stream, int nprobe);

extern "C" int scamd_knn_l2_f32(const float* x, int64_t n, int d, int64_t ld_x, int64_t q_begin,
                                int64_t n_query, int k, int32_t* out_idx, double* out_dist,
                                double cert_scale, int64_t* n_fallback_host, void* workspace,
                                size_t workspace_bytes, scamd_stream_t stream) {
  return knn_l2_impl(x, n, d, ld_x, q_begin, n_query, k, out_idx, out_dist, cert_scale, n_fallback_host, workspace,
                     workspace_bytes, stream, 0);
}

// Approximate variant (BASELINE configs[4]: "IVF-tiled approximate kNN"; the reference's own default above 8192 cells is
// approximate too, src/scanpy/neighbors/__init__.py:734-739, 769-781): every query sees the rows of the nprobe cells
// nearest to its own cell (centroid distance) of the same k-means quantiser the exact search prunes with.  Inside those
// cells the search is the exact one (same kernels, same float64 re-rank and certificate), so the lists are the true
// nearest neighbours AMONG THE PROBED ROWS: recall < 1 comes from unprobed cells only.  nprobe <= 0 or >= the cell count:
// the exact search.  Shapes the register-list kernel does not take (k > 24, d > 64) and n < 4096 are answered exactly.
extern "C" int scamd_knn_l2_ivf_f32(const float* x, int64_t n, int d, int64_t ld_x, int64_t q_begin,
                                    int64_t n_query, int k, int nprobe, int32_t* out_idx, double* out_dist,
                                    int64_t* n_fallback_host, void* workspace, size_t workspace_bytes,
                                    scamd_stream_t stream) {
  return knn_l2_impl(x, n, d, ld_x, q_begin, n_query, k, out_idx, out_dist, 1.0, n_fallback_host, workspace,
                     workspace_bytes, stream, nprobe > 0 ? nprobe : 0);
}

static int knn_l2_impl(const float* x, int64_t n, int d, int64_t ld_x, int64_t q_begin, int64_t n_query, int k,
                       int32_t* out_idx, double* out_dist, double cert_scale, int64_t* n_fallback_host, void* workspace,
                       size_t workspace_bytes, scamd_stream_t stream, int nprobe) {
  SCAMD_REQUIRE(x && out_idx && out_dist, SCAMD_EINVAL, "knn: null pointer");
  SCAMD_REQUIRE(n >= 1 && d >= 1 && ld_x >= d, SCAMD_EINVAL, "knn: bad shape n=%lld d=%d ld=%lld",
                (long long)n, d, (long long)ld_x);
  SCAMD_REQUIRE(n < (int64_t)1 << 31, SCAMD_EUNSUPPORTED, "knn: n=%lld exceeds int32 row ids", (long long)n);
  SCAMD_REQUIRE(q_begin >= 0 && n_query >= 0 && q_begin + n_query <= n, SCAMD_EINVAL,
                "knn: query range [%lld, %lld) outside [0, %lld)", (long long)q_begin,
                (long long)(q_begin + n_query), (long long)n);
  SCAMD_REQUIRE(k >= 1, SCAMD_EINVAL, "knn: k=%d", k);
  KnnPlan p;
  SCAMD_REQUIRE(knn_plan(n, d, n_query, k, &p, nprobe), SCAMD_EUNSUPPORTED,
                "knn: unsupported d=%d (max 256) or k=%d (max 256)", d, k);
  g_last_nprobe = p.nprobe;
  if (n_fallback_host) *n_fallback_host = 0;
  if (n_query == 0) return SCAMD_OK;
  Workspace ws(workspace, workspace_bytes);
  KnnBuffers b;
  knn_carve(ws, p, n_query, &b);
  SCAMD_REQUIRE(workspace && ws.ok, SCAMD_EWORKSPACE, "knn: workspace %zu < required %zu",
                workspace_bytes, ws.used());
  hipStream_t s = stream;

  SCAMD_HIP_CHECK(hipMemsetAsync(b.cmax, 0, 16, s));
  SCAMD_HIP_CHECK(hipMemsetAsync(b.counters, 0, 32, s));
  hipLaunchKernelGGL(knn_colsum_kernel, dim3(MEAN_BLOCKS), dim3(1024), 0, s, x, n, d, ld_x, b.mean_partial);
  SCAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(knn_colmean_kernel, dim3(1), dim3(KNN_MAX_D), 0, s, b.mean_partial, n, d, b.mu);
  SCAMD_LAUNCH_CHECK();
  if (!p.ivf) {
    int blocks = (int)std::min<int64_t>((p.n_pad + 3) / 4, 256 * 16);
    if (p.reg) {
      const int HP = (p.H + 1 + 3) / 4 * 4;
      hipLaunchKernelGGL(knn_pack_image_kernel, dim3(blocks), dim3(256), 0, s, x, b.mu, n, d, ld_x, p.H, HP,
                         p.row_dwords, p.n_pad, b.xp, b.cmax, p.b3 ? 1 : 0);
    } else {
      hipLaunchKernelGGL(knn_pack_kernel, dim3(blocks), dim3(256), 0, s, x, b.mu, n, d, ld_x, 2 * p.H,
                         p.n_pad, b.xp, b.cn, b.cmax);
    }
    SCAMD_LAUNCH_CHECK();
  }
  int64_t ivf_layout[2] = {0, 0};  // image rows, query slots of the pruned sweep
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  SCAMD_HIP_CHECK(hipEventCreate(&ev0));
  SCAMD_HIP_CHECK(hipEventCreate(&ev1));
  int rc;
  if (p.ivf) {
    rc = dispatch_ivf(p, b, x, n, d, ld_x, q_begin, n_query, s, ev0, ev1, ivf_layout);
  } else {
    SCAMD_HIP_CHECK(hipEventRecord(ev0, s));
    rc = dispatch_select(p, b, q_begin, s);
    if (rc == SCAMD_OK && hipEventRecord(ev1, s) != hipSuccess) rc = SCAMD_EHIP;
  }
  if (rc != SCAMD_OK) {
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);
    return rc;
  }
  if (p.reg && p.KP == 32 && d <= 64) {
    // register-list kernels: sorted lists, row-wise re-rank of the entries below the threshold, in slot order when pruned
    const int n_rank = std::max(1, p.thr_rank - 1);
    const int64_t n_list = p.ivf ? ivf_layout[1] : n_query;
    const size_t lds = rerank_rows_lds_bytes(d, n_rank);
    SCAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(knn_rerank_rows_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(knn_rerank_rows_kernel, dim3((unsigned)ceil_div(n_list, 8)), dim3(256), lds, s, x, b.mu, n, d, ld_x,
                       q_begin, n_query, k, n_rank, b.cand_idx, b.cand_tau, b.cmax, cert_scale,
                       p.b3 ? CERT_K_B3 : CERT_K_F32, p.b3 ? CERT_K2_B3 : 0.0, out_idx, out_dist, b.kth_d2, b.flag_list,
                       b.counters, p.ivf ? (const int*)b.qorder : (const int*)nullptr, n_list);
    SCAMD_LAUNCH_CHECK();
  } else {
    int blocks = (int)((n_query + 3) / 4);
#define RERANK(KP_)                                                                              \
  hipLaunchKernelGGL(knn_rerank_kernel<KP_>, dim3(blocks), dim3(256), 0, s, x, b.mu, n, d, ld_x, q_begin, \
                     n_query, k, b.cand_idx, b.cand_tau, b.cmax, cert_scale, p.b3 ? CERT_K_B3 : CERT_K_F32 + (p.H > 64 ? 2.0 * (p.H - 64) : 0.0), p.b3 ? CERT_K2_B3 : 0.0, out_idx, out_dist,  \
                     b.kth_d2, b.flag_list, b.counters, (const int*)nullptr, 0)
    if (p.KP == 32) RERANK(32);
    else if (p.KP == 64) RERANK(64);
    else if (p.KP == 128) RERANK(128);
    else RERANK(288);
#undef RERANK
    SCAMD_LAUNCH_CHECK();
  }
  int h_counters[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  SCAMD_READBACK_NOW(h_counters, b.counters, 32, s);
  {
    float ms = -1.f;
    if (hipEventElapsedTime(&ms, ev0, ev1) != hipSuccess) ms = -1.f;
    g_last_select_ms = ms;
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);
    unsigned long long pairs = 0;
    memcpy(&pairs, &h_counters[2], 8);
    g_last_select_pairs = p.ivf ? (double)pairs : (double)n_query * (double)n;
    unsigned long long pre = 0;
    memcpy(&pre, &h_counters[4], 8);
    g_last_select_prepass_pairs = p.ivf ? (double)pre : 0.0;
    g_last_select_engine = p.b3 ? 1 : 0;
  }
  SCAMD_REQUIRE(h_counters[6] == 0, SCAMD_EINTERNAL, "knn: a query block of the pruned sweep got no launch slot (XCD-aware order)");
  int n_flag = h_counters[0];
  const int* flag_list = b.flag_list;
  g_last_second_tier = 0;
  if (p.ivf && p.b3) {
    // second tier: the float32 engine on what the bf16 engine's certificate rejected (SCAMD_KNN_TIER2_MIN, default 256
    // queries: below that the float64 scan of a few queries is cheaper than a second image)
    const char* e = getenv("SCAMD_KNN_TIER2_MIN");
    const int t2_min = e ? atoi(e) : 256;
    if (n_flag > t2_min) {
      int n_flag2 = 0;
      rc = run_ivf_tier2(p, b, x, n, d, ld_x, q_begin, n_query, k, cert_scale, n_flag, ivf_layout[0], out_idx, out_dist, s,
                         &n_flag2);
      if (rc != SCAMD_OK) return rc;
      g_last_second_tier = n_flag;
      n_flag = n_flag2;
      flag_list = b.flag_list2;
    }
  }
  if (n_fallback_host) *n_fallback_host = n_flag;
  // float64 scan of what is left; a query whose table overflowed comes back with a tighter bound (knn_fallback_rank_kernel)
  int n_todo = n_flag;
  const int* todo = flag_list;
  for (int round = 0; n_todo > 0; ++round) {
    SCAMD_REQUIRE(round < 8, SCAMD_EUNSUPPORTED, "knn: %d queries have more than %d rows tied within their k-th distance",
                  n_todo, FALLBACK_CAP);
    int* retry = b.fb_retry[round & 1];
    SCAMD_HIP_CHECK(hipMemsetAsync(b.counters + 1, 0, sizeof(int), s));
    for (int begin = 0; begin < n_todo; begin += FALLBACK_CHUNK) {
      int count = std::min(FALLBACK_CHUNK, n_todo - begin);
      SCAMD_HIP_CHECK(hipMemsetAsync(b.fb_counts, 0, sizeof(int) * count, s));
      if (p.ivf) {
        // cell tables of run_ivf_select (same carving): tile0 = cell_ints + 7 nc, ntiles = the recycled sums buffer
        const int nc = p.n_cells;
        // (few queries: one workgroup per cell and query -- a launch lasts as long as its busiest workgroup)
        hipLaunchKernelGGL(knn_fallback_scan_cells_kernel, dim3(count <= 128 ? nc : std::min(nc, 64), count), dim3(256), 0, s, x, d, ld_x, q_begin,
                           todo, begin, b.kth_d2, b.scratch_d, b.scratch_i, b.fb_counts, b.cent,
                           reinterpret_cast<const float*>(b.radius_bits), b.cell_ints + 7 * nc,
                           reinterpret_cast<const int*>(b.sums), b.perm, nc);
      } else {
        const int chunks = (int)std::max<int64_t>(1, std::min<int64_t>(FALLBACK_ROW_CHUNKS, n / 2048));
        hipLaunchKernelGGL(knn_fallback_scan_kernel, dim3(chunks, count), dim3(256), 0, s, x, n, d, ld_x, q_begin,
                           todo, begin, b.kth_d2, b.scratch_d, b.scratch_i, b.fb_counts);
      }
      SCAMD_LAUNCH_CHECK();
      hipLaunchKernelGGL(knn_fallback_rank_kernel, dim3(count), dim3(256), 0, s, k, todo, begin, b.scratch_d,
                         b.scratch_i, b.fb_counts, out_idx, out_dist, b.kth_d2, retry, b.counters + 1);
      SCAMD_LAUNCH_CHECK();
    }
    SCAMD_READBACK_NOW(h_counters, b.counters, 16, s);
    n_todo = h_counters[1];
    todo = retry;
  }
  return SCAMD_OK;
}

// ------------------------------------------------------------------------------------------------
// Test entry for the certificate of the bf16 engine: the RAW scores ||c||^2 - 2 q.c of a block of (query, candidate)
// pairs, produced by the select kernel's own image packing (knn_colsum / knn_colmean / knn_pack_image_kernel) and its own
// operand construction and MFMA chain (b3_query_operand, b3_chain) -- one wave per 32 x 32 sub-tile, no lists, threshold 0.
// tests/test_gpu_knn_certificate.py compares them with float64 and reports max |error| / bound; the bound's factors are
// exported by scamd_knn_cert_factors so that the test cannot drift from the kernel.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void knn_debug_b3_scores_kernel(const float* __restrict__ xp, int64_t q0, int nq,
                                                                  int64_t c0, int nc, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int ctiles = nc / 32;
  if (wid >= (int64_t)(nq / 32) * ctiles) return;
  const int qt = (int)(wid / ctiles), ct = (int)(wid % ctiles);
  i32x4 qh[4], ql[4];
  b3_query_operand(xp, q0 + (int64_t)qt * 32 + l31, half, qh, ql);
  BFragBf16 b;
  const i32x4* p = reinterpret_cast<const i32x4*>(xp + (c0 + (int64_t)ct * 32 + l31) * B3_DPL);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    b.h[s] = p[2 * s + half];
    b.l[s] = p[8 + 2 * s + half];
  }
  const f32x16 acc = b3_chain(qh, ql, b);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * half;  // query of accumulator register r in this half (select kernel's map)
    out[((int64_t)qt * 32 + i) * nc + (int64_t)ct * 32 + l31] = acc[r];
  }
}

extern "C" void scamd_knn_cert_factors(int engine, double* cert_k, double* cert_k2, double* key_slack) {
  if (cert_k) *cert_k = engine == 1 ? CERT_K_B3 : CERT_K_F32;
  if (cert_k2) *cert_k2 = engine == 1 ? CERT_K2_B3 : 0.0;
  if (key_slack) *key_slack = 128.0;
}

extern "C" size_t scamd_knn_debug_b3_scores_workspace_bytes(int64_t n) {
  Workspace ws(nullptr, 0);
  ws.take<float>((size_t)n * B3_DPL);
  ws.take<float>(KNN_MAX_D);
  ws.take<double>((size_t)MEAN_BLOCKS * KNN_MAX_D);
  ws.take<unsigned int>(4);
  return ws.used();
}

extern "C" int scamd_knn_debug_b3_scores_f32(const float* x, int64_t n, int d, int64_t ld_x, int64_t q0, int nq,
                                             int64_t c0, int nc, float* out_scores, float* out_mu, float* out_cmax,
                                             void* workspace, size_t workspace_bytes, scamd_stream_t stream) {
  SCAMD_REQUIRE(x && out_scores, SCAMD_EINVAL, "knn debug scores: null pointer");
  SCAMD_REQUIRE(n >= 1 && d >= 1 && d <= 50 && ld_x >= d, SCAMD_EUNSUPPORTED, "knn debug scores: the bf16 engine takes d <= 50 (d=%d)", d);
  SCAMD_REQUIRE(nq > 0 && nc > 0 && nq % 32 == 0 && nc % 32 == 0 && q0 >= 0 && c0 >= 0 && q0 + nq <= n && c0 + nc <= n,
                SCAMD_EINVAL, "knn debug scores: query / candidate blocks must be multiples of 32 inside [0, n)");
  Workspace ws(workspace, workspace_bytes);
  float* xp = ws.take<float>((size_t)n * B3_DPL);
  float* mu = ws.take<float>(KNN_MAX_D);
  double* partial = ws.take<double>((size_t)MEAN_BLOCKS * KNN_MAX_D);
  unsigned int* cmax = ws.take<unsigned int>(4);
  SCAMD_REQUIRE(workspace && ws.ok, SCAMD_EWORKSPACE, "knn debug scores: workspace %zu < required %zu", workspace_bytes, ws.used());
  hipStream_t s = stream;
  SCAMD_HIP_CHECK(hipMemsetAsync(cmax, 0, 16, s));
  hipLaunchKernelGGL(knn_colsum_kernel, dim3(MEAN_BLOCKS), dim3(1024), 0, s, x, n, d, ld_x, partial);
  SCAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(knn_colmean_kernel, dim3(1), dim3(KNN_MAX_D), 0, s, partial, n, d, mu);
  SCAMD_LAUNCH_CHECK();
  const int blocks = (int)std::min<int64_t>((n + 3) / 4, 256 * 16);
  hipLaunchKernelGGL(knn_pack_image_kernel, dim3(blocks), dim3(256), 0, s, x, mu, n, d, ld_x, 25, 28, B3_DPL, n, xp, cmax, 1);
  SCAMD_LAUNCH_CHECK();
  const int64_t waves = (int64_t)(nq / 32) * (nc / 32);
  hipLaunchKernelGGL(knn_debug_b3_scores_kernel, dim3((unsigned int)((waves + 3) / 4)), dim3(256), 0, s, xp, q0, nq, c0, nc, out_scores);
  SCAMD_LAUNCH_CHECK();
  if (out_mu) SCAMD_HIP_CHECK(hipMemcpyAsync(out_mu, mu, sizeof(float) * 128, hipMemcpyDeviceToDevice, s));
  if (out_cmax) SCAMD_HIP_CHECK(hipMemcpyAsync(out_cmax, cmax, sizeof(float), hipMemcpyDeviceToDevice, s));
  return SCAMD_OK;
}

// ------------------------------------------------------------------------------------------------
// MFMA layout self test: D = A(32x2) * B(2x32) + C with asymmetric integer-valued operands.
// ------------------------------------------------------------------------------------------------
namespace scamd {
__global__ void mfma_layout_kernel(int* mismatches) {
  const int lane = threadIdx.x & 63;
  const int i = lane & 31, kk = lane >> 5;
  // A[i][k] = i + 100*k + 1 ; B[k][j] = 3*j + 7*k + 2 ; C[i][j] = 1000*i + j
  float a = (float)(i + 100 * kk + 1);
  float bval = (float)(3 * i + 7 * kk + 2);  // lane holds B[k=kk][j=i]
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
    acc[r] = (float)(1000 * row + i);
  }
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bval, acc, 0, 0, 0);
  int bad = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
    int col = i;
    float expect = (float)(1000 * row + col);
    for (int k2 = 0; k2 < 2; ++k2) expect += (float)(row + 100 * k2 + 1) * (float)(3 * col + 7 * k2 + 2);
    if (acc[r] != expect) ++bad;
  }
  if (bad) atomicAdd(mismatches, bad);
}
}  // namespace scamd

extern "C" int scamd_selftest_mfma_layout(scamd_stream_t stream) {
  int* d_bad = nullptr;
  SCAMD_HIP_CHECK(hipMalloc(&d_bad, sizeof(int)));
  SCAMD_HIP_CHECK(hipMemsetAsync(d_bad, 0, sizeof(int), stream));
  hipLaunchKernelGGL(scamd::mfma_layout_kernel, dim3(1), dim3(64), 0, stream, d_bad);
  int h_bad = -1;
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(&h_bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  (void)hipFree(d_bad);
  if (e != hipSuccess) {
    scamd::set_error("mfma selftest: %s", hipGetErrorString(e));
    return SCAMD_EHIP;
  }
  if (h_bad != 0) {
    scamd::set_error("mfma selftest: %d mismatching accumulator entries", h_bad);
    return SCAMD_EUNSUPPORTED;
  }
  return SCAMD_OK;
}
