// Leiden community detection on gfx950: wave-per-vertex label moves over the CSR neighbour graph.
//
// Replaces leidenalg.find_partition(RBConfigurationVertexPartition) / igraph community_leiden
// (objective 'modularity') as called at src/scanpy/tools/_leiden.py:167-196 on the graph built by
// src/scanpy/_utils/__init__.py:278-304.  Same three phases as Traag et al. 2019,
// re-designed for a GPU:
//   1. local moving   -- SWEEPS of K class sub-rounds (K = 8 on large levels): the active vertices of a sweep are split
//                        into K hash classes; sub-round r decides (one wave / half / quarter wave per vertex: weight
//                        towards each neighbouring community in an LDS hash, best move from a snapshot of the community
//                        totals) and applies the moves of class r only, so class r + 1 sees them -- a deterministic
//                        stand-in for the sequential queue: every sub-round is a synchronous step on a snapshot, the
//                        classes are re-drawn every sweep, only vertices next to a change stay active.
//   2. refinement     -- ONE sweep of K class sub-rounds over the well-connected singletons of each community
//                        (leidenalg visits every vertex exactly once, in random order): the singletons of class r merge
//                        into a well-connected sub-community drawn with probability ~ exp(gain / beta); targets are
//                        grown sub-communities and singletons of OTHER classes, so simultaneous merges cannot chain.
//   3. aggregation    -- members of every coarse node gathered contiguously, one wave / workgroup per coarse node
//                        combines their rows in an LDS hash.
// ALL weight sums are 64-bit fixed point (weight * 2^32): integer addition is associative, so atomic
// accumulation order cannot change any result -- labels are bitwise reproducible run to run.
// Memory-latency / gather bound: per round ~E*(4+8+4) B of edge data + n*24 B of vertex data.
#include "common.h"
#include "scan.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <ctime>

namespace scamd {

// per-call statistics of the last scamd_leiden_csr_f32 on this thread (scamd_leiden_last_stats): [0] outer iterations,
// [1] kernel launches, [2] blocking host round trips, [3] full sweeps / [4] rounds / [5] moves of the final polish,
// [6] 1 if the polish was skipped because the last iteration had already proven node optimality, [7] levels of iteration 0,
// [8] local-moving sweeps of the levels that run as separate kernels (all iterations), [9] their algorithmic traffic in MB:
//     active rows x (12 B per entry + 16 B per vertex), SURVEY.md 8(d)'s per-sweep figure restricted to the rows a sweep visits,
// [10] communities the polish split off because a departure had cut them in two, [11] 1 if the iteration cap ended the run,
// [12] 1 if a polish pass stopped at MAX_POLISH_ROUNDS (node optimality then NOT proven), [13] the iteration cap in force
// register budgets of the decision kernels (second launch bound = waves per SIMD the compiler must fit; A/B builds: -D...)
#ifndef SCAMD_LD_PROPOSE_WAVES
#define SCAMD_LD_PROPOSE_WAVES 0
#endif
#ifndef SCAMD_LD_MOVE_WAVES
#define SCAMD_LD_MOVE_WAVES 0
#endif
#if SCAMD_LD_PROPOSE_WAVES > 0
#define SCAMD_LD_PROPOSE_LB __launch_bounds__(256, SCAMD_LD_PROPOSE_WAVES)
#else
#define SCAMD_LD_PROPOSE_LB __launch_bounds__(256)
#endif
#if SCAMD_LD_MOVE_WAVES > 0
#define SCAMD_LD_MOVE_LB __launch_bounds__(256, SCAMD_LD_MOVE_WAVES)
#else
#define SCAMD_LD_MOVE_LB __launch_bounds__(256)
#endif
// table slots of a vertex of the sixteen-lanes-per-vertex kernels (the kNN graph itself); rows longer than 3/4 of them go to
// the wave-per-vertex tier
#ifndef SCAMD_LD_G16_SLOTS
#define SCAMD_LD_G16_SLOTS 128
#endif
constexpr int G16_SLOTS = SCAMD_LD_G16_SLOTS;
constexpr int G16_MAX = G16_SLOTS * 3 / 4;
constexpr int LD_NSTATS = 16;
static thread_local int g_ld_stats[LD_NSTATS] = {0};
static thread_local double g_ld_sweep_bytes = 0.0;  // -> stats[9] (MB)
#undef SCAMD_LAUNCH_CHECK
#define SCAMD_LAUNCH_CHECK()                \
  do {                                      \
    ++g_ld_stats[1];                        \
    SCAMD_HIP_CHECK(hipGetLastError());     \
  } while (0)
// host round trips: the values the host decides on are fetched into a pinned page (common.h: HostReadback -- a copy into
// pageable memory is a blocking round trip of its own) and handed out by ONE synchronisation
#define LD_FETCH(dst, src, bytes, stream)                                          \
  do {                                                                             \
    const int rc_fetch_ = host_readback().fetch((dst), (src), (bytes), (stream)); \
    if (rc_fetch_ != SCAMD_OK) return rc_fetch_;                                   \
  } while (0)
#define LD_SYNC(stream)                                       \
  do {                                                        \
    ++g_ld_stats[2];                                          \
    const int rc_sync_ = host_readback().sync(stream);        \
    if (rc_sync_ != SCAMD_OK) return rc_sync_;                \
  } while (0)

constexpr double WSCALE = 4294967296.0;  // 2^32
constexpr int MAX_LM_SWEEPS = 96;
constexpr int LM_DIR_AFTER = 32;  // sweeps after which the direction rule (termination guarantee) is switched on
constexpr int MAX_CLASSES = 32;   // most class sub-rounds per sweep (local moving) / per refinement
constexpr int DEF_CLASSES = 8;    // default
constexpr int CTR_STRIDE = 16;    // ints per sub-round counter block
// counter area of a sweep / a refinement: MAX_CLASSES class-list lengths, then one CTR_STRIDE block per sub-round
constexpr int CTR_AREA = MAX_CLASSES + CTR_STRIDE * MAX_CLASSES;
constexpr int MAX_LEVELS = 64;
// Outer iterations of an `n_iterations < 0` run ("until an iteration does not improve", src/scanpy/tools/_leiden.py:65, 166).
// On graphs without clear communities that takes dozens of iterations in the sequential algorithm as well: oracle/leiden.c
// needs 44 on the 1M-cell `weak` graph of bench.py (gains of 1e-7 .. 1e-5 Q per iteration from the 6th on; round-6 probe,
// profiles/r06a_oracle_iters_weak.log), 18 on a 100k sample.  The loop is bounded (SCAMD_LEIDEN_ITER_CAP overrides); a run the
// bound ends sets stats[11] and `tl.leiden` warns.
constexpr int MAX_OUTER_ITERS = 32;

__host__ __device__ __forceinline__ unsigned int hash32(unsigned int x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ unsigned int prio(int c, unsigned int seed) { return hash32((unsigned int)c ^ seed); }

__device__ __forceinline__ long long readlane_i64(long long v, int l) {
  int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffll), l);
  int hi = __builtin_amdgcn_readlane((int)(v >> 32), l);
  return ((long long)hi << 32) | (unsigned int)lo;
}

// ---- small utility kernels -----------------------------------------------------------------------
__global__ void ld_quantize_kernel(const float* __restrict__ w, int64_t nnz, long long* __restrict__ wq) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < nnz) {
    double v = (double)w[e];
    wq[e] = (v > 0.0) ? (long long)llrint(v * WSCALE) : 0ll;
  }
}

// one wave per row: k[v] = sum of row
__global__ void ld_strength_kernel(const int64_t* __restrict__ indptr, const long long* __restrict__ wq, int n,
                                   long long* __restrict__ k) {
  const int lane = threadIdx.x & 63;
  const int v = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (v >= n) return;
  long long s = 0;
  for (int64_t e = indptr[v] + lane; e < indptr[v + 1]; e += 64) s += wq[e];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) k[v] = s;
}

// total += sum a[0..n)  (grid-stride, one atomic per wave; integer => order independent)
__global__ __launch_bounds__(256) void ld_sum_kernel(const long long* __restrict__ a, int n,
                                                     unsigned long long* __restrict__ total) {
  long long s = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) s += a[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0 && s != 0) atomicAdd(total, (unsigned long long)s);
}

// row-length statistics of a level: out[0] = longest row, out[1 .. 3] = rows longer than 96 / 192 / 384 entries (the
// bounds of the quarter- / half- / full-wave tables: levels without such rows skip the overflow / hub launches, the
// others size those grids by the counts)
__global__ __launch_bounds__(1024) void ld_degstats_kernel(const int64_t* __restrict__ indptr, int n, int* __restrict__ out) {
  // (one set of atomics per 1024-row block: a million rows through four same-address atomics per WAVE took 0.3 ms)
  __shared__ int sh[4];
  if (threadIdx.x < 4) sh[threadIdx.x] = 0;
  __syncthreads();
  int v = blockIdx.x * blockDim.x + threadIdx.x;
  int d = (v < n) ? (int)(indptr[v + 1] - indptr[v]) : 0;
  const unsigned long long m1 = __ballot(d > G16_MAX), m2 = __ballot(d > 192), m3 = __ballot(d > 384);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) d = max(d, __shfl_xor(d, o));
  if ((threadIdx.x & 63) == 0 && d > 0) {
    atomicMax(&sh[0], d);
    if (m1) atomicAdd(&sh[1], __popcll(m1));
    if (m2) atomicAdd(&sh[2], __popcll(m2));
    if (m3) atomicAdd(&sh[3], __popcll(m3));
  }
  __syncthreads();
  if (threadIdx.x == 0 && sh[0] > 0) {
    atomicMax(out, sh[0]);
    if (sh[1]) atomicAdd(out + 1, sh[1]);
    if (sh[2]) atomicAdd(out + 2, sh[2]);
    if (sh[3]) atomicAdd(out + 3, sh[3]);
  }
}

__global__ void ld_iota_kernel(int* __restrict__ a, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = i;
}

// CPM with the caller's vertex weights (igraph `node_weights`): fixed point with 16 fractional bits; a negative or non-finite
// weight raises the flag (the host turns it into SCAMD_EINVAL)
constexpr double NODE_WEIGHT_SCALE = 65536.0;
__global__ void ld_nodeweight_quantize_kernel(const float* __restrict__ w, int n, long long* __restrict__ k, int* __restrict__ err) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  const float x = w[v];
  if (!(x >= 0.f) || !(x <= 1.0e6f)) {  // (the totals are int64 sums of 2^16 x: 1e6 x 1e8 vertices still fit)
    atomicOr(err, 1);
    k[v] = 0ll;
    return;
  }
  k[v] = llrint((double)x * NODE_WEIGHT_SCALE);
}
__global__ void ld_fill_i64_kernel(long long* __restrict__ a, int n, long long v) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = v;
}
// coarse vertex weights of CPM: kout[cid[v]] += k[v]  (kout zeroed by the caller; integer: order free)
__global__ void ld_agg_nodeweight_kernel(int n, const int* __restrict__ cid, const long long* __restrict__ k,
                                         long long* __restrict__ kout) {
  int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v < n) atomicAdd(reinterpret_cast<unsigned long long*>(&kout[cid[v]]), (unsigned long long)k[v]);
}

__global__ void ld_fill_u8_kernel(unsigned char* __restrict__ a, int n, unsigned char v) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = v;
}

// Clears of the orchestration (round 6).  Every hipMemsetAsync / device-to-device hipMemcpyAsync is a blit kernel of the
// runtime (`__amd_rocclr_fillBufferAligned` / `__amd_rocclr_copyBuffer`: 365 launches and 4.35 ms per pass of the path in the
// round-5 profile).  What could not be folded into a kernel that is launched anyway (per-vertex initialisations, the counter
// blocks of the next sweep) goes through ONE launch of this kernel per phase: up to six regions, each filled with its own
// 32-bit pattern (0, 0x7f7f7f7f = the `memset 0x7f` sentinel of the atomicMin targets, 0xffffffff).
struct FillArgs {
  static constexpr int MAX_REGIONS = 6;
  unsigned int* p[MAX_REGIONS];
  unsigned long long words[MAX_REGIONS];
  unsigned int pat[MAX_REGIONS];
  int n;
};
__global__ __launch_bounds__(256) void ld_fill_kernel(FillArgs a) {
  const unsigned long long t = (unsigned long long)blockIdx.x * 256ull + threadIdx.x, stride = (unsigned long long)gridDim.x * 256ull;
  for (int r = 0; r < a.n; ++r) {
    const unsigned long long words = a.words[r];
    const unsigned int pat = a.pat[r];
    unsigned int* __restrict__ p = a.p[r];
    // (regions are carved on 256-byte boundaries: whole 16-byte stores, then the tail)
    if ((reinterpret_cast<uintptr_t>(p) & 15u) == 0) {
      const unsigned long long quads = words >> 2;
      uint4* __restrict__ q = reinterpret_cast<uint4*>(p);
      for (unsigned long long i = t; i < quads; i += stride) q[i] = make_uint4(pat, pat, pat, pat);
      for (unsigned long long i = (quads << 2) + t; i < words; i += stride) p[i] = pat;
    } else {
      for (unsigned long long i = t; i < words; i += stride) p[i] = pat;
    }
  }
}
// start of an outer iteration: comm = the partition it starts from, node_of = identity
__global__ void ld_iter_init_kernel(int n, const int* __restrict__ memb, int* __restrict__ comm, int* __restrict__ node_of) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v < n) {
    comm[v] = memb[v];
    node_of[v] = v;
  }
}
__global__ void ld_copy_i32_kernel(int n, const int* __restrict__ src, int* __restrict__ dst) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v < n) dst[v] = src[v];
}

// ---- block-local pre-aggregation of per-community reductions ---------------------------------------
// Late in the optimisation a million vertices reduce into a few dozen communities: a global atomic per vertex
// serialises on those few L2 lines (~12 ns each).  Every 1024-thread block first combines its vertices in an LDS
// hash table (open addressing, <= 8 probes; a vertex that finds no slot goes to global memory directly) and then
// issues one global atomic per distinct community it saw.  All reductions are integer (+, min): order free.
constexpr int BH_SLOTS = 2048;
constexpr int REDUCE_GRID = 256;  // workgroups of the per-community reductions (one per CU)
constexpr int BH_EMPTY = -1;

__device__ __forceinline__ int bh_find_slot(int* keys, int c) {
  unsigned int slot = hash32((unsigned int)c) & (BH_SLOTS - 1);
  for (int probe = 0; probe < 8; ++probe) {
    const int prev = atomicCAS(&keys[slot], BH_EMPTY, c);
    if (prev == BH_EMPTY || prev == c) return (int)slot;
    slot = (slot + 1) & (BH_SLOTS - 1);
  }
  return -1;
}

// Ktot[c] = sum k[v], csize[c] = #members   (both zeroed by the caller)
__global__ __launch_bounds__(1024) void ld_totals_kernel(const int* __restrict__ comm, const long long* __restrict__ k, int n,
                                                         unsigned long long* __restrict__ Ktot, int* __restrict__ csize,
                                                         unsigned long long* __restrict__ zero_me) {
  __shared__ int keys[BH_SLOTS];
  if (zero_me && blockIdx.x == 0 && threadIdx.x == 0) *zero_me = 0ull;  // (the internal-weight accumulator of `quality`)
  __shared__ unsigned long long ksum[BH_SLOTS];
  __shared__ int cnt[BH_SLOTS];
  for (int i = threadIdx.x; i < BH_SLOTS; i += 1024) {
    keys[i] = BH_EMPTY;
    ksum[i] = 0ull;
    cnt[i] = 0;
  }
  __syncthreads();
  // (at most REDUCE_GRID workgroups walk the vertices: with a few dozen communities left, every workgroup ends with one
  // global atomic per community on the same few lines -- 977 workgroups at 1M vertices took 85 us, round 5 profile)
  for (int v = blockIdx.x * 1024 + threadIdx.x; v < n; v += gridDim.x * 1024) {
    const int c = comm[v];
    const int slot = bh_find_slot(keys, c);
    if (slot >= 0) {
      atomicAdd(&ksum[slot], (unsigned long long)k[v]);
      atomicAdd(&cnt[slot], 1);
    } else {
      atomicAdd(&Ktot[c], (unsigned long long)k[v]);
      atomicAdd(&csize[c], 1);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < BH_SLOTS; i += 1024) {
    const int c = keys[i];
    if (c != BH_EMPTY) {
      atomicAdd(&Ktot[c], ksum[i]);
      atomicAdd(&csize[c], cnt[i]);
    }
  }
}

// ---- wave-wide argmax helper ---------------------------------------------------------------------
struct Cand {
  double val;
  int c;
  unsigned int pr;
};
__device__ __forceinline__ bool cand_better(const Cand& a, const Cand& b) {  // is a better than b
  if (a.c < 0) return false;
  if (b.c < 0) return true;
  if (a.val != b.val) return a.val > b.val;
  if (a.pr != b.pr) return a.pr < b.pr;
  return a.c < b.c;
}
__device__ __forceinline__ Cand wave_best(Cand x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    Cand y;
    y.val = __shfl_xor(x.val, o);
    y.c = __shfl_xor(x.c, o);
    y.pr = (unsigned int)__shfl_xor((int)x.pr, o);
    if (cand_better(y, x)) x = y;
  }
  return x;
}

// ---- per-wave LDS hash: weight towards each distinct neighbouring group --------------------------------
// A vertex with <= WH_MAX_DEG neighbours accumulates sum_w per distinct group id (community / refined community)
// in a 128-slot open-addressing table private to its wave: O(deg) LDS atomics instead of the O(deg^2) all-pairs
// readlane compare (which remains the path for the few high-degree coarse vertices).  Integer adds: the sums do
// not depend on the order of insertion; the table is read back two slots per lane.
constexpr int WH_SLOTS = 512;    // per wave (6 KB); a vertex uses the first 128 / 256 / 512 of them
constexpr int WH_MAX_DEG = 384;
constexpr int WH_EMPTY = -1;

struct WaveHash {
  int* keys;                 // [WH_SLOTS]
  unsigned long long* vals;  // [WH_SLOTS]
  int nslots;                // power of two >= 1.33 * deg
  __device__ __forceinline__ void size_for(int deg) { nslots = deg <= 96 ? 128 : (deg <= 192 ? 256 : 512); }
  __device__ __forceinline__ void clear(int lane) {
    for (int i = lane; i < nslots; i += 64) {
      keys[i] = WH_EMPTY;
      vals[i] = 0ull;
    }
  }
  __device__ __forceinline__ void add(int c, long long w) {
    unsigned int slot = hash32((unsigned int)c) & (nslots - 1);
    for (;;) {
      const int prev = atomicCAS(&keys[slot], WH_EMPTY, c);
      if (prev == WH_EMPTY || prev == c) break;
      slot = (slot + 1) & (nslots - 1);
    }
    atomicAdd(&vals[slot], (unsigned long long)w);
  }
  __device__ __forceinline__ int key(int slot) const { return __hip_atomic_load(&keys[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
  __device__ __forceinline__ long long val(int slot) const {
    return (long long)__hip_atomic_load(&vals[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
};

// ---- high-degree vertices: one workgroup per vertex, 8192-slot LDS table ---------------------------------
// kNN graphs have hubs (in-degree in the thousands) and aggregated graphs are dense: the wave-per-vertex kernels
// hand every vertex with more than WH_MAX_DEG neighbours to these block-per-vertex kernels (hub list filled
// through counters[4]) instead of letting ONE wave grind through an O(deg^2) compare while the grid waits.
constexpr int BHUB_SLOTS = 8192;
constexpr int BHUB_MAX_DEG = 6000;  // longest row one pass of the table takes (load <= 0.73)
// Longer rows (the coarse levels of a weakly clustered graph are nearly dense: 20k vertices with up to 18k
// neighbours each) are swept in P = ceil(deg / BHUB_PASS_DEG) passes: pass p only sees the neighbour groups of hash
// class p, so a pass holds ~deg / P <= 3072 distinct keys.  (Round 1 sent such rows through an all-pairs
// readlane compare, O(deg^2): 1.6 s for ONE local-moving round of a 26k-vertex level, 43 s per Leiden call on a
// structure-less 1M-cell graph.)
constexpr int BHUB_PASS_DEG = 3072;
__device__ __forceinline__ int bhub_passes(int deg) { return deg <= BHUB_MAX_DEG ? 1 : (deg + BHUB_PASS_DEG - 1) / BHUB_PASS_DEG; }
__device__ __forceinline__ int bhub_class(int c, int n_pass) {
  return (int)((hash32((unsigned int)c * 0x9E3779B1u + 0x7F4A7C15u) >> 7) % (unsigned int)n_pass);
}
// The pass count of a long row is set by its LENGTH (the only bound on its distinct neighbour groups known in advance), but
// on the coarse levels where such rows live the groups are few: a row of 19 000 entries pointing at a few hundred
// communities was swept seven times (round 5 profile of the structure-less graph: ld_move_hub_kernel 20 % of the Leiden time, a
// launch lasting as long as its longest row).  So a multi-pass row first tries ONE pass into the whole table with bounded
// probing; only if some key finds no slot within HUB_TRY_PROBES steps is the table cleared and the row swept class by class.
constexpr int HUB_TRY_PROBES = 64;

struct BlockHash {
  int* keys;
  unsigned long long* vals;
  int nslots;  // power of two >= 2 * deg (256 .. BHUB_SLOTS): clearing and scanning cost O(deg), not O(table)
  __device__ __forceinline__ void size_for(int deg) {
    int s = 256;
    while (s < 2 * deg && s < BHUB_SLOTS) s <<= 1;
    nslots = s;
  }
  __device__ __forceinline__ void clear() {
    for (int i = threadIdx.x; i < nslots; i += blockDim.x) {
      keys[i] = WH_EMPTY;
      vals[i] = 0ull;
    }
  }
  __device__ __forceinline__ void add(int c, long long w) {
    unsigned int slot = hash32((unsigned int)c) & (nslots - 1);
    for (;;) {
      const int prev = atomicCAS(&keys[slot], WH_EMPTY, c);
      if (prev == WH_EMPTY || prev == c) break;
      slot = (slot + 1) & (nslots - 1);
    }
    atomicAdd(&vals[slot], (unsigned long long)w);
  }
  // optimistic single pass over a long row (see hub_try_single_pass): gives up after max_probes occupied slots
  __device__ __forceinline__ bool add_limited(int c, long long w, int max_probes) {
    unsigned int slot = hash32((unsigned int)c) & (nslots - 1);
    for (int probes = 0; probes < max_probes; ++probes) {
      const int prev = atomicCAS(&keys[slot], WH_EMPTY, c);
      if (prev == WH_EMPTY || prev == c) {
        atomicAdd(&vals[slot], (unsigned long long)w);
        return true;
      }
      slot = (slot + 1) & (nslots - 1);
    }
    return false;
  }
  // multi-pass rows: the number of distinct keys of a pass is bounded only in expectation -> bounded probing;
  // false = table full (the caller raises the error flag, the host turns it into SCAMD_EINTERNAL)
  __device__ __forceinline__ bool add_bounded(int c, long long w) {
    unsigned int slot = hash32((unsigned int)c) & (nslots - 1);
    for (int probes = 0; probes < nslots; ++probes) {
      const int prev = atomicCAS(&keys[slot], WH_EMPTY, c);
      if (prev == WH_EMPTY || prev == c) {
        atomicAdd(&vals[slot], (unsigned long long)w);
        return true;
      }
      slot = (slot + 1) & (nslots - 1);
    }
    return false;
  }
};

// block-wide argmax of candidates (+ max of w_own); result valid in thread 0
__device__ __forceinline__ Cand block_best(Cand x, long long& w_own, Cand* sh_c, long long* sh_w) {
  x = wave_best(x);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) w_own = max(w_own, __shfl_xor(w_own, o));
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sh_c[wv] = x;
    sh_w[wv] = w_own;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) {
      if (cand_better(sh_c[i], x)) x = sh_c[i];
      w_own = max(w_own, sh_w[i]);
    }
  }
  return x;
}

// ---- phase 1: local moving -------------------------------------------------------------------------
// Round = decide (one wave per ACTIVE vertex, reads the state only) -> apply (one thread per active vertex,
// mutates the state, flags the vertices to revisit) -> compact (flags -> next active list + counters).  No
// kernel appends to a list through a single global counter: at 1M vertices that serialised on one L2 atomic
// (~12 ns each, 15 ms for the first round alone).

// G lanes per ACTIVE vertex (list[0..n_act)): decision[w] = community to move to, -1 = stay,
// -2 = wants to move but the direction rule forbids it this round (stays active).
//   G = 64: one wave per vertex, table of up to 512 slots (deg <= 384); hubs go to hub_list (counters[4]).
//   G = 16: FOUR vertices per wave, 128 slots each (deg <= 96) -- a kNN graph's rows are ~2k long, and the kernel
//           is bound by its chain of dependent gathers (list -> indptr -> indices -> comm -> Ktot), so four
//           vertices per wave put four times as many gathers in flight per wave slot.  Longer rows go to
//           ovf_list (counters[5]) and are decided by the G = 64 instantiation in indirect mode
//           (sub_list / sub_count = positions in `list`, grid-strided).
struct MoveArgs {
  int n_act;
  const int* list;
  const int* sub_list;
  const int* sub_count;
  const int64_t* indptr;
  const int* indices;
  const long long* wq;
  const long long* k;
  const int* comm;
  const unsigned long long* Ktot;
  const int* csize;
  double g;  // gamma / 2m
  int round;
  unsigned int seed;
  int* decision;
  int* ovf_list;
  int* hub_list;
  int* counters;
};
// (the body takes its workgroup number and the number of workgroups as arguments: ld_requeue_move_kernel runs it on the
// tail of a grid whose head re-queues the previous sub-round's movers)
template <int G>
__device__ __forceinline__ void ld_move_body(int bid, int nblk, const MoveArgs& ma) {
  const int n_act = ma.n_act;
  const int* __restrict__ list = ma.list;
  const int* __restrict__ sub_list = ma.sub_list;
  const int* __restrict__ sub_count = ma.sub_count;
  const int64_t* __restrict__ indptr = ma.indptr;
  const int* __restrict__ indices = ma.indices;
  const long long* __restrict__ wq = ma.wq;
  const long long* __restrict__ k = ma.k;
  const int* __restrict__ comm = ma.comm;
  const unsigned long long* __restrict__ Ktot = ma.Ktot;
  const int* __restrict__ csize = ma.csize;
  const double g = ma.g;
  const int round = ma.round;
  const unsigned int seed = ma.seed;
  int* __restrict__ decision = ma.decision;
  int* __restrict__ ovf_list = ma.ovf_list;
  int* __restrict__ hub_list = ma.hub_list;
  int* __restrict__ counters = ma.counters;
  constexpr int GROUPS = 256 / G;            // vertices per workgroup
  constexpr int GSLOTS = G == 16 ? G16_SLOTS : WH_SLOTS * G / 64;  // table slots per vertex
  constexpr int GMAX = GSLOTS * 3 / 4;       // longest row the table takes
  __shared__ int hkeys[GROUPS][GSLOTS];
  __shared__ unsigned long long hvals[GROUPS][GSLOTS];
  const int sub = threadIdx.x % G;
  const int grp = threadIdx.x / G;
  const int n_items = sub_list ? *sub_count : n_act;
  for (int item = bid * GROUPS + grp; item < n_items; item += nblk * GROUPS) {
    // The decision of one vertex is a chain of dependent gathers (list -> indptr -> indices -> comm -> Ktot) and the
    // kernel is bound by its length: everything that depends on the same address is requested together, before the
    // first branch that needs any of it (written the obvious way the compiler keeps seven round trips in series:
    // list, indptr, [degree branch] comm[v] / k[v], Ktot[a], indices, comm[u], Ktot[c]; now five).
    const int w = sub_list ? sub_list[item] : item;
    const int v = list[w];
    const int64_t beg = indptr[v];
    const int64_t end = indptr[v + 1];
    const int a = comm[v];
    const long long kq = k[v];
    const int deg = (int)(end - beg);
    // the row's first 2 G entries (rows of the kNN graph itself: all of them)
    int u_pre[2];
    long long w_pre[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int e = sub + t * G;
      u_pre[t] = e < deg ? indices[beg + e] : v;
      w_pre[t] = e < deg ? wq[beg + e] : 0ll;
    }
    const unsigned long long Ka = Ktot[a];
    int c_pre[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) c_pre[t] = comm[u_pre[t]];
    if (G < 64) {
      if (deg > GMAX) {  // decided by the wave-per-vertex instantiation
        if (sub == 0) ovf_list[atomicAdd(&counters[5], 1)] = w;
        continue;
      }
    } else if (deg > WH_MAX_DEG) {  // hub: decided by ld_move_hub_kernel (any length: multi-pass table)
      if (sub == 0) hub_list[atomicAdd(&counters[4], 1)] = w;
      continue;
    }
    const double kv = (double)kq;
    const double Ka_wo = (double)(long long)(Ka - (unsigned long long)kq);  // own community without v
    Cand best;
    best.val = 0.0;
    best.c = -1;
    best.pr = 0;
    long long w_own = 0;
    if (deg <= GMAX) {
      int* keys = hkeys[grp];
      unsigned long long* vals = hvals[grp];
      int nslots = 64;  // smallest power of two with load <= 3/4 (deg <= GMAX guarantees nslots <= GSLOTS)
      while (nslots * 3 / 4 < deg) nslots <<= 1;
      for (int i = sub; i < nslots; i += G) {
        keys[i] = WH_EMPTY;
        vals[i] = 0ull;
      }
      auto insert = [&](int c, long long wt) {
        unsigned int slot = hash32((unsigned int)c) & (nslots - 1);
        for (;;) {
          const int prev = atomicCAS(&keys[slot], WH_EMPTY, c);
          if (prev == WH_EMPTY || prev == c) break;
          slot = (slot + 1) & (nslots - 1);
        }
        atomicAdd(&vals[slot], (unsigned long long)wt);
      };
#pragma unroll
      for (int t = 0; t < 2; ++t)
        if (sub + t * G < deg && u_pre[t] != v) insert(c_pre[t], w_pre[t]);
      for (int e = sub + 2 * G; e < deg; e += G) {
        const int u = indices[beg + e];
        if (u != v) insert(comm[u], wq[beg + e]);
      }
      // candidates: the community totals of all occupied slots of this lane are requested before the first is used
      constexpr int MAXSL = GSLOTS / G;
      // (no per-lane branch around the gather and no constant merged into its result: either makes the compiler wait
      // for the load where it is issued; empty slots read Ktot[a], the wave skips slot ranges none of its tables has)
      int cs[MAXSL];
      unsigned long long kt[MAXSL];
#pragma unroll
      for (int t = 0; t < MAXSL; ++t) cs[t] = WH_EMPTY;
#pragma unroll
      for (int t = 0; t < MAXSL; ++t) {
        const int sl = sub + t * G;
        if (__ballot(sl < nslots)) {
          cs[t] = sl < nslots ? __hip_atomic_load(&keys[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : WH_EMPTY;
          kt[t] = Ktot[cs[t] != WH_EMPTY ? cs[t] : a];
        }
      }
#pragma unroll
      for (int t = 0; t < MAXSL; ++t) {
        const int c = cs[t];
        if (c != WH_EMPTY) {
          const long long sum =
              (long long)__hip_atomic_load(&vals[sub + t * G], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (c == a) {
            w_own = sum;
          } else {
            Cand x;
            x.val = (double)sum - g * kv * (double)(long long)kt[t];
            x.c = c;
            x.pr = prio(c, seed);
            if (cand_better(x, best)) best = x;
          }
        }
      }
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) {
      Cand y;
      y.val = __shfl_xor(best.val, o);
      y.c = __shfl_xor(best.c, o);
      y.pr = (unsigned int)__shfl_xor((int)best.pr, o);
      if (cand_better(y, best)) best = y;
      w_own = max(w_own, __shfl_xor(w_own, o));
    }
    const double stay = (double)w_own - g * kv * Ka_wo;
    int target = a;
    bool wants = false, allowed = true;
    if (best.c >= 0 && best.val > stay) {
      wants = true;
      target = best.c;
      const unsigned int pa = prio(a, seed), pb = best.pr;
      // direction rule (round >= 0 only: the termination guarantee of the late sweeps, see local_moving): moves towards
      // lower / higher priority communities alternate, so that two communities cannot keep swapping members
      if (round >= 0)
        allowed = (round & 1) ? (pb > pa || (pb == pa && target > a)) : (pb < pa || (pb == pa && target < a));
    } else if (stay < 0.0 && Ka_wo > 0.0 && csize[v] == 0) {
      // leaving for an empty community (id = own vertex id, free at the snapshot) beats staying
      wants = true;
      target = v;
    }
    if (sub == 0) decision[w] = (wants && allowed) ? target : (wants ? -2 : -1);
  }
}
template <int G>
__global__ SCAMD_LD_MOVE_LB void ld_move_kernel(
    int n_act, const int* __restrict__ list, const int* __restrict__ sub_list, const int* __restrict__ sub_count,
    const int64_t* __restrict__ indptr, const int* __restrict__ indices, const long long* __restrict__ wq,
    const long long* __restrict__ k, const int* __restrict__ comm, const unsigned long long* __restrict__ Ktot,
    const int* __restrict__ csize, double g /* gamma / 2m */, int round, unsigned int seed,
    int* __restrict__ decision, int* __restrict__ ovf_list, int* __restrict__ hub_list, int* __restrict__ counters) {
  const MoveArgs ma{n_act, list, sub_list, sub_count, indptr, indices, wq, k, comm, Ktot, csize, g, round, seed,
                    decision, ovf_list, hub_list, counters};
  ld_move_body<G>((int)blockIdx.x, (int)gridDim.x, ma);
}

// Hub vertices of the active list (positions in hub_list[0 .. counters[4])): one workgroup each.
// HUB_THREADS = 1024 (round 3; 256 before): a launch lasts as long as its longest row -- up to 20k entries on the coarse
// levels of unstructured graphs, swept in 7 hash-class passes -- and the row is walked blockDim entries at a time.
constexpr int HUB_THREADS = 1024;
__global__ __launch_bounds__(HUB_THREADS) void ld_move_hub_kernel(
    const int* __restrict__ hub_list, int* __restrict__ counters, const int* __restrict__ list,
    const int64_t* __restrict__ indptr, const int* __restrict__ indices, const long long* __restrict__ wq,
    const long long* __restrict__ k, const int* __restrict__ comm, const unsigned long long* __restrict__ Ktot,
    const int* __restrict__ csize, double g, int round, unsigned int seed, int* __restrict__ decision,
    int* __restrict__ err, int try_probes) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long hub_smem[];
  BlockHash bh{reinterpret_cast<int*>(hub_smem + BHUB_SLOTS), hub_smem, BHUB_SLOTS};
  __shared__ Cand sh_c[HUB_THREADS / 64];
  __shared__ long long sh_w[HUB_THREADS / 64];
  __shared__ int sh_fail;
  const int n_hub = counters[4];
  for (int i = blockIdx.x; i < n_hub; i += gridDim.x) {
    const int w = hub_list[i];
    const int v = list[w];
    const int a = comm[v];
    const double kv = (double)k[v];
    const int64_t beg = indptr[v];
    const int deg = (int)(indptr[v + 1] - beg);
    const double Ka_wo = (double)(long long)(Ktot[a] - (unsigned long long)k[v]);
    const int n_pass = bhub_passes(deg);
    bh.size_for(n_pass > 1 ? BHUB_SLOTS : deg);
    Cand best;
    best.val = 0.0;
    best.c = -1;
    best.pr = 0;
    long long w_own = 0;
    bool filled = false;  // the table already holds the whole row (a successful optimistic pass)
    if (n_pass > 1 && try_probes > 0) {
      if (threadIdx.x == 0) sh_fail = 0;
      bh.clear();
      __syncthreads();
      for (int e = threadIdx.x; e < deg; e += blockDim.x) {
        if (__hip_atomic_load(&sh_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
        const int u = indices[beg + e];
        if (u == v) continue;
        if (!bh.add_limited(comm[u], wq[beg + e], try_probes)) __hip_atomic_store(&sh_fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      __syncthreads();
      filled = sh_fail == 0;
      __syncthreads();  // (sh_fail is rewritten for the next row)
    }
    const int n_sweep = filled ? 1 : n_pass;
    for (int pass = 0; pass < n_sweep; ++pass) {
      if (!filled) {
        bh.clear();
        __syncthreads();
        for (int e = threadIdx.x; e < deg; e += blockDim.x) {
          const int u = indices[beg + e];
          if (u == v) continue;
          const int c = comm[u];
          if (n_pass == 1) bh.add(c, wq[beg + e]);
          else if (bhub_class(c, n_pass) == pass && !bh.add_bounded(c, wq[beg + e])) *err = 1;
        }
        __syncthreads();
      }
      for (int sl = threadIdx.x; sl < bh.nslots; sl += blockDim.x) {
        const int c = bh.keys[sl];
        if (c != WH_EMPTY) {
          const long long sum = (long long)bh.vals[sl];
          if (c == a) {
            w_own = sum;
          } else {
            Cand x;
            x.val = (double)sum - g * kv * (double)(long long)Ktot[c];
            x.c = c;
            x.pr = prio(c, seed);
            if (cand_better(x, best)) best = x;
          }
        }
      }
      __syncthreads();  // the scan of this pass is done before the next pass clears the table
    }
    best = block_best(best, w_own, sh_c, sh_w);
    if (threadIdx.x == 0) {
      const double stay = (double)w_own - g * kv * Ka_wo;
      int target = a;
      bool wants = false, allowed = true;
      if (best.c >= 0 && best.val > stay) {
        wants = true;
        target = best.c;
        const unsigned int pa = prio(a, seed), pb = best.pr;
        if (round >= 0)
          allowed = (round & 1) ? (pb > pa || (pb == pa && target > a)) : (pb < pa || (pb == pa && target < a));
      } else if (stay < 0.0 && Ka_wo > 0.0 && csize[v] == 0) {
        wants = true;
        target = v;
      }
      decision[w] = (wants && allowed) ? target : (wants ? -2 : -1);
    }
    __syncthreads();
  }
}

// One thread per active vertex: apply the decided moves in place (integer atomics on the community totals); vertices
// blocked by the direction rule stay active.  counters: [0] moved, [1] blocked
__global__ __launch_bounds__(256) void ld_apply_kernel(int n_act, const int* __restrict__ list,
                                                       const int* __restrict__ decision, const long long* __restrict__ k,
                                                       int* __restrict__ comm, unsigned long long* __restrict__ Ktot,
                                                       int* __restrict__ csize, int* __restrict__ flag,
                                                       int* __restrict__ counters) {
  __shared__ int s_moved, s_blocked;
  if (threadIdx.x == 0) {
    s_moved = 0;
    s_blocked = 0;
  }
  __syncthreads();
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w < n_act) {
    const int d = decision[w];
    if (d >= 0) {
      const int v = list[w];
      const int a = comm[v];
      comm[v] = d;
      const unsigned long long kq = (unsigned long long)k[v];
      atomicAdd(&Ktot[d], kq);
      atomicAdd(&Ktot[a], 0ull - kq);
      atomicAdd(&csize[d], 1);
      atomicSub(&csize[a], 1);
      atomicAdd(&s_moved, 1);
    } else if (d == -2) {
      flag[list[w]] = 1;
      atomicAdd(&s_blocked, 1);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (s_moved) atomicAdd(&counters[0], s_moved);
    if (s_blocked) atomicAdd(&counters[1], s_blocked);
  }
}

// After ALL moves of the sub-round are applied: every mover re-activates its neighbours that are NOT in its new
// community (the queue rule of the sequential algorithm, leidenalg `move_nodes`: a neighbour inside the new community can
// only have become more attached).  A separate launch, because the test reads the neighbours' communities and those
// must be the sub-round's final ones for the flags to be reproducible.  Round 2 flagged every neighbour: after the first
// sweep of a level most of the active list was vertices deep inside their community.
// The rows are walked by the whole wave, two movers at a time (32 lanes each, coalesced row reads).
__device__ __forceinline__ void ld_requeue_body(int bid, int n_act, const int* __restrict__ list,
                                                const int* __restrict__ decision, const int64_t* __restrict__ indptr,
                                                const int* __restrict__ indices, const int* __restrict__ comm,
                                                int* __restrict__ flag) {
  const int w = bid * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  int64_t beg = 0, end = 0;
  int d = -1;
  if (w < n_act) {
    d = decision[w];
    if (d >= 0) {
      const int v = list[w];
      beg = indptr[v];
      end = indptr[v + 1];
    }
  }
  unsigned long long m = __ballot(d >= 0);
  while (m) {
    const int b0 = __ffsll((long long)m) - 1;
    m &= m - 1;
    int b1 = b0;
    if (m) {
      b1 = __ffsll((long long)m) - 1;
      m &= m - 1;
    }
    const int src = lane < 32 ? b0 : b1;
    // (all shuffles run with every lane active: a bpermute under a partial exec mask reads 0 from disabled lanes)
    const int64_t rb = __shfl(beg, src), re_s = __shfl(end, src);
    const int dd = __shfl(d, src);
    const int64_t re = (lane >= 32 && b1 == b0) ? rb : re_s;
    for (int64_t e = rb + (lane & 31); e < re; e += 32) {
      const int u = indices[e];
      if (comm[u] != dd) flag[u] = 1;
    }
  }
}
__global__ __launch_bounds__(256) void ld_requeue_kernel(int n_act, const int* __restrict__ list,
                                                         const int* __restrict__ decision,
                                                         const int64_t* __restrict__ indptr,
                                                         const int* __restrict__ indices, const int* __restrict__ comm,
                                                         int* __restrict__ flag) {
  ld_requeue_body((int)blockIdx.x, n_act, list, decision, indptr, indices, comm, flag);
}
// Sub-round c's re-queue and sub-round c + 1's decisions in ONE launch (round 4).  They are independent: the re-queue
// reads the decisions of class c (their own buffer: `decision` alternates between two buffers from sub-round to
// sub-round) and the communities after apply(c), and writes flags; the decisions of class c + 1 read the same state and
// write their own buffer and lists.  The first nb_rq workgroups re-queue, the rest decide.  One launch of ~16 us less per
// sub-round (152 per call at 1M cells).
template <int G>
__global__ SCAMD_LD_MOVE_LB void ld_requeue_move_kernel(int nb_rq, int rq_n_act, const int* __restrict__ rq_list,
                                                              const int* __restrict__ rq_decision,
                                                              const int64_t* __restrict__ indptr,
                                                              const int* __restrict__ indices, const int* __restrict__ comm,
                                                              int* __restrict__ flag, MoveArgs ma) {
  if ((int)blockIdx.x < nb_rq) ld_requeue_body((int)blockIdx.x, rq_n_act, rq_list, rq_decision, indptr, indices, comm, flag);
  else ld_move_body<G>((int)blockIdx.x - nb_rq, (int)gridDim.x - nb_rq, ma);
}

// ---- final polish (n_iterations < 0): strictly monotone single-vertex moves ----------------------------------------
// The class sub-rounds above apply many moves decided on ONE snapshot: fast, but not monotone, and the outer loop keeps the
// best partition it saw -- which, on ambiguous graphs, may be one whose last coarse-level moves left a few level-0
// vertices improvable (round 4: 9 / 186 of 300k vertices, gains < 1e-6 Q; the paper's "node optimality" of a stable
// partition, oracle/leiden_guarantees.py, violated by that residue).  The polish decides on a snapshot as before, but
// applies a move v: A -> B only if v holds the locks of BOTH communities: lock[c] = max over this round's movers touching
// c of (round << 32 | ~v) -- the smallest vertex id wins, stale rounds lose to the current one, nothing is ever reset.
// The gain of v's move depends on k_v(A), k_v(B), K_A, K_B only; moves with disjoint community pairs leave those four
// untouched, so every applied move realises exactly the gain it was decided on: Q rises strictly, round after round,
// and at least one mover (the smallest id) wins per round.  Losers stay flagged.
__global__ __launch_bounds__(256) void ld_polish_lock_kernel(int n_act, const int* __restrict__ list,
                                                             const int* __restrict__ decision, const int* __restrict__ comm,
                                                             unsigned long long* __restrict__ lock, unsigned int round) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_act) return;
  const int d = decision[w];
  if (d < 0) return;
  const int v = list[w];
  const unsigned long long key = ((unsigned long long)round << 32) | (unsigned long long)(0xffffffffu - (unsigned int)v);
  atomicMax(&lock[comm[v]], key);
  atomicMax(&lock[d], key);
}
// winners move (as ld_apply_kernel), losers keep their flag and their decision becomes "stay" (the re-queue that follows
// reads decision >= 0 as "moved").  counters: [0] moved, [1] lost a lock
__global__ __launch_bounds__(256) void ld_polish_apply_kernel(int n_act, const int* __restrict__ list, int* __restrict__ decision,
                                                              const long long* __restrict__ k, int* __restrict__ comm,
                                                              unsigned long long* __restrict__ Ktot, int* __restrict__ csize,
                                                              const unsigned long long* __restrict__ lock, unsigned int round,
                                                              int* __restrict__ flag, int* __restrict__ counters) {
  __shared__ int s_moved, s_lost;
  if (threadIdx.x == 0) {
    s_moved = 0;
    s_lost = 0;
  }
  __syncthreads();
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w < n_act) {
    const int d = decision[w];
    if (d >= 0) {
      const int v = list[w];
      const int a = comm[v];
      const unsigned long long key = ((unsigned long long)round << 32) | (unsigned long long)(0xffffffffu - (unsigned int)v);
      if (lock[a] == key && lock[d] == key) {
        comm[v] = d;
        const unsigned long long kq = (unsigned long long)k[v];
        atomicAdd(&Ktot[d], kq);
        atomicAdd(&Ktot[a], 0ull - kq);
        atomicAdd(&csize[d], 1);
        atomicSub(&csize[a], 1);
        atomicAdd(&s_moved, 1);
      } else {
        decision[w] = -1;
        flag[v] = 1;
        atomicAdd(&s_lost, 1);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (s_moved) atomicAdd(&counters[0], s_moved);
    if (s_lost) atomicAdd(&counters[1], s_lost);
  }
}

// A vertex that leaves may have been the only link between two parts of its community.  After a polish that moved
// anything: connected components INSIDE the communities by min-label propagation (comp[v] -> the smallest vertex id of
// its component; in place, monotone, with one pointer jump per visit), and every component becomes a community of its
// own (splitting a community along a cut without edges raises Q by 2 g K1 K2 / (2m)^2 > 0).
__global__ __launch_bounds__(256) void ld_cc_prop_kernel(int n, const int64_t* __restrict__ indptr, const int* __restrict__ indices,
                                                         const int* __restrict__ comm, int* __restrict__ comp, int* __restrict__ changed) {
  // 16 lanes per vertex: the row is read coalesced, the neighbours' labels gathered 16 at a time (a thread per vertex walked
  // its row one dependent gather after the other: 1.0 ms per pass at 1M vertices)
  const int v = (blockIdx.x * 256 + threadIdx.x) >> 4, sub = threadIdx.x & 15;
  const bool live = v < n;
  const int vv = live ? v : 0;
  const int c = comm[vv];
  const int old = __hip_atomic_load(&comp[vv], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int m = old;
  if (live) {
    for (int64_t e = indptr[v] + sub; e < indptr[v + 1]; e += 16) {
      const int u = indices[e];
      if (comm[u] == c) m = min(m, __hip_atomic_load(&comp[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) m = min(m, __shfl_xor(m, o));  // (every lane of the wave takes part)
  if (live && sub == 0) {
    m = min(m, __hip_atomic_load(&comp[m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));  // (comp[m] <= m, same component)
    if (m < old) {
      atomicMin(&comp[v], m);
      *changed = 1;
    }
  }
}
// out[0] += components (roots), out[1] += non-empty communities
__global__ __launch_bounds__(256) void ld_cc_count_kernel(int n, const int* __restrict__ comp, const int* __restrict__ csize,
                                                          int* __restrict__ out) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  const bool root = v < n && comp[v] == v, alive = v < n && csize[v] > 0;
  const unsigned long long mr = __ballot(root), ma = __ballot(alive);
  if ((threadIdx.x & 63) == 0) {
    if (mr) atomicAdd(&out[0], __popcll(mr));
    if (ma) atomicAdd(&out[1], __popcll(ma));
  }
}

// Class of a vertex in a sweep: a fresh hash per sweep (salt), so two neighbours that shared a class -- and could
// therefore move at the same time, e.g. swap communities -- almost surely do not share it in the next sweep.
__device__ __forceinline__ int lm_class(int v, unsigned int salt, int n_cls) {
  return (int)((hash32((unsigned int)v * 0x9E3779B1u + salt) >> 9) & (unsigned int)(n_cls - 1));
}

// flags -> n_cls class lists (lists[c * n ..]), counts in cls_count[0 .. n_cls); clears the flags.  flag == nullptr:
// every vertex is active (first sweep of a level).  The order inside a list is irrelevant (every decision of a
// sub-round reads the same snapshot).  One atomic per class and 1024-vertex block.
// Long rows of a class, counted while its list is built (round 4): sub-round c's counter block gets the number of its
// vertices whose rows the main decision kernel hands on -- longer than thr_mid (the lanes-per-vertex table) / longer than
// thr_hub (the wave table) -- so that the host launches the overflow / hub kernels of a sub-round only when it has any
// (they were launched blind: ~300 near-empty launches of ~10 us per call at 1M cells).  thr_mid < 0: no such tier.
constexpr int CTR_N_MID = 8, CTR_N_HUB = 9;
// tier of a vertex's row: bit 0 = longer than thr_mid, bit 1 = longer than thr_hub (0 for inactive vertices)
__device__ __forceinline__ int long_row_tier(int cls, int v, const int64_t* __restrict__ indptr, int thr_mid, int thr_hub) {
  if (cls < 0 || !indptr) return 0;
  const int deg = (int)(indptr[v + 1] - indptr[v]);
  return ((thr_mid >= 0 && deg > thr_mid) ? 1 : 0) | (deg > thr_hub ? 2 : 0);
}
// one atomic per wave, class and tier (inside the class loop of the list builders; every lane of the wave calls it)
__device__ __forceinline__ void count_long_rows(int c, int cls, int tier, int lane, int* __restrict__ cls_count) {
  const unsigned long long mm = __ballot(cls == c && (tier & 1)), mh = __ballot(cls == c && (tier & 2));
  if (lane == 0) {
    int* ctr = cls_count + MAX_CLASSES + CTR_STRIDE * c;
    if (mm) atomicAdd(&ctr[CTR_N_MID], __popcll(mm));
    if (mh) atomicAdd(&ctr[CTR_N_HUB], __popcll(mh));
  }
}

__global__ __launch_bounds__(1024) void ld_compact_cls_kernel(int n, int* __restrict__ flag, int* __restrict__ lists,
                                                              int* __restrict__ cls_count, int n_cls, unsigned int salt,
                                                              const int64_t* __restrict__ indptr, int thr_mid, int thr_hub,
                                                              int* __restrict__ clear_next, int clear_words) {
  __shared__ int wcnt[16][MAX_CLASSES];
  __shared__ int base[MAX_CLASSES];
  // (the counter block the NEXT sweep will count into: its last users were the sub-rounds of the sweep before this one)
  if (clear_next && blockIdx.x == 0)
    for (int i = threadIdx.x; i < clear_words; i += 1024) clear_next[i] = 0;
  const int v = blockIdx.x * 1024 + threadIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int cls = -1;
  if (v < n) {
    int f = 1;
    if (flag) {
      f = flag[v];
      if (f) flag[v] = 0;
    }
    if (f) cls = lm_class(v, salt, n_cls);
  }
  const int tier = long_row_tier(cls, v, indptr, thr_mid, thr_hub);
  int rank = 0;
  for (int c = 0; c < n_cls; ++c) {
    const unsigned long long m = __ballot(cls == c);
    if (cls == c) rank = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wcnt[wv][c] = __popcll(m);
    count_long_rows(c, cls, tier, lane, cls_count);
  }
  __syncthreads();
  if ((int)threadIdx.x < n_cls) {
    const int c = threadIdx.x;
    int tot = 0;
    for (int i = 0; i < 16; ++i) {
      const int t = wcnt[i][c];
      wcnt[i][c] = tot;
      tot += t;
    }
    base[c] = tot ? atomicAdd(&cls_count[c], tot) : 0;
  }
  __syncthreads();
  if (cls >= 0) lists[(size_t)cls * n + base[cls] + wcnt[wv][cls] + rank] = v;
}

// Packed mirrors of the refinement's state (round 4).  Counters of the decision kernels (profiles/r04n_leiden_kernels_pmc*.csv):
// ld_refine_propose_kernel<16> waits 55 % of its wave-cycles on memory and moves 0.7 GB (FETCH_SIZE with the gfx950
// correction) through the memory side of L2 per launch of ~100k candidates -- ~7 KB per candidate: a 64-byte sector for every
// 4- or 8-byte gather of comm[u], ref[u] (two per neighbour) and Kref[c], refsize[c], Eref[c] (three per distinct target),
// at an L2 hit rate of 24 %.  The proposing kernels and the cut update therefore read ONE record per neighbour and ONE per
// target; the plain arrays stay the authoritative copy for everything else (aggregation, coarse ids), every writer of
// the refinement updates both.  All fields are integers: same results.  Measured: 31.3 -> 30.6 ms on the planted 1M graph
// (profiles/r04p_leiden_packed_records_ab.log) -- less than the sector count promised: the kernels wait on the length of
// their dependency chain as much as on the bytes.
struct alignas(16) VertRec {  // per vertex: phase-1 community, refined community, sub-round in which it joined (-1: never)
  int comm, ref, stamp, pad;
};
struct alignas(32) TargRec {  // per refined community (indexed by its representative vertex)
  unsigned long long Kref, Eref;
  int refsize, pad0, pad1, pad2;
};

// ---- phase 2: refinement ---------------------------------------------------------------------------
// a_in[v] = w(v, C(v) - v): weight from v to the rest of its (phase-1) community
template <int G>
__global__ __launch_bounds__(256) void ld_within_kernel(int n, const int64_t* __restrict__ indptr,
                                                        const int* __restrict__ indices,
                                                        const long long* __restrict__ wq, const int* __restrict__ comm,
                                                        long long* __restrict__ a_in) {
  const int sub = threadIdx.x % G;
  const int v = blockIdx.x * (256 / G) + threadIdx.x / G;
  if (v >= n) return;
  const int a = comm[v];
  const int64_t beg = indptr[v], end = indptr[v + 1];
  long long s = 0;
  // four entries per lane in flight: a hub row (1.4k entries on the 1M kNN graph) walked one dependent gather pair at
  // a time by 16 lanes is ~90 serial round trips -- the tail of the whole launch
  for (int64_t e = beg + sub; e < end; e += 4 * G) {
    int u[4], cu[4];
    long long we[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int64_t ee = e + t * G;
      u[t] = ee < end ? indices[ee] : v;
      we[t] = ee < end ? wq[ee] : 0ll;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) cu[t] = comm[u[t]];
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (u[t] != v && cu[t] == a) s += we[t];
  }
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (sub == 0) a_in[v] = s;
}

// candidates of the refinement: vertices that are well connected inside their community
// (w(v, C - v) >= gamma k_v (K_C - k_v) / 2m); all of them start as singletons.  They are written as n_cls class lists
// (lists[c * n ..], counts in cls_count[0 .. n_cls)): class c is visited in sub-round c of the refinement.
__global__ __launch_bounds__(1024) void ld_refine_candidates_kernel(
    int n, const long long* __restrict__ k, const int* __restrict__ comm, const unsigned long long* __restrict__ Ktot,
    const long long* __restrict__ a_in, double g, int* __restrict__ lists, int* __restrict__ cls_count, int n_cls,
    unsigned int salt, const int64_t* __restrict__ indptr, int thr_mid, int thr_hub) {
  __shared__ int wcnt[16][MAX_CLASSES];
  __shared__ int base[MAX_CLASSES];
  const int v = blockIdx.x * 1024 + threadIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int cls = -1;
  if (v < n) {
    const double kv = (double)k[v];
    const double KC = (double)(long long)Ktot[comm[v]];
    if ((double)a_in[v] >= g * kv * (KC - kv)) cls = lm_class(v, salt, n_cls);
  }
  const int tier = long_row_tier(cls, v, indptr, thr_mid, thr_hub);
  int rank = 0;
  for (int c = 0; c < n_cls; ++c) {
    const unsigned long long m = __ballot(cls == c);
    if (cls == c) rank = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wcnt[wv][c] = __popcll(m);
    count_long_rows(c, cls, tier, lane, cls_count);
  }
  __syncthreads();
  if ((int)threadIdx.x < n_cls) {
    const int c = threadIdx.x;
    int tot = 0;
    for (int i = 0; i < 16; ++i) {
      const int t = wcnt[i][c];
      wcnt[i][c] = tot;
      tot += t;
    }
    base[c] = tot ? atomicAdd(&cls_count[c], tot) : 0;
  }
  __syncthreads();
  if (cls >= 0) lists[(size_t)cls * n + base[cls] + wcnt[wv][cls] + rank] = v;
}

// Exact incremental update of w(r, C - r) after a round of merges.  For a vertex v that joined t this round
// (stamp[v] == round), with t_old = the members t had before the round and J = the other joiners of the round:
//   Eref[t] += w(v, C - v) - 2 w(v, t_old) - w(v, J)
// (summed over the joiners every {v, v'} in J x J pair is subtracted twice, as the cut of the union requires).
// One wave per joiner; all integer, so the result equals a from-scratch recomputation bit for bit.
template <int G>
__global__ __launch_bounds__(256) void ld_refine_cut_update_kernel(
    int n_join, const int* __restrict__ jlist, const int64_t* __restrict__ indptr, const int* __restrict__ indices,
    const long long* __restrict__ wq, const int* __restrict__ comm, const int* __restrict__ ref,
    const int* __restrict__ stamp, const long long* __restrict__ a_in, int round,
    unsigned long long* __restrict__ Eref, const int* __restrict__ n_join_dev, const VertRec* __restrict__ vr,
    TargRec* __restrict__ tr) {
  const int sub = threadIdx.x % G;
  n_join = *n_join_dev;
  for (int w = blockIdx.x * (256 / G) + threadIdx.x / G; w < n_join; w += gridDim.x * (256 / G)) {
  // (everything that hangs on the same address is requested together: the && chain of the obvious form is four
  // round trips per neighbour -- comm[u], then ref[u], then stamp[u], then the weight)
  const int v = jlist[w];
  const int a = comm[v], t = ref[v];
  const int64_t beg = indptr[v], end = indptr[v + 1];
  const long long av = a_in[v];
  long long s = 0;
  for (int64_t e = beg + sub; e < end; e += 2 * G) {  // two entries per lane in flight (hub rows: see ld_within_kernel)
    int u[2], cu[2], ru[2], su[2];
    long long we[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t ee = e + j * G;
      u[j] = ee < end ? indices[ee] : v;
      we[j] = ee < end ? wq[ee] : 0ll;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {  // one 16-byte record per neighbour instead of three gathers
      const VertRec r = vr[u[j]];
      cu[j] = r.comm;
      ru[j] = r.ref;
      su[j] = r.stamp;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (u[j] != v && cu[j] == a && ru[j] == t) s += (su[j] == round) ? we[j] : 2 * we[j];
  }
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (sub == 0) {
    atomicAdd(&Eref[t], (unsigned long long)(av - s));
    atomicAdd(&tr[t].Eref, (unsigned long long)(av - s));
  }
  }
}

// Sub-round c of the refinement: the singletons of class c propose, the targets are grown sub-communities and singletons
// of OTHER classes (a singleton of the same class may itself leave in this sub-round), so simultaneous merges cannot
// chain; classes < c have had their turn (their results are visible), classes > c get theirs later unless somebody
// joins them first.  After the n_cls sub-rounds every candidate was considered exactly once -- leidenalg's
// `merge_nodes_constrained` visits every vertex once, in random order.
// Randomised merge rule of the refinement (Traag et al. 2019, leidenalg's `refine_consider_comms` with theta = beta):
// v joins r with probability ~ exp(gain(v, r) / beta) among the well-connected candidates of non-negative gain, "stay"
// (gain 0) included.  Sampled with the Gumbel-max trick: argmax_r gain_r / beta + G(v, r, round, seed) with
// counter-based noise, so the choice is a pure function of its arguments -- every kernel variant draws the same
// target, runs are bitwise reproducible.  (Round 1 took the beta -> 0 limit; on graphs without clean structure the
// greedy rule ends in visibly worse optima: 0.8101 vs 0.8121 on the bundled fixture, the same value the CPU oracle
// reaches when its beta is set to 1e-7.)  c = v encodes "stay".
__device__ __forceinline__ double refine_noise(int v, int c, int round, unsigned int seed) {
  const unsigned int h = hash32((unsigned int)v * 0x9E3779B1u ^
                                hash32((unsigned int)c * 0x85EBCA77u + (unsigned int)round * 0xC2B2AE3Du + seed));
  const double u = ((double)h + 0.5) * (1.0 / 4294967296.0);
  return -log(-log(u));
}

// G lanes per candidate (see ld_move_kernel: G = 16 puts four candidates in a wave on short-rowed levels, rows longer
// than the 128-slot table go to ovf_list / counters[5] and are proposed by the G = 64 instantiation in indirect mode).
// `list` holds the candidates of class `round` (the sub-round's number).  target[v] = refined community to join,
// -1 = no admissible target, -2 = no longer a singleton (somebody joined it) or "stay" was drawn.
template <int G>
__global__ SCAMD_LD_PROPOSE_LB void ld_refine_propose_kernel(
    const int* __restrict__ list, const int* __restrict__ sub_list, const int* __restrict__ sub_count,
    const int64_t* __restrict__ indptr, const int* __restrict__ indices, const long long* __restrict__ wq,
    const long long* __restrict__ k, const int* __restrict__ comm, const unsigned long long* __restrict__ Ktot,
    const int* __restrict__ ref, const int* __restrict__ refsize, const unsigned long long* __restrict__ Kref,
    const unsigned long long* __restrict__ Eref, double g, double inv_beta /* 1 / (beta * 2^32); 0 = greedy */,
    int round, int n_cls, unsigned int salt, unsigned int seed, int* __restrict__ target,
    int* __restrict__ ovf_list, int* __restrict__ hub_list, int* __restrict__ counters, int n_cand,
    const VertRec* __restrict__ vr, const TargRec* __restrict__ tr) {
  constexpr int GROUPS = 256 / G;
  constexpr int GSLOTS = G == 16 ? G16_SLOTS : WH_SLOTS * G / 64;
  constexpr int GMAX = GSLOTS * 3 / 4;
  __shared__ int hkeys[GROUPS][GSLOTS];
  __shared__ unsigned long long hvals[GROUPS][GSLOTS];
  const int sub = threadIdx.x % G;
  const int grp = threadIdx.x / G;
  const int n_items = sub_list ? *sub_count : n_cand;
  for (int item = blockIdx.x * GROUPS + grp; item < n_items; item += gridDim.x * GROUPS) {
    // (gathers grouped by what they depend on, as in ld_move_kernel: list -> {refsize, ref, k, comm, indptr}[v] ->
    // {Ktot[a], indices, wq} -> {comm, ref}[u] -> {Kref, refsize, Eref}[c]; most launches of a refinement are short
    // lists whose time IS the length of this chain)
    const int w = sub_list ? sub_list[item] : item;
    const int v = list[w];
    const int rs_v = refsize[v], ref_v = ref[v];
    const long long kq = k[v];
    const int a = comm[v];
    const int64_t beg = indptr[v];
    const int64_t end = indptr[v + 1];
    int tgt = -1;
    if (rs_v != 1 || ref_v != v) {
      tgt = -2;
    } else {
      const int deg = (int)(end - beg);
      int u_pre[2];
      long long w_pre[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int e = sub + t * G;
        u_pre[t] = e < deg ? indices[beg + e] : v;
        w_pre[t] = e < deg ? wq[beg + e] : 0ll;
      }
      const double KC = (double)(long long)Ktot[a];
      int cm_pre[2], rf_pre[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const VertRec r = vr[u_pre[t]];
        cm_pre[t] = r.comm;
        rf_pre[t] = r.ref;
      }
      const double kv = (double)kq;
      if (G < 64) {
        if (deg > GMAX) {  // proposed by the wave-per-candidate instantiation
          if (sub == 0) ovf_list[atomicAdd(&counters[5], 1)] = w;
          continue;
        }
      } else if (deg > WH_MAX_DEG) {  // hub: proposed by ld_refine_propose_hub_kernel (any length)
        if (sub == 0) hub_list[atomicAdd(&counters[4], 1)] = v;
        continue;
      }
      Cand best;
      best.val = 0.0;
      best.c = -1;
      best.pr = 0;
      if (deg <= GMAX) {
        int* keys = hkeys[grp];
        unsigned long long* vals = hvals[grp];
        const int nslots = G < 64 ? (deg <= 48 ? 64 : 128) : (deg <= 96 ? 128 : (deg <= 192 ? 256 : 512));
        for (int i = sub; i < nslots; i += G) {
          keys[i] = WH_EMPTY;
          vals[i] = 0ull;
        }
        auto insert = [&](int c, long long wt) {
          unsigned int slot = hash32((unsigned int)c) & (nslots - 1);
          for (;;) {
            const int prev = atomicCAS(&keys[slot], WH_EMPTY, c);
            if (prev == WH_EMPTY || prev == c) break;
            slot = (slot + 1) & (nslots - 1);
          }
          atomicAdd(&vals[slot], (unsigned long long)wt);
        };
#pragma unroll
        for (int t = 0; t < 2; ++t)
          if (sub + t * G < deg && u_pre[t] != v && cm_pre[t] == a) insert(rf_pre[t], w_pre[t]);
        for (int e = sub + 2 * G; e < deg; e += G) {
          const int u = indices[beg + e];
          const VertRec r = vr[u];
          if (u != v && r.comm == a) insert(r.ref, wq[beg + e]);
        }
        constexpr int MAXSL = GSLOTS / G;
        int cs[MAXSL], rsz[MAXSL];
        unsigned long long kr[MAXSL], er[MAXSL];
#pragma unroll
        for (int t = 0; t < MAXSL; ++t) cs[t] = WH_EMPTY;
#pragma unroll
        for (int t = 0; t < MAXSL; ++t) {
          const int sl = sub + t * G;
          if (__ballot(sl < nslots)) {  // (see ld_move_kernel: uniform skip, no per-lane branch around the gathers)
            int c = sl < nslots ? __hip_atomic_load(&keys[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : WH_EMPTY;
            if (c == v) c = WH_EMPTY;
            cs[t] = c;
            const int ci = c != WH_EMPTY ? c : v;
            const TargRec x = tr[ci];  // one 32-byte record per target instead of three gathers
            kr[t] = x.Kref;
            rsz[t] = x.refsize;
            er[t] = x.Eref;
          }
        }
#pragma unroll
        for (int t = 0; t < MAXSL; ++t) {
          const int c = cs[t];
          if (c != WH_EMPTY) {
            const long long sum =
                (long long)__hip_atomic_load(&vals[sub + t * G], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const double Kr = (double)(long long)kr[t];
            const bool single = rsz[t] == 1;
            const bool ok_target = (!single || lm_class(c, salt, n_cls) != round) &&
                                   ((double)(long long)er[t] >= g * Kr * (KC - Kr));  // target well connected
            const double gain = (double)sum - g * kv * Kr;
            if (ok_target && gain >= 0.0) {
              Cand x;
              x.val = inv_beta > 0.0 ? gain * inv_beta + refine_noise(v, c, round, seed) : gain;
              x.c = c;
              x.pr = prio(c, seed);
              if (cand_better(x, best)) best = x;
            }
          }
        }
      }
#pragma unroll
      for (int o = G / 2; o > 0; o >>= 1) {
        Cand y;
        y.val = __shfl_xor(best.val, o);
        y.c = __shfl_xor(best.c, o);
        y.pr = (unsigned int)__shfl_xor((int)best.pr, o);
        if (cand_better(y, best)) best = y;
      }
      tgt = best.c;
      // "stay" drawn against the best candidate: a singleton that stays is done (it may still be joined)
      if (inv_beta > 0.0 && tgt >= 0 && !(best.val > refine_noise(v, v, round, seed))) tgt = -2;
    }
    if (sub == 0) target[v] = tgt;
  }
}

// Hub candidates (vertex ids in hub_list[0 .. counters[4])): one workgroup each; same rule as the wave kernel.
__global__ __launch_bounds__(HUB_THREADS) void ld_refine_propose_hub_kernel(
    const int* __restrict__ hub_list, int* __restrict__ counters, const int64_t* __restrict__ indptr,
    const int* __restrict__ indices, const long long* __restrict__ wq, const long long* __restrict__ k,
    const int* __restrict__ comm, const unsigned long long* __restrict__ Ktot, const int* __restrict__ ref,
    const int* __restrict__ refsize, const unsigned long long* __restrict__ Kref,
    const unsigned long long* __restrict__ Eref, double g, double inv_beta, int round, int n_cls, unsigned int salt,
    unsigned int seed, int* __restrict__ target, int* __restrict__ err, const VertRec* __restrict__ vr,
    const TargRec* __restrict__ tr, int try_probes) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long hub_smem[];
  BlockHash bh{reinterpret_cast<int*>(hub_smem + BHUB_SLOTS), hub_smem, BHUB_SLOTS};
  __shared__ Cand sh_c[HUB_THREADS / 64];
  __shared__ long long sh_w[HUB_THREADS / 64];
  __shared__ int sh_fail;
  const int n_hub = counters[4];
  for (int i = blockIdx.x; i < n_hub; i += gridDim.x) {
    const int v = hub_list[i];
    const double kv = (double)k[v];
    const int a = comm[v];
    const double KC = (double)(long long)Ktot[a];
    const int64_t beg = indptr[v];
    const int deg = (int)(indptr[v + 1] - beg);
    const int n_pass = bhub_passes(deg);
    bh.size_for(n_pass > 1 ? BHUB_SLOTS : deg);
    Cand best;
    best.val = 0.0;
    best.c = -1;
    best.pr = 0;
    long long dummy = 0;
    bool filled = false;  // (hub_try_single_pass: the comment at HUB_TRY_PROBES)
    if (n_pass > 1 && try_probes > 0) {
      if (threadIdx.x == 0) sh_fail = 0;
      bh.clear();
      __syncthreads();
      for (int e = threadIdx.x; e < deg; e += blockDim.x) {
        if (__hip_atomic_load(&sh_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
        const int u = indices[beg + e];
        const VertRec r = vr[u];
        if (u == v || r.comm != a) continue;
        if (!bh.add_limited(r.ref, wq[beg + e], try_probes)) __hip_atomic_store(&sh_fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      __syncthreads();
      filled = sh_fail == 0;
      __syncthreads();
    }
    const int n_sweep = filled ? 1 : n_pass;
    for (int pass = 0; pass < n_sweep; ++pass) {
      if (!filled) {
        bh.clear();
        __syncthreads();
        for (int e = threadIdx.x; e < deg; e += blockDim.x) {
          const int u = indices[beg + e];
          const VertRec r = vr[u];
          if (u == v || r.comm != a) continue;
          const int c = r.ref;
          if (n_pass == 1) bh.add(c, wq[beg + e]);
          else if (bhub_class(c, n_pass) == pass && !bh.add_bounded(c, wq[beg + e])) *err = 1;
        }
        __syncthreads();
      }
      for (int sl = threadIdx.x; sl < bh.nslots; sl += blockDim.x) {
        const int c = bh.keys[sl];
        if (c != WH_EMPTY && c != v) {
          const long long sum = (long long)bh.vals[sl];
          const TargRec x = tr[c];
          const double Kr = (double)(long long)x.Kref;
          const bool single = x.refsize == 1;
          const bool ok_target = (!single || lm_class(c, salt, n_cls) != round) &&
                                 ((double)(long long)x.Eref >= g * Kr * (KC - Kr));
          const double gain = (double)sum - g * kv * Kr;
          if (ok_target && gain >= 0.0) {
            Cand x;
            x.val = inv_beta > 0.0 ? gain * inv_beta + refine_noise(v, c, round, seed) : gain;
            x.c = c;
            x.pr = prio(c, seed);
            if (cand_better(x, best)) best = x;
          }
        }
      }
      __syncthreads();
    }
    best = block_best(best, dummy, sh_c, sh_w);
    if (threadIdx.x == 0)
      target[v] = (inv_beta > 0.0 && best.c >= 0 && !(best.val > refine_noise(v, v, round, seed))) ? -2 : best.c;
    __syncthreads();
  }
}

// counters: [0] merges of this sub-round (= joiner list length)
__global__ void ld_refine_apply_kernel(int n_cand, const int* __restrict__ list, const int* __restrict__ target,
                                       const long long* __restrict__ k, int* __restrict__ ref,
                                       int* __restrict__ refsize, unsigned long long* __restrict__ Kref,
                                       unsigned long long* __restrict__ Eref, int* __restrict__ stamp, int round,
                                       int* __restrict__ jlist, int* __restrict__ counters, VertRec* __restrict__ vr,
                                       TargRec* __restrict__ tr) {
  const int lane = threadIdx.x & 63;
  for (int w0 = blockIdx.x * blockDim.x; w0 < n_cand; w0 += gridDim.x * blockDim.x) {
    const int w = w0 + threadIdx.x;
    const int v = (w < n_cand) ? list[w] : -1;
    const int t = (v >= 0) ? target[v] : -2;
    // list appends are aggregated per wave: one returning atomic per wave instead of one per element
    const unsigned long long mj = __ballot(t >= 0);
    int bj = 0;
    if (lane == 0 && mj) bj = atomicAdd(&counters[0], __popcll(mj));
    bj = __shfl(bj, 0);
    if (t >= 0) {
      // v is a singleton (ref[v] == v) joining t; t's members do not move in this sub-round
      ref[v] = t;
      atomicAdd(&refsize[t], 1);
      atomicAdd(&Kref[t], (unsigned long long)k[v]);
      refsize[v] = 0;
      Kref[v] = 0;
      Eref[v] = 0;
      stamp[v] = round;
      // the packed mirrors (v's community field does not change during the refinement)
      vr[v].ref = t;
      vr[v].stamp = round;
      atomicAdd(&tr[t].refsize, 1);
      atomicAdd(&tr[t].Kref, (unsigned long long)k[v]);
      tr[v].Kref = 0;
      tr[v].Eref = 0;
      tr[v].refsize = 0;
      jlist[bj + __popcll(mj & ((1ull << lane) - 1ull))] = v;
    }
  }
}

__global__ void ld_refine_init_kernel(int n, const long long* __restrict__ k, const long long* __restrict__ a_in,
                                      const int* __restrict__ comm, int* __restrict__ ref, int* __restrict__ refsize,
                                      unsigned long long* __restrict__ Kref, unsigned long long* __restrict__ Eref,
                                      VertRec* __restrict__ vr, TargRec* __restrict__ tr, int* __restrict__ touched,
                                      int* __restrict__ rc0, int rc0_words, int* __restrict__ counters) {
  int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.x == 0) {  // class list lengths + per-sub-round counters of this refinement, phase counters
    for (int i = threadIdx.x; i < rc0_words; i += blockDim.x) rc0[i] = 0;
    if (threadIdx.x < 8) counters[threadIdx.x] = 0;
  }
  if (v < n) {
    touched[v] = -1;  // join sub-round stamps: -1 = never
    ref[v] = v;
    refsize[v] = 1;
    Kref[v] = (unsigned long long)k[v];
    Eref[v] = (unsigned long long)a_in[v];  // singleton: w(v, C - v)
    vr[v] = VertRec{comm[v], v, -1, 0};
    tr[v] = TargRec{(unsigned long long)k[v], (unsigned long long)a_in[v], 1, 0, 0, 0};
  }
}

// ---- phase 3: aggregation --------------------------------------------------------------------------
__global__ void ld_flag_kernel(int n, const int* __restrict__ size, int* __restrict__ flag, int* __restrict__ fill_max) {
  int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v < n) {
    flag[v] = size[v] > 0 ? 1 : 0;
    if (fill_max) fill_max[v] = 0x7f7f7f7f;  // (`rep`: the atomicMin target of ld_coarse_ids_kernel)
  }
}

// coarse id of every node + representative coarse id of its phase-1 community (min over its members)
__global__ __launch_bounds__(1024) void ld_coarse_ids_kernel(int n, const int* __restrict__ ref,
                                                             const int64_t* __restrict__ newid_of_ref,
                                                             const int* __restrict__ comm, int* __restrict__ cid,
                                                             int* __restrict__ rep) {
  __shared__ int keys[BH_SLOTS];
  __shared__ int mn[BH_SLOTS];
  for (int i = threadIdx.x; i < BH_SLOTS; i += 1024) {
    keys[i] = BH_EMPTY;
    mn[i] = 0x7fffffff;
  }
  __syncthreads();
  const int v = blockIdx.x * 1024 + threadIdx.x;
  if (v < n) {
    const int c = (int)newid_of_ref[ref[v]];
    cid[v] = c;
    const int cm = comm[v];
    const int slot = bh_find_slot(keys, cm);
    if (slot >= 0) atomicMin(&mn[slot], c);
    else atomicMin(&rep[cm], c);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < BH_SLOTS; i += 1024)
    if (keys[i] != BH_EMPTY) atomicMin(&rep[keys[i]], mn[i]);
}
__global__ void ld_coarse_comm_kernel(int n, const int* __restrict__ cid, const int* __restrict__ comm,
                                      const int* __restrict__ rep, int* __restrict__ comm_new, int nn,
                                      int* __restrict__ cursor, int* __restrict__ counters) {
  int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v < n) comm_new[cid[v]] = rep[comm[v]];
  if (v < nn) cursor[v] = 0;  // (nn <= n: fill cursors of the member scatter)
  if (v < 8) counters[v] = 0;
}
__global__ void ld_remap_kernel(int n_orig, const int* __restrict__ cid, int* __restrict__ node_of) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_orig) node_of[i] = cid[node_of[i]];
}

// Coarse-graph construction, member-grouped: the members of every coarse node are gathered contiguously
// (counting sort by coarse id), then ONE wave / workgroup per coarse node accumulates its members' rows into an
// LDS hash keyed by the neighbour's coarse id and emits the combined row.  No global hash table and no global
// atomics on the edges: the level-0 graph (24M entries) is read once, sequentially per member.
//   rows land at an upper-bound offset (sum of the members' degrees) in a scratch CSR and are compacted after
//   the row lengths are scanned.
constexpr int AGG_MID_SLOTS = 4096;   // workgroup tier for <= 2048 distinct neighbours (48 KB LDS: 3 blocks/CU)
constexpr int AGG_MID_MAX = 2048;
constexpr int AGG_BIG_PASS = 4096;    // the 8192-slot tier takes ~4096 distinct keys per pass

__global__ void ld_agg_mcount_kernel(int n, const int* __restrict__ refsize, const int64_t* __restrict__ newid,
                                     int* __restrict__ mcount) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n && refsize[r] > 0) mcount[(int)newid[r]] = refsize[r];
}

__global__ void ld_agg_scatter_kernel(int n, const int* __restrict__ cid, const int64_t* __restrict__ moff,
                                      int* __restrict__ cursor, const int64_t* __restrict__ indptr,
                                      int* __restrict__ members, int* __restrict__ mdeg) {
  int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  const int c = cid[v];
  const int64_t p = moff[c] + atomicAdd(&cursor[c], 1);
  members[p] = v;
  mdeg[p] = (int)(indptr[v + 1] - indptr[v]);
}

// distinct neighbours of a coarse node are bounded by both its degree sum and the coarse node count
__device__ __forceinline__ int64_t agg_need(int64_t dsum, int nn) { return dsum < nn ? dsum : (int64_t)nn; }
// a split row's parts are merged through ONE 8192-slot table: every key of the level must fit it (load <= 0.69)
constexpr int AGG_SPLIT_NN_MAX = 5600;

// one wave per coarse node; nodes needing more than the wave table go to the mid / big lists
// (counters[4] / counters[5])
__global__ __launch_bounds__(256) void ld_agg_wave_kernel(
    int nn, const int64_t* __restrict__ moff, const int64_t* __restrict__ eoff, const int* __restrict__ members,
    const int64_t* __restrict__ indptr, const int* __restrict__ indices, const long long* __restrict__ wq,
    const int* __restrict__ cid, int* __restrict__ s_col, long long* __restrict__ s_w, int* __restrict__ rowcnt,
    int* __restrict__ mid_list, int* __restrict__ big_list, int* __restrict__ counters, int wave_max, int mid_max,
    int wave_work, int mid_work, int* __restrict__ split_list, int64_t split_work) {
  __shared__ int hkeys[4][WH_SLOTS];
  __shared__ unsigned long long hvals[4][WH_SLOTS];
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= nn) return;
  const int64_t m0 = moff[c], m1 = moff[c + 1];
  const int64_t u0 = eoff[m0];
  const int64_t dsum = eoff[m1] - u0;
  const int64_t need = agg_need(dsum, nn);
  // tiers by table size (distinct neighbours: `need`) AND by work (member entries: `dsum`).  Round 5: on the coarse levels of
  // a graph without clear clusters a coarse vertex has few distinct neighbours (need <= nn, a few hundred) but tens of
  // thousands of member entries -- one wave walked them all (this kernel was 26 % of the Leiden time on the weak 1M graph,
  // single launches of up to 7.9 ms); rows beyond wave_work entries go to a workgroup, beyond mid_work to the 1024-thread one
  if (need > wave_max || dsum > wave_work) {
    if (lane == 0) {
      // (the very largest rows of a coarse level -- ~1e6 member entries behind a few hundred distinct neighbours -- are cut
      // into parts for several workgroups and merged: ld_agg_parts_kernel; only where one table holds every key of the level)
      if (dsum > split_work && nn <= AGG_SPLIT_NN_MAX) split_list[atomicAdd(&counters[6], 1)] = c;
      else if (need <= mid_max && dsum <= mid_work) mid_list[atomicAdd(&counters[4], 1)] = c;
      else big_list[atomicAdd(&counters[5], 1)] = c;
    }
    return;
  }
  WaveHash wh{hkeys[threadIdx.x >> 6], hvals[threadIdx.x >> 6], WH_SLOTS};
  wh.size_for((int)need);
  wh.clear(lane);
  // The members' row extents are fetched 64 at a time (one coalesced gather), then two rows are in flight at once:
  // walking the members one by one is a chain members -> indptr -> indices -> cid per 24-entry row with nothing else
  // in flight.
  for (int64_t i0 = m0; i0 < m1; i0 += 64) {
    const int cnt = (int)(m1 - i0 < 64 ? m1 - i0 : 64);
    long long mb = 0, me = 0;
    if (lane < cnt) {
      const int v = members[i0 + lane];
      mb = indptr[v];
      me = indptr[v + 1];
    }
    for (int j = 0; j < cnt; j += 2) {
      const int j1 = j + 1 < cnt ? j + 1 : j;
      const long long b0 = readlane_i64(mb, j), e0 = readlane_i64(me, j);
      const long long b1 = readlane_i64(mb, j1), e1 = j1 != j ? readlane_i64(me, j1) : b1;
      const bool h0 = b0 + lane < e0, h1 = b1 + lane < e1;
      const int x0 = h0 ? indices[b0 + lane] : 0, x1 = h1 ? indices[b1 + lane] : 0;
      const long long w0 = h0 ? wq[b0 + lane] : 0ll, w1 = h1 ? wq[b1 + lane] : 0ll;
      const int c0 = cid[x0], c1 = cid[x1];
      if (h0) wh.add(c0, w0);
      if (h1) wh.add(c1, w1);
      for (int64_t e = b0 + 64 + lane; e < e0; e += 64) wh.add(cid[indices[e]], wq[e]);
      for (int64_t e = b1 + 64 + lane; e < e1; e += 64) wh.add(cid[indices[e]], wq[e]);
    }
  }
  int base = 0;
  for (int s0 = 0; s0 < wh.nslots; s0 += 64) {
    const int key = wh.key(s0 + lane);
    const bool has = key != WH_EMPTY;
    const unsigned long long m = __ballot(has);
    if (has) {
      const int64_t p = u0 + base + __popcll(m & ((1ull << lane) - 1ull));
      s_col[p] = key;
      s_w[p] = wh.val(s0 + lane);
    }
    base += __popcll(m);
  }
  if (lane == 0) rowcnt[c] = base;
}

// one workgroup per listed coarse node.  SLOTS-entry LDS table; when the node may have more distinct neighbours
// than pass_keys the keys are split into hash classes and the members' rows are swept once per class.
template <int SLOTS, int THREADS>
__global__ __launch_bounds__(THREADS) void ld_agg_block_kernel(
    const int* __restrict__ list, const int* __restrict__ list_len, int nn, const int64_t* __restrict__ moff,
    const int64_t* __restrict__ eoff, const int* __restrict__ members, const int64_t* __restrict__ indptr,
    const int* __restrict__ indices, const long long* __restrict__ wq, const int* __restrict__ cid,
    int* __restrict__ s_col, long long* __restrict__ s_w, int* __restrict__ rowcnt, int* __restrict__ err,
    int pass_keys, int try_probes) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long agg_smem[];
  unsigned long long* vals = agg_smem;
  int* keys = reinterpret_cast<int*>(agg_smem + SLOTS);
  __shared__ int sh_cnt, sh_fail;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int n_list = *list_len;
  for (int li = blockIdx.x; li < n_list; li += gridDim.x) {
    const int c = list[li];
    const int64_t m0 = moff[c], m1 = moff[c + 1];
    const int64_t u0 = eoff[m0];
    const int64_t need = agg_need(eoff[m1] - u0, nn);
    // (a part of a split row may be empty -- a member row longer than the part size spans several cuts: need = 0)
    const unsigned int n_pass = (unsigned int)std::max<int64_t>(1, (need + pass_keys - 1) / pass_keys);
    int nslots = 256;
    {
      const int64_t per_pass = (need + n_pass - 1) / n_pass;
      while (nslots < 2 * per_pass && nslots < SLOTS) nslots <<= 1;
      if (n_pass > 1) nslots = SLOTS;
    }
    if (threadIdx.x == 0) {
      sh_cnt = 0;
      sh_fail = 0;
    }
    // pass -1 (rows that would need several passes): ONE optimistic pass over all keys with bounded probing -- the bound
    // `need` is the members' entry count or the coarse vertex count, the distinct neighbours are usually far fewer (the
    // comment at HUB_TRY_PROBES); if a key finds no slot the class-by-class passes 0 .. n_pass - 1 follow as before
    for (int pass = ((n_pass > 1 && try_probes > 0) ? -1 : 0); pass < (int)n_pass; ++pass) {
      for (int i = threadIdx.x; i < nslots; i += THREADS) {
        keys[i] = WH_EMPTY;
        vals[i] = 0ull;
      }
      __syncthreads();
      // the extent of a wave's next member row is requested while the current row is walked
      int64_t nb = 0, ne = 0;
      if (m0 + wv < m1) {
        const int v = members[m0 + wv];
        nb = indptr[v];
        ne = indptr[v + 1];
      }
      for (int64_t i = m0 + wv; i < m1; i += THREADS / 64) {
        if (pass < 0 && __hip_atomic_load(&sh_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;  // (trial lost)
        const int64_t rb = nb, re = ne;
        if (i + THREADS / 64 < m1) {
          const int vn = members[i + THREADS / 64];
          nb = indptr[vn];
          ne = indptr[vn + 1];
        }
        auto add_entry = [&](int key, long long we) {
          if (pass >= 0 && n_pass > 1 && (hash32((unsigned int)key * 0x9E3779B1u + 0x7F4A7C15u) >> 8) % n_pass != (unsigned int)pass) return;
          unsigned int slot = hash32((unsigned int)key) & (nslots - 1);
          const int max_tries = pass < 0 ? try_probes : nslots;
          int tries = 0;
          for (;;) {
            const int prev = atomicCAS(&keys[slot], WH_EMPTY, key);
            if (prev == WH_EMPTY || prev == key) {
              atomicAdd(&vals[slot], (unsigned long long)we);
              break;
            }
            slot = (slot + 1) & (nslots - 1);
            if (++tries > max_tries) {
              // trial pass: fall back to the class passes; class pass: table full -- cannot happen for uniformly
              // hashed classes; reported, not hidden
              if (pass < 0) __hip_atomic_store(&sh_fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              else *err = 1;
              break;
            }
          }
        };
        // Long member rows (the coarse levels of a graph without clear clusters: thousands of entries per row, a million
        // per coarse vertex) were walked 64 entries per dependent round trip (indices -> cid): four steps' loads are
        // requested before the first is used (round 5)
        int64_t e = rb + lane;
        for (; e + 192 < re; e += 256) {
          const int x0 = indices[e], x1 = indices[e + 64], x2 = indices[e + 128], x3 = indices[e + 192];
          const long long w0 = wq[e], w1 = wq[e + 64], w2 = wq[e + 128], w3 = wq[e + 192];
          const int k0 = cid[x0], k1 = cid[x1], k2 = cid[x2], k3 = cid[x3];
          add_entry(k0, w0);
          add_entry(k1, w1);
          add_entry(k2, w2);
          add_entry(k3, w3);
        }
        for (; e < re; e += 64) add_entry(cid[indices[e]], wq[e]);
      }
      __syncthreads();
      if (pass < 0 && sh_fail) continue;  // (read by every thread after the barrier: uniform)
      for (int s0 = 0; s0 < nslots; s0 += THREADS) {
        const int sl = s0 + threadIdx.x;
        const int key = sl < nslots ? keys[sl] : WH_EMPTY;  // the table in use may be shorter than the workgroup
        const bool has = key != WH_EMPTY;
        const unsigned long long m = __ballot(has);
        int wbase = 0;
        if (lane == 0 && m) wbase = atomicAdd(&sh_cnt, __popcll(m));
        wbase = __shfl(wbase, 0);
        if (has) {
          const int64_t p = u0 + wbase + __popcll(m & ((1ull << lane) - 1ull));
          s_col[p] = key;
          s_w[p] = (long long)vals[sl];
        }
      }
      __syncthreads();
      if (pass < 0) break;  // the one pass held every key
    }
    if (threadIdx.x == 0) rowcnt[c] = sh_cnt;
    __syncthreads();
  }
}

// Split rows (round 5).  A launch of the workgroup builders lasts as long as its longest row, and on the coarse levels of a
// graph without clear clusters a few coarse vertices hold ~1e6 member entries each (ld_agg_block_kernel<8192, 1024>: launches
// of 1-3.5 ms, 10-13 % of the Leiden time on the weak / structure-less 1M graphs).  Such a row is cut into PARTS of ~chunk
// member entries (whole member rows: the cut points are found by bisection in the members' entry offsets); every part is
// built like a row of its own by ld_agg_block_kernel -- the parts are given pseudo-row ids whose member ranges live in
// `pmoff` (one extra slot per row closes its last part), their lists land at their own members' scratch offsets -- and one
// workgroup per split row merges the parts' lists through a table and writes the row.  Integer sums: the row is the same.
__global__ void ld_agg_parts_kernel(const int* __restrict__ split_list, int* __restrict__ counters, const int64_t* __restrict__ moff,
                                    const int64_t* __restrict__ eoff, int64_t chunk, int64_t* __restrict__ pmoff,
                                    int* __restrict__ part_list, int* __restrict__ split_first, int* __restrict__ split_np) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= counters[6]) return;
  const int c = split_list[s];
  const int64_t m0 = moff[c], m1 = moff[c + 1];
  const int64_t u0 = eoff[m0], dsum = eoff[m1] - u0;
  const int np = (int)((dsum + chunk - 1) / chunk);
  const int base = atomicAdd(&counters[2], np + 1);
  split_first[s] = base;
  split_np[s] = np;
  for (int j = 0; j < np; ++j) {
    int64_t lo = m0;
    if (j > 0) {  // first member whose entries start at or after the j-th cut
      const int64_t target = u0 + (int64_t)j * chunk;
      int64_t hi = m1;
      while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (eoff[mid] < target) lo = mid + 1;
        else hi = mid;
      }
    }
    pmoff[base + j] = lo;
    part_list[atomicAdd(&counters[3], 1)] = base + j;
  }
  pmoff[base + np] = m1;
}

__global__ __launch_bounds__(1024) void ld_agg_merge_kernel(const int* __restrict__ split_list, const int* __restrict__ counters,
                                                            const int64_t* __restrict__ moff, const int64_t* __restrict__ eoff,
                                                            const int64_t* __restrict__ pmoff, const int* __restrict__ part_cnt,
                                                            const int* __restrict__ split_first, const int* __restrict__ split_np,
                                                            int* __restrict__ s_col, long long* __restrict__ s_w,
                                                            int* __restrict__ rowcnt, int* __restrict__ err) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long agg_smem[];
  unsigned long long* vals = agg_smem;
  int* keys = reinterpret_cast<int*>(agg_smem + BHUB_SLOTS);
  __shared__ int sh_cnt;
  const int lane = threadIdx.x & 63;
  const int n_split = counters[6];
  for (int s = blockIdx.x; s < n_split; s += gridDim.x) {
    const int c = split_list[s];
    const int64_t u0 = eoff[moff[c]];
    const int first = split_first[s], np = split_np[s];
    for (int i = threadIdx.x; i < BHUB_SLOTS; i += 1024) {
      keys[i] = WH_EMPTY;
      vals[i] = 0ull;
    }
    if (threadIdx.x == 0) sh_cnt = 0;
    __syncthreads();
    for (int j = 0; j < np; ++j) {
      const int64_t up = eoff[pmoff[first + j]];
      const int cnt = part_cnt[first + j];
      for (int t = threadIdx.x; t < cnt; t += 1024) {
        const int key = s_col[up + t];
        const long long w = s_w[up + t];
        unsigned int slot = hash32((unsigned int)key) & (BHUB_SLOTS - 1);
        for (int tries = 0;; ++tries) {
          const int prev = atomicCAS(&keys[slot], WH_EMPTY, key);
          if (prev == WH_EMPTY || prev == key) {
            atomicAdd(&vals[slot], (unsigned long long)w);
            break;
          }
          slot = (slot + 1) & (BHUB_SLOTS - 1);
          if (tries > BHUB_SLOTS) {  // (cannot happen: at most AGG_SPLIT_NN_MAX distinct keys; reported, not hidden)
            *err = 1;
            break;
          }
        }
      }
    }
    __syncthreads();  // every partial list has been read: the row is written over them
    for (int s0 = 0; s0 < BHUB_SLOTS; s0 += 1024) {
      const int sl = s0 + threadIdx.x;
      const int key = keys[sl];
      const bool has = key != WH_EMPTY;
      const unsigned long long m = __ballot(has);
      int wbase = 0;
      if (lane == 0 && m) wbase = atomicAdd(&sh_cnt, __popcll(m));
      wbase = __shfl(wbase, 0);
      if (has) {
        const int64_t p = u0 + wbase + __popcll(m & ((1ull << lane) - 1ull));
        s_col[p] = key;
        s_w[p] = (long long)vals[sl];
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) rowcnt[c] = sh_cnt;
    __syncthreads();
  }
}

// scratch rows (upper-bound offsets) -> final CSR rows; one wave per coarse node
__global__ __launch_bounds__(256) void ld_agg_compact_kernel(int nn, const int64_t* __restrict__ moff,
                                                             const int64_t* __restrict__ eoff,
                                                             const int64_t* __restrict__ indptr_new,
                                                             const int* __restrict__ s_col,
                                                             const long long* __restrict__ s_w,
                                                             int* __restrict__ out_col, long long* __restrict__ out_w,
                                                             int* __restrict__ dstat) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (blockIdx.x == 0 && threadIdx.x < 4) dstat[threadIdx.x] = 0;  // (row-length statistics of the level being built: ld_degstats_kernel follows)
  if (c >= nn) return;
  const int64_t u0 = eoff[moff[c]];
  const int64_t p0 = indptr_new[c];
  const int cnt = (int)(indptr_new[c + 1] - p0);
  for (int i = lane; i < cnt; i += 64) {
    out_col[p0 + i] = s_col[u0 + i];
    out_w[p0 + i] = s_w[u0 + i];
  }
}

// ---- small levels: everything in ONE workgroup ---------------------------------------------------------
// The last levels of every run are tiny (planted 1M cells: 170, 80, 67, 64 nodes; the 700-cell fixture is such a level
// from the start): as separate kernels each of them costs ~65 launches and 5 host round trips = 0.5 ms of pure
// latency, ten such levels per call.  From the first level with at most SMALL_N nodes and SMALL_NNZ entries on, ONE
// workgroup runs local moving, refinement and aggregation of all remaining levels: the per-node state lives in LDS, the
// rows stay in global memory (written by this workgroup, read after a device-scope fence), 8 waves take 8 nodes at a
// time -- decide on a snapshot, barrier, apply, barrier -- so at most 8 nodes move at once, closer to the sequential
// algorithm than any class schedule (a node's neighbour weights are summed in a dense per-wave table indexed by
// community, with a touched list: no hashing, any row length).  Same arithmetic and same rules as the kernels above.
constexpr int SMALL_N = 1024;
constexpr int SMALL_NNZ = 65536;
constexpr int SM_WAVES = 8;
constexpr int SM_THREADS = SM_WAVES * 64;

struct SmallArgs {
  const int64_t* indptr;   // entry level
  const int* indices;
  const long long* wq;
  const long long* k;
  int n;
  int* ix[2];              // coarse levels, ping-pong
  long long* w[2];
  int first_dst;
  int* comm;               // in: partition of the entry nodes; out: final community of every entry node
  double gg;               // gamma / 2m
  double inv_beta;
  unsigned int seed;
  int iter;
  int lm_stop_permille;
  int seq_n;               // levels of at most this many vertices move one vertex at a time
  int* info;               // [0] levels, [1] moves, [2] merges, [3] nodes of the last level
};

struct SmallLds {
  unsigned long long acc[SM_WAVES][SMALL_N];
  unsigned short touched[SM_WAVES][SMALL_N];
  unsigned long long Ktot[SMALL_N];
  long long kk[SMALL_N];
  int ip[SMALL_N + 1];
  int comm[SMALL_N];
  int csize[SMALL_N];          // local moving: community sizes; aggregation: new partition of the coarse nodes
  unsigned short order[SMALL_N];
  unsigned short tail_of[SMALL_N];
  unsigned char flag[SMALL_N];
  int ref[SMALL_N];
  int refsize[SMALL_N];
  union {
    struct {
      unsigned long long Kref[SMALL_N];
      unsigned long long Eref[SMALL_N];
      long long a_in[SMALL_N];
      int stamp[SMALL_N];
    } rf;
    struct {
      int cid[SMALL_N];
      int members[SMALL_N];
      int moff[SMALL_N + 1];
      int rep[SMALL_N];
      int rowptr[SMALL_N + 1];
    } ag;
  } u;
  int tcnt[SM_WAVES];
  int dec[SM_WAVES];
  int rset[SM_WAVES];
  int scan_w[SM_WAVES];
  int s_n_act, s_moved, s_merged, s_nn;
};
static_assert(sizeof(SmallLds) <= 160 * 1024, "SmallLds must fit the 160 KB of LDS");

// exclusive scan of x[0 .. n) (n <= 2 * SM_THREADS) in place; returns the total.  All threads call.
__device__ __forceinline__ int small_scan(int* x, int n, int* scan_w) {
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int i0 = 2 * t, i1 = 2 * t + 1;
  const int a0 = i0 < n ? x[i0] : 0, a1 = i1 < n ? x[i1] : 0;
  int inc = a0 + a1;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(inc, o);
    if (lane >= o) inc += y;
  }
  if (lane == 63) scan_w[wv] = inc;
  __syncthreads();
  int off = 0, tot = 0;
  for (int i = 0; i < SM_WAVES; ++i) {
    const int s = scan_w[i];
    if (i < wv) off += s;
    tot += s;
  }
  const int ex = off + inc - (a0 + a1);
  if (i0 < n) x[i0] = ex;
  if (i1 < n) x[i1] = ex + a0;
  __syncthreads();
  return tot;
}

// flagged items of [0, n) -> order[0 .. count), ascending; returns count.  pred(v) evaluated by thread v % SM_THREADS.
template <typename P>
__device__ __forceinline__ int small_compact(int n, P pred, unsigned short* order, int* scan_w) {
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  int base = 0;
  for (int v0 = 0; v0 < n; v0 += SM_THREADS) {
    const int v = v0 + t;
    const bool f = v < n && pred(v);
    const unsigned long long m = __ballot(f);
    if (lane == 0) scan_w[wv] = __popcll(m);
    __syncthreads();
    int off = 0, tot = 0;
    for (int i = 0; i < SM_WAVES; ++i) {
      const int s = scan_w[i];
      if (i < wv) off += s;
      tot += s;
    }
    if (f) order[base + off + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)v;
    base += tot;
    __syncthreads();
  }
  return base;
}

constexpr int SMALL_SEQ_N = 16;   // levels of at most this many vertices: local moving one vertex at a time (SCAMD_LEIDEN_SMALL_SEQ)
__global__ __launch_bounds__(SM_THREADS) void ld_small_levels_kernel(SmallArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char small_smem[];
  SmallLds& L = *reinterpret_cast<SmallLds*>(small_smem);
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int n_in = a.n;
  int n = a.n;
  const int* ix = a.indices;
  const long long* wq = a.wq;
  int dst = a.first_dst;
  // ---- entry level into LDS
  for (int v = t; v < n; v += SM_THREADS) {
    L.ip[v] = (int)(a.indptr[v] - a.indptr[0]);
    L.kk[v] = a.k[v];
    L.comm[v] = a.comm[v];
    L.tail_of[v] = (unsigned short)v;
  }
  if (t == 0) L.ip[n] = (int)(a.indptr[n] - a.indptr[0]);
  ix += a.indptr[0];
  wq += a.indptr[0];
  for (int i = t; i < SM_WAVES * SMALL_N; i += SM_THREADS) (&L.acc[0][0])[i] = 0ull;
  __syncthreads();
  int levels = 0, tot_moves = 0, tot_merges = 0;
  unsigned long long* acc = L.acc[wv];
  unsigned short* touched = L.touched[wv];

  // weight from row v towards each distinct group key(u) of its neighbours u (u != v, pass(u)): dense table + touched
  // list of this wave; returns the number of distinct groups.  Entries of weight 0 carry nothing and are skipped (the
  // first-touch test is `old sum == 0`).
  auto gather_row = [&](int v, auto key, auto pass) -> int {
    if (lane == 0) L.tcnt[wv] = 0;
    const int beg = L.ip[v], end = L.ip[v + 1];
    for (int e = beg + lane; e < end; e += 64) {
      const int u = ix[e];
      const long long w = wq[e];
      if (u != v && w != 0 && pass(u)) {
        const int c = key(u);
        const unsigned long long old = atomicAdd(&acc[c], (unsigned long long)w);
        if (old == 0ull) touched[atomicAdd(&L.tcnt[wv], 1)] = (unsigned short)c;
      }
    }
    return __hip_atomic_load(&L.tcnt[wv], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };

  for (;;) {
    ++levels;
    // ================= local moving =================
    for (int c = t; c < n; c += SM_THREADS) {
      L.Ktot[c] = 0ull;
      L.csize[c] = 0;
      L.flag[c] = 1;
    }
    __syncthreads();
    for (int v = t; v < n; v += SM_THREADS) {
      atomicAdd(&L.Ktot[L.comm[v]], (unsigned long long)L.kk[v]);
      atomicAdd(&L.csize[L.comm[v]], 1);
    }
    __syncthreads();
    int moved_prev2 = 0, quiet = 0;
    for (int sweep = 0; sweep < MAX_LM_SWEEPS; ++sweep) {
      const int n_act = small_compact(n, [&](int v) { return L.flag[v] != 0; }, L.order, L.scan_w);
      if (n_act == 0) break;
      if (t == 0) L.s_moved = 0;
      int m = 1;
      while (m < n_act) m <<= 1;
      const unsigned int hs = hash32(a.seed + 0x85EBCA77u * (unsigned int)(sweep + 1) + 0xC2B2AE3Du * (unsigned int)a.iter +
                                     0x27D4EB2Fu * (unsigned int)levels);
      const unsigned int pa = hs | 1u, pb = hs >> 11;
      const int dir_round = sweep >= LM_DIR_AFTER ? sweep : -1;
      __syncthreads();
      // Vertices that decide in the same round do not see each other's moves.  Eight at a time is harmless among
      // hundreds of vertices; on the small DENSE graphs of the upper levels (every pair adjacent) it made neighbours swap
      // communities round after round -- found on the host emulator: a 2-vertex graph ended as two singletons
      // (Q = -0.5), 21 of 60 random graphs of <= 40 vertices below the CPU oracle.  There the vertices decide one at a
      // time, as in the sequential algorithm (a round is three barriers: 128 rounds cost less than one launch).
      const int par = n <= a.seq_n ? 1 : SM_WAVES;
      for (int r0 = 0; r0 < m; r0 += par) {
        const int pos = (int)(((unsigned int)(r0 + wv) * pa + pb) & (unsigned int)(m - 1));
        const int v = (wv < par && pos < n_act) ? (int)L.order[pos] : -1;
        int decision = -1;
        if (v >= 0) {
          const int ca = L.comm[v];
          const long long kq = L.kk[v];
          const double kv = (double)kq;
          const double Ka_wo = (double)(long long)(L.Ktot[ca] - (unsigned long long)kq);
          const int tc = gather_row(v, [&](int u) { return L.comm[u]; }, [&](int) { return true; });
          Cand best;
          best.val = 0.0;
          best.c = -1;
          best.pr = 0;
          long long w_own = 0;
          for (int i = lane; i < tc; i += 64) {
            const int c = touched[i];
            const long long sum = (long long)acc[c];
            acc[c] = 0ull;
            if (c == ca) {
              w_own = sum;
            } else {
              Cand x;
              x.val = (double)sum - a.gg * kv * (double)(long long)L.Ktot[c];
              x.c = c;
              x.pr = prio(c, a.seed);
              if (cand_better(x, best)) best = x;
            }
          }
          best = wave_best(best);
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) w_own = max(w_own, __shfl_xor(w_own, o));
          const double stay = (double)w_own - a.gg * kv * Ka_wo;
          if (best.c >= 0 && best.val > stay) {
            bool allowed = true;
            if (dir_round >= 0) {
              const unsigned int p0 = prio(ca, a.seed), p1 = best.pr;
              allowed = (dir_round & 1) ? (p1 > p0 || (p1 == p0 && best.c > ca)) : (p1 < p0 || (p1 == p0 && best.c < ca));
            }
            decision = allowed ? best.c : -2;
          } else if (stay < 0.0 && Ka_wo > 0.0 && L.csize[v] == 0) {
            decision = v;  // an empty community (its id = the node's own id) beats staying
          }
        }
        if (lane == 0) {
          L.dec[wv] = decision;
          if (v >= 0) L.flag[v] = 0;  // (cleared BEFORE the barrier: flags set by this round's movers must survive)
        }
        __syncthreads();
        if (v >= 0) {
          const int d = L.dec[wv];
          if (d == -2 && lane == 0) L.flag[v] = 1;  // blocked by the direction rule: stays active
          if (d >= 0) {
            if (lane == 0) {
              const int ca = L.comm[v];
              const unsigned long long kq = (unsigned long long)L.kk[v];
              L.comm[v] = d;
              atomicAdd(&L.Ktot[d], kq);
              atomicAdd(&L.Ktot[ca], 0ull - kq);
              atomicAdd(&L.csize[d], 1);
              atomicSub(&L.csize[ca], 1);
              atomicAdd(&L.s_moved, 1);
            }
          }
        }
        __syncthreads();
        if (v >= 0 && L.dec[wv] >= 0) {  // after every move of the round: re-activate the neighbours outside the new community
          const int d = L.dec[wv];
          for (int e = L.ip[v] + lane; e < L.ip[v + 1]; e += 64) {
            const int u = ix[e];
            if (L.comm[u] != d) L.flag[u] = 1;
          }
        }
        __syncthreads();
      }
      const int moved = L.s_moved;
      tot_moves += moved;
      __syncthreads();
      if (moved == 0 && dir_round < 0) break;
      quiet = moved == 0 ? quiet + 1 : 0;
      if (quiet >= 2) break;
      if (a.iter == 0 && sweep >= 1) {
        if ((long long)moved * 1000 < (long long)n * a.lm_stop_permille) break;
        if (sweep >= 2 && (long long)moved * 10 > (long long)moved_prev2 * 9) break;
      }
      moved_prev2 = moved;
    }
    __syncthreads();
    // ================= refinement =================
    for (int v0 = 0; v0 < n; v0 += SM_WAVES) {  // a_in[v] = w(v, C(v) - v)
      const int v = v0 + wv;
      if (v < n) {
        const int ca = L.comm[v];
        long long s = 0;
        for (int e = L.ip[v] + lane; e < L.ip[v + 1]; e += 64) {
          const int u = ix[e];
          if (u != v && L.comm[u] == ca) s += wq[e];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) {
          L.u.rf.a_in[v] = s;
          L.ref[v] = v;
          L.refsize[v] = 1;
          L.u.rf.Kref[v] = (unsigned long long)L.kk[v];
          L.u.rf.Eref[v] = (unsigned long long)s;
          L.u.rf.stamp[v] = -1;
        }
      }
    }
    if (t == 0) L.s_merged = 0;
    __syncthreads();
    const unsigned int rseed = a.seed + 0x9E3779B9u * (unsigned int)a.iter;
    const int n_cand = small_compact(n, [&](int v) {
      const double kv = (double)L.kk[v];
      const double KC = (double)(long long)L.Ktot[L.comm[v]];
      return (double)L.u.rf.a_in[v] >= a.gg * kv * (KC - kv);
    }, L.order, L.scan_w);
    if (n_cand > 0) {
      int m = 1;
      while (m < n_cand) m <<= 1;
      const unsigned int hs = hash32(rseed ^ (0x5bd1e995u + 0x27D4EB2Fu * (unsigned int)levels));
      const unsigned int pa = hs | 1u, pb = hs >> 11;
      int round = 0;
      for (int r0 = 0; r0 < m; r0 += SM_WAVES, ++round) {
        const int pos = (int)(((unsigned int)(r0 + wv) * pa + pb) & (unsigned int)(m - 1));
        int v = pos < n_cand ? (int)L.order[pos] : -1;
        if (v >= 0 && (L.refsize[v] != 1 || L.ref[v] != v)) v = -1;  // somebody joined it: no longer a singleton
        if (lane == 0) L.rset[wv] = v;
        __syncthreads();
        int tgt = -1;
        if (v >= 0) {
          const int ca = L.comm[v];
          const double kv = (double)L.kk[v];
          const double KC = (double)(long long)L.Ktot[ca];
          const int tc = gather_row(v, [&](int u) { return L.ref[u]; }, [&](int u) { return L.comm[u] == ca; });
          Cand best;
          best.val = 0.0;
          best.c = -1;
          best.pr = 0;
          for (int i = lane; i < tc; i += 64) {
            const int c = touched[i];
            const long long sum = (long long)acc[c];
            acc[c] = 0ull;
            if (c == v) continue;
            const double Kr = (double)(long long)L.u.rf.Kref[c];
            bool ok = (double)(long long)L.u.rf.Eref[c] >= a.gg * Kr * (KC - Kr);  // target well connected
            if (ok && L.refsize[c] == 1) {  // a singleton that proposes in this very round may leave: not a target
#pragma unroll
              for (int j = 0; j < SM_WAVES; ++j) ok = ok && L.rset[j] != c;
            }
            const double gain = (double)sum - a.gg * kv * Kr;
            if (ok && gain >= 0.0) {
              Cand x;
              x.val = a.inv_beta > 0.0 ? gain * a.inv_beta + refine_noise(v, c, round, rseed) : gain;
              x.c = c;
              x.pr = prio(c, rseed);
              if (cand_better(x, best)) best = x;
            }
          }
          best = wave_best(best);
          tgt = best.c;
          if (a.inv_beta > 0.0 && tgt >= 0 && !(best.val > refine_noise(v, v, round, rseed))) tgt = -1;  // "stay" drawn
        }
        __syncthreads();  // every proposal of the round is made before any is applied
        if (v >= 0 && tgt >= 0 && lane == 0) {
          L.ref[v] = tgt;
          atomicAdd(&L.refsize[tgt], 1);
          atomicAdd(&L.u.rf.Kref[tgt], (unsigned long long)L.kk[v]);
          L.refsize[v] = 0;
          L.u.rf.Kref[v] = 0ull;
          L.u.rf.Eref[v] = 0ull;
          L.u.rf.stamp[v] = round;
          atomicAdd(&L.s_merged, 1);
        }
        __syncthreads();
        if (v >= 0 && tgt >= 0) {  // exact update of w(t, C - t): see ld_refine_cut_update_kernel
          const int ca = L.comm[v];
          long long s = 0;
          for (int e = L.ip[v] + lane; e < L.ip[v + 1]; e += 64) {
            const int u = ix[e];
            if (u != v && L.comm[u] == ca && L.ref[u] == tgt) s += (L.u.rf.stamp[u] == round) ? wq[e] : 2 * wq[e];
          }
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
          if (lane == 0) atomicAdd(&L.u.rf.Eref[tgt], (unsigned long long)(L.u.rf.a_in[v] - s));
        }
        __syncthreads();
      }
    }
    __syncthreads();
    const int merged = L.s_merged;
    tot_merges += merged;
    if (merged == 0) break;
    // ================= aggregation =================
    __syncthreads();
    for (int r = t; r < n; r += SM_THREADS) L.u.ag.rowptr[r] = L.refsize[r] > 0 ? 1 : 0;  // (rowptr as scratch: new ids)
    __syncthreads();
    const int nn = small_scan(L.u.ag.rowptr, n, L.scan_w);
    if (nn == n) break;
    for (int v = t; v < n; v += SM_THREADS) L.u.ag.rep[v] = 0x7fffffff;
    __syncthreads();
    for (int v = t; v < n; v += SM_THREADS) {
      const int c = L.u.ag.rowptr[L.ref[v]];
      L.u.ag.cid[v] = c;
      atomicMin(&L.u.ag.rep[L.comm[v]], c);
    }
    __syncthreads();
    for (int r = t; r < n; r += SM_THREADS)
      if (L.refsize[r] > 0) L.u.ag.moff[L.u.ag.rowptr[r]] = L.refsize[r];  // member counts per coarse node
    for (int v = t; v < n; v += SM_THREADS) L.csize[L.u.ag.cid[v]] = L.u.ag.rep[L.comm[v]];  // partition of the coarse nodes
    for (int i = t; i < n_in; i += SM_THREADS) L.tail_of[i] = (unsigned short)L.u.ag.cid[L.tail_of[i]];
    __syncthreads();
    small_scan(L.u.ag.moff, nn, L.scan_w);
    if (t == 0) L.u.ag.moff[nn] = n;
    __syncthreads();
    // members of every coarse node in ascending node order: every node counts the members of its coarse node that
    // precede it (n <= 1024: ~1000 LDS reads per thread) -- an atomic cursor would make the order run dependent
    for (int v = t; v < n; v += SM_THREADS) {
      const int c = L.u.ag.cid[v];
      int before = 0;
      for (int u = 0; u < v; ++u) before += L.u.ag.cid[u] == c ? 1 : 0;
      L.u.ag.members[L.u.ag.moff[c] + before] = v;
    }
    __syncthreads();
    int* nix = a.ix[dst];
    long long* nw = a.w[dst];
    for (int pass = 0; pass < 2; ++pass) {
      for (int c0 = 0; c0 < nn; c0 += SM_WAVES) {
        const int c = c0 + wv;
        if (c < nn) {
          if (lane == 0) L.tcnt[wv] = 0;
          for (int mi = L.u.ag.moff[c]; mi < L.u.ag.moff[c + 1]; ++mi) {
            const int v = L.u.ag.members[mi];
            for (int e = L.ip[v] + lane; e < L.ip[v + 1]; e += 64) {
              const long long w = wq[e];
              if (w != 0) {
                const int key = L.u.ag.cid[ix[e]];
                const unsigned long long old = atomicAdd(&acc[key], (unsigned long long)w);
                if (old == 0ull) touched[atomicAdd(&L.tcnt[wv], 1)] = (unsigned short)key;
              }
            }
          }
          const int tc = __hip_atomic_load(&L.tcnt[wv], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          long long ksum = 0;
          const int p0 = pass ? L.u.ag.rowptr[c] : 0;
          for (int i = lane; i < tc; i += 64) {
            const int col = touched[i];
            const unsigned long long s = acc[col];
            acc[col] = 0ull;
            if (pass) {
              nix[p0 + i] = col;
              nw[p0 + i] = (long long)s;
              ksum += (long long)s;
            }
          }
          if (pass) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) ksum += __shfl_xor(ksum, o);
            if (lane == 0) L.kk[c] = ksum;   // (kk of the OLD level is dead: nothing reads it during aggregation)
          } else if (lane == 0) {
            L.u.ag.rep[c] = tc;  // (rep is dead after the partition of the coarse nodes was written)
          }
        }
      }
      __syncthreads();
      if (pass == 0) {
        for (int c = t; c < nn; c += SM_THREADS) L.u.ag.rowptr[c] = L.u.ag.rep[c];
        __syncthreads();
        const int nnz_new = small_scan(L.u.ag.rowptr, nn, L.scan_w);
        if (t == 0) L.u.ag.rowptr[nn] = nnz_new;
        __syncthreads();
      }
    }
    // switch to the coarse level
    for (int c = t; c <= nn; c += SM_THREADS) L.ip[c] = L.u.ag.rowptr[c];
    for (int c = t; c < nn; c += SM_THREADS) L.comm[c] = L.csize[c];
    __threadfence();  // the rows just written are read through the vector cache from now on
    __syncthreads();
    ix = nix;
    wq = nw;
    dst ^= 1;
    n = nn;
  }
  __syncthreads();
  for (int i = t; i < n_in; i += SM_THREADS) a.comm[i] = L.comm[L.tail_of[i]];
  if (t == 0) {
    a.info[0] = levels;
    a.info[1] = tot_moves;
    a.info[2] = tot_merges;
    a.info[3] = n;
  }
}

// ---- quality ---------------------------------------------------------------------------------------
// internal[0] += sum over stored entries inside a community (self loops included)
// G lanes per vertex (16 on the kNN graph itself: four rows in flight per wave instead of one)
template <int G>
__global__ __launch_bounds__(256) void ld_internal_kernel(int n, const int64_t* __restrict__ indptr,
                                                          const int* __restrict__ indices,
                                                          const long long* __restrict__ wq, const int* __restrict__ comm,
                                                          unsigned long long* __restrict__ internal) {
  const int lane = threadIdx.x & 63, sub = threadIdx.x % G;
  long long s = 0;
  for (int v = blockIdx.x * (256 / G) + threadIdx.x / G; v < n; v += gridDim.x * (256 / G)) {
    const int a = comm[v];
    const int64_t beg = indptr[v], end = indptr[v + 1];
    for (int64_t e = beg + sub; e < end; e += 4 * G) {  // (four entries per lane in flight: see ld_within_kernel)
      int u[4], cu[4];
      long long we[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int64_t ee = e + t * G;
        u[t] = ee < end ? indices[ee] : v;
        we[t] = ee < end ? wq[ee] : 0ll;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) cu[t] = comm[u[t]];
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (cu[t] == a) s += we[t];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  // (one global atomic per workgroup: the per-wave version serialised 8192 atomics on one address, 0.2 ms)
  __shared__ long long sh_s[4];
  if (lane == 0) sh_s[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const long long tot = sh_s[0] + sh_s[1] + sh_s[2] + sh_s[3];
    if (tot != 0) atomicAdd(internal, (unsigned long long)tot);
  }
}

// sumsq[0] = sum_c (Ktot[c] / 2m)^2 in a fixed order: SUMSQ_BLOCKS partial sums (fixed ranges, fixed tree), then one
// block adds the partials
constexpr int SUMSQ_BLOCKS = 256;
__device__ __forceinline__ double block_sum_1024(double s, double* sh) {
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  return sh[0];
}
__global__ __launch_bounds__(1024) void ld_sumsq_kernel(int n, const unsigned long long* __restrict__ Ktot,
                                                        double m2, double* __restrict__ part) {
  __shared__ double sh[1024];
  double s = 0.0;
  for (int c = blockIdx.x * 1024 + threadIdx.x; c < n; c += SUMSQ_BLOCKS * 1024) {
    double f = (double)(long long)Ktot[c] / m2;
    s += f * f;
  }
  s = block_sum_1024(s, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(1024) void ld_sumsq_final_kernel(const double* __restrict__ part, double* __restrict__ out) {
  __shared__ double sh[1024];
  double s = block_sum_1024((int)threadIdx.x < SUMSQ_BLOCKS ? part[threadIdx.x] : 0.0, sh);
  if (threadIdx.x == 0) out[0] = s;
}

// ---- final renumbering by decreasing size ------------------------------------------------------------
__global__ __launch_bounds__(1024) void ld_minmember_kernel(int n, const int* __restrict__ memb,
                                                            int* __restrict__ minmember, int* __restrict__ size) {
  __shared__ int keys[BH_SLOTS];
  __shared__ int mn[BH_SLOTS];
  __shared__ int cnt[BH_SLOTS];
  for (int i = threadIdx.x; i < BH_SLOTS; i += 1024) {
    keys[i] = BH_EMPTY;
    mn[i] = 0x7fffffff;
    cnt[i] = 0;
  }
  __syncthreads();
  for (int v = blockIdx.x * 1024 + threadIdx.x; v < n; v += gridDim.x * 1024) {
    const int c = memb[v];
    const int slot = bh_find_slot(keys, c);
    if (slot >= 0) {
      atomicMin(&mn[slot], v);
      atomicAdd(&cnt[slot], 1);
    } else {
      atomicMin(&minmember[c], v);
      atomicAdd(&size[c], 1);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < BH_SLOTS; i += 1024)
    if (keys[i] != BH_EMPTY) {
      atomicMin(&minmember[keys[i]], mn[i]);
      atomicAdd(&size[keys[i]], cnt[i]);
    }
}
// compact list of non-empty communities: key = (~size << 32) | minmember  (ascending = size desc)
__global__ void ld_commkeys_kernel(int n, const int* __restrict__ size, const int* __restrict__ minmember,
                                   const int64_t* __restrict__ pos, unsigned long long* __restrict__ keys,
                                   int* __restrict__ ids) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < n && size[c] > 0) {
    const int64_t p = pos[c];
    keys[p] = ((unsigned long long)(~(unsigned int)size[c]) << 32) | (unsigned int)minmember[c];
    ids[p] = c;
  }
}
__global__ void ld_rank_kernel(int nc, const unsigned long long* __restrict__ keys, const int* __restrict__ ids,
                               int* __restrict__ newlabel /* indexed by old community id */) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nc) return;
  const unsigned long long ki = keys[i];
  int rank = 0;
  for (int j = 0; j < nc; ++j) rank += keys[j] < ki ? 1 : 0;
  newlabel[ids[i]] = rank;
}
__global__ void ld_compact_label_kernel(int nc, const int* __restrict__ ids, int* __restrict__ newlabel) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nc) newlabel[ids[i]] = i;
}
__global__ void ld_relabel_kernel(int n, const int* __restrict__ newlabel, const int* memb, int* out) {
  int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v < n) out[v] = newlabel[memb[v]];  // (out may be memb itself)
}
__global__ void ld_gather_kernel(int n, const int* __restrict__ comm, const int* __restrict__ node_of,
                                 int* __restrict__ out) {
  int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v < n) out[v] = comm[node_of[v]];
}

// ---- host orchestration --------------------------------------------------------------------------
struct LevelGraph {
  int n = 0;
  int max_deg = 0;  // largest row length: levels without hubs skip the block-per-vertex kernels
  int n_gt96 = 0, n_gt192 = 0, n_gt384 = 0;  // rows beyond the quarter- / half- / full-wave tables
  int64_t nnz = 0;
  const int64_t* indptr = nullptr;
  const int* indices = nullptr;
  const long long* wq = nullptr;
  const long long* k = nullptr;
};

struct CoarseBuf {
  int64_t* indptr; int* indices; long long* wq; long long* k;
};

struct LeidenBuffers {
  long long* wq0; long long* k0;
  CoarseBuf cb[2];
  int* comm; int* csize; unsigned long long* Ktot;
  int* cls_lists; int* rlist; int* touched; int* hub_list;
  int* ref; int* target; int* target2; int* refsize; unsigned long long* Kref; unsigned long long* Eref; long long* a_in;
  VertRec* vrec; TargRec* trec;  // packed mirrors of {comm, ref, stamp} and {Kref, Eref, refsize} for the refinement's gathers
  int* flag; int64_t* newid; int64_t* scan_tmp; int* cid; int* rep; int* comm_tmp;
  int* node_of; int* memb; int* memb_work;  // memb: the best partition so far = the input of the next iteration; memb_work: its output
  int* agg_col; long long* agg_w;  // scratch CSR of the coarse-graph build (rows at upper-bound offsets)
  int* mcount; int64_t* moff; int64_t* eoff; int* members; int* mdeg; int* mid_list; int* big_list;
  int64_t* pmoff; int* part_list; int* part_cnt;  // split rows of the coarse-graph build (ld_agg_parts_kernel)
  int* rowcnt; int* cursor;
  int* counters; int* rcounters; unsigned long long* total; double* dscratch;
  unsigned long long* ckeys; int* cids; int* newlabel; int* minmember;
};

// class sub-rounds the SCAMD_LEIDEN_LM_CLASSES / SCAMD_LEIDEN_RF_CLASSES overrides can ask for on a level of any size
static int classes_env(const char* name) {
  const char* e = getenv(name);
  const int v = e ? atoi(e) : 0;
  return (v == 1 || v == 2 || v == 4 || v == 8 || v == 16 || v == 32) ? v : 0;
}
static int carve_classes() {
  return std::max((int)DEF_CLASSES, std::max(classes_env("SCAMD_LEIDEN_LM_CLASSES"), classes_env("SCAMD_LEIDEN_RF_CLASSES")));
}
// member entries per part of a split coarse row / entries from which a row is split (SCAMD_LEIDEN_AGG_SPLIT_CHUNK / _WORK:
// tests push small graphs through the split path; resolved here so that the workspace query and the run agree)
static int64_t agg_split_chunk() {
  const char* e = getenv("SCAMD_LEIDEN_AGG_SPLIT_CHUNK");
  return e ? std::max<int64_t>(64, atoll(e)) : 131072;
}
static int64_t agg_split_work() {
  const char* e = getenv("SCAMD_LEIDEN_AGG_SPLIT_WORK");
  return std::max<int64_t>(agg_split_chunk(), e ? atoll(e) : 262144);
}
static void leiden_carve(Workspace& ws, int64_t n, int64_t nnz, LeidenBuffers* b) {
  const size_t N = (size_t)n, E = (size_t)std::max<int64_t>(nnz, 1);
  b->wq0 = ws.take<long long>(E);
  b->k0 = ws.take<long long>(N);
  for (int i = 0; i < 2; ++i) {
    b->cb[i].indptr = ws.take<int64_t>(N + 1);
    b->cb[i].indices = ws.take<int>(E);
    b->cb[i].wq = ws.take<long long>(E);
    b->cb[i].k = ws.take<long long>(N);
  }
  b->comm = ws.take<int>(N);
  b->csize = ws.take<int>(N);
  b->Ktot = ws.take<unsigned long long>(N);
  // class lists of a sweep (local moving) / of the refinement: lists[c * n_level ..].  Levels above 4096 vertices use
  // DEF_CLASSES lists unless the SCAMD_LEIDEN_*_CLASSES overrides ask for more (resolved here, so that the workspace query
  // and the run agree); only levels of <= 4096 vertices take MAX_CLASSES refinement sub-rounds.  (Round 3 carved
  // MAX_CLASSES x N: 1.28 GB at 10M cells for lists of which 8 were ever used)
  b->cls_lists = ws.take<int>(std::max((size_t)carve_classes() * N, (size_t)MAX_CLASSES * std::min<size_t>(N, 4096)));
  b->rlist = ws.take<int>(N);
  b->hub_list = ws.take<int>(N);
  b->touched = ws.take<int>(N);
  b->ref = ws.take<int>(N);
  b->target = ws.take<int>(N);
  b->target2 = ws.take<int>(N);  // decisions of alternate sub-rounds (ld_requeue_move_kernel)
  b->refsize = ws.take<int>(N);
  b->Kref = ws.take<unsigned long long>(N);
  b->Eref = ws.take<unsigned long long>(N);
  b->a_in = ws.take<long long>(N);
  b->vrec = ws.take<VertRec>(N);
  b->trec = ws.take<TargRec>(N);
  b->flag = ws.take<int>(N);
  b->newid = ws.take<int64_t>(N + 1);
  b->scan_tmp = ws.take<int64_t>((size_t)scan_num_blocks(n) + 2);
  b->cid = ws.take<int>(N);
  b->rep = ws.take<int>(N);
  b->comm_tmp = ws.take<int>(N);
  b->node_of = ws.take<int>(N);
  b->memb = ws.take<int>(N);
  b->memb_work = ws.take<int>(N);
  b->agg_col = ws.take<int>(E);
  b->agg_w = ws.take<long long>(E);
  b->mcount = ws.take<int>(N);
  b->moff = ws.take<int64_t>(N + 1);
  b->eoff = ws.take<int64_t>(N + 1);
  b->members = ws.take<int>(N);
  b->mdeg = ws.take<int>(N);
  b->mid_list = ws.take<int>(N);
  b->big_list = ws.take<int>(N);
  {
    const size_t slots = 3 * E / (size_t)agg_split_chunk() + 16;  // parts + one closing slot per split row
    b->pmoff = ws.take<int64_t>(slots);
    b->part_list = ws.take<int>(slots);
    b->part_cnt = ws.take<int>(slots);
  }
  b->rowcnt = ws.take<int>(N);
  b->cursor = ws.take<int>(N);
  b->counters = ws.take<int>(16);  // [0..7] phase counters, [8..11] row-length statistics of the level being built
  b->rcounters = ws.take<int>(2 * CTR_AREA);  // two counter areas: the sweeps of the local moving alternate between them
  b->total = ws.take<unsigned long long>(4);
  b->dscratch = ws.take<double>(4 + SUMSQ_BLOCKS);
  b->ckeys = ws.take<unsigned long long>(N);
  b->cids = ws.take<int>(N);
  b->newlabel = ws.take<int>(N);
  b->minmember = ws.take<int>(N);
}

#define GRID1(n) dim3((unsigned)ceil_div((n), 256)), dim3(256)
#define GRIDW(n) dim3((unsigned)ceil_div((n), 4)), dim3(256)
#define GRIDK(n) dim3((unsigned)ceil_div((n), 1024)), dim3(1024)
constexpr int HUB_GRID = 512;
constexpr size_t HUB_LDS = (size_t)BHUB_SLOTS * 12;

struct LeidenCtx {
  HostReadbackScope readback_scope;  // (the read-backs' destinations are locals of the functions this context is passed to)
  hipStream_t s;
  LeidenBuffers b;
  double gamma;
  double m2;  // total (quantised) weight = sum of strengths
  // Objective.  Modularity (default): vertex weight = strength, the gain of joining C is w(v, C) - (gamma / 2m) k_v K_C.
  // CPM (igraph `objective_function='CPM'`, src/scanpy/tools/_leiden.py:188-196): vertex weight = 1 (sums on the coarse
  // levels), resolution not normalised: w(v, C) - gamma n_v N_C.  Every kernel takes "the vertex weights" and "g": the two
  // objectives differ only in what the host hands them (weights are 2^32 fixed point: g carries the scale).
  bool cpm = false;
  // CPM vertex weights: ones (nw_scale 1), or the caller's in fixed point (nw_scale = NODE_WEIGHT_SCALE: a power of two, so
  // unit weights handed in as an array give bit-identical gains)
  const float* node_weights = nullptr;
  double nw_scale = 1.0;
  double gscale() const { return cpm ? gamma * WSCALE / (nw_scale * nw_scale) : gamma / m2; }
  double inv_beta = 0.0;  // 1 / (beta * 2^32): randomness of the refinement's merge rule (0 = greedy)
  int iter = 0;           // outer iteration: part of the refinement's noise seed
  unsigned int seed;
  int lm_stop_permille = 20;  // local moving of a level stops once < 2 % of its vertices move in a sweep (first iteration)
  int lm_stop_permille_big = 40;  // ... < 4 % on the levels that run four classes per sweep (lm_classes): 0.9 ms / 11 ms faster on the
                                  // planted / weak 1M graph at the same Q / inside the seeds' spread (profiles/r06ze_lm_stop_with_4_classes.log);
                                  // everywhere, one seed of the 700-cell fixture fell out of the oracle's seed distribution
  int lm_classes = 0;         // class sub-rounds per local-moving sweep (0 = by level size; SCAMD_LEIDEN_LM_CLASSES)
  int rf_classes = 0;         // class sub-rounds of the refinement (0 = by level size; SCAMD_LEIDEN_RF_CLASSES)
  bool small_levels = true;   // levels of <= SMALL_N nodes in one workgroup (SCAMD_LEIDEN_SMALL=0: separate kernels)
  int l0_moves = -1;          // moves of the last iteration's level-0 local moving (0: its input was node optimal)
  int n_levels = 0;           // levels the last iteration went through
  bool polish = true;         // SCAMD_LEIDEN_POLISH=0: no final polish (A/B; the round-4 behaviour)
  bool no_fuse = false;       // SCAMD_LEIDEN_FUSE=0: re-queue and the next sub-round's decisions as separate launches (A/B)
  int small_seq_n = SMALL_SEQ_N;  // ... of which those of <= small_seq_n vertices move one vertex at a time (SCAMD_LEIDEN_SMALL_SEQ)
  // coarse-row build tiers (distinct-neighbour bounds); the env overrides exist so the tests can push small graphs
  // through the workgroup and multi-pass tiers
  int agg_wave_max = WH_MAX_DEG;
  int agg_mid_max = AGG_MID_MAX;
  int agg_pass_keys = AGG_BIG_PASS;
  // workgroups of the two workgroup tiers of the coarse-row builder (each loops over its share of the listed rows): a
  // launch of N such workgroups costs ~0.27 us x N before any work is done (SCAMD_LEIDEN_AGG_MID_GRID / _BIG_GRID)
  int agg_mid_grid = 768;
  int agg_big_grid = HUB_GRID;
  // member entries beyond which a coarse row leaves the wave tier / the 512-thread tier whatever its table size
  // (SCAMD_LEIDEN_AGG_WAVE_WORK / _MID_WORK)
  int agg_wave_work = 2048;
  int agg_mid_work = 65536;
  int hub_try_probes = HUB_TRY_PROBES;  // 0: no optimistic single pass over multi-pass rows (SCAMD_LEIDEN_HUB_TRY_PROBES; tests)
  // sweeps of the local moving that follow a sweep with at most `small_sweep_act` active vertices use `small_sweep_classes`
  // class sub-rounds instead of the level's (0: off) -- see local_moving
  int small_sweep_act = 0, small_sweep_classes = 2;
};

static bool g_leiden_debug = false;  // SCAMD_LEIDEN_DEBUG=1, read at every entry (tools switch it inside one process)
static bool leiden_debug() { return g_leiden_debug; }

// SCAMD_LEIDEN_DEBUG=2: drain the stream after every launch of the class sub-rounds and name it (a device fault then
// surfaces at the launch that caused it)
static bool g_leiden_debug_sync = false;
#define LD_DBG_SYNC(cx, ...)                                  \
  do {                                                        \
    if (g_leiden_debug_sync) {                                \
      (void)hipStreamSynchronize((cx).s);                     \
      fprintf(stderr, "[leiden] done: " __VA_ARGS__);         \
      fputc('\n', stderr);                                    \
      fflush(stderr);                                         \
    }                                                         \
  } while (0)

// one launch of ld_fill_kernel for up to FillArgs::MAX_REGIONS regions (byte counts are multiples of 4)
struct Filler {
  FillArgs a;
  unsigned long long max_words = 0;
  Filler() { a.n = 0; }
  Filler& add(void* p, size_t bytes, unsigned int pat = 0u) {
    if (bytes == 0) return *this;
    a.p[a.n] = static_cast<unsigned int*>(p);
    a.words[a.n] = bytes / 4;
    a.pat[a.n] = pat;
    max_words = std::max<unsigned long long>(max_words, a.words[a.n]);
    ++a.n;
    return *this;
  }
};
static int run_fill(LeidenCtx& cx, const Filler& f) {
  if (f.a.n == 0) return SCAMD_OK;
  const unsigned grid = (unsigned)std::min<unsigned long long>(2048ull, std::max<unsigned long long>(1ull, (f.max_words / 4 + 255) / 256));
  hipLaunchKernelGGL(ld_fill_kernel, dim3(grid), dim3(256), 0, cx.s, f.a);
  SCAMD_LAUNCH_CHECK();
  ++g_ld_stats[15];
  return SCAMD_OK;
}

static int read_counters(LeidenCtx& cx, int* h, int cnt) {
  LD_FETCH(h, cx.b.counters, sizeof(int) * cnt, cx.s);
  LD_SYNC(cx.s);
  return SCAMD_OK;
}

// Ktot / csize of `comm`; `extra`: further regions the caller wants cleared by the same launch
static int compute_totals(LeidenCtx& cx, const LevelGraph& g, const int* comm, Filler extra = Filler()) {
  extra.add(cx.b.Ktot, sizeof(unsigned long long) * g.n).add(cx.b.csize, sizeof(int) * g.n);
  const int rcf = run_fill(cx, extra);
  if (rcf != SCAMD_OK) return rcf;
  hipLaunchKernelGGL(ld_totals_kernel, dim3((unsigned)std::min(REDUCE_GRID, ceil_div(g.n, 1024))), dim3(1024), 0, cx.s, comm, g.k, g.n, cx.b.Ktot, cx.b.csize,
                     cx.b.total + 1);
  SCAMD_LAUNCH_CHECK();
  return SCAMD_OK;
}

// quality of `comm` on level graph g (needs Ktot up to date): modularity, or the CPM objective in the same units
// (b.total[1], the accumulator of the internal weight, was zeroed by the ld_totals_kernel launch of compute_totals)
static int quality(LeidenCtx& cx, const LevelGraph& g, const int* comm, double* q) {
  // (a group walks its vertices one after the other, three dependent loads each: the more groups, the fewer steps of that
  // walk -- and one same-word atomic per workgroup at the end; SCAMD_LEIDEN_QUALITY_GRID: A/B)
  static const int qgrid = [] {
    const char* e = getenv("SCAMD_LEIDEN_QUALITY_GRID");
    return e ? std::max(64, atoi(e)) : 2048;
  }();
  if (g.nnz <= (int64_t)48 * g.n)
    hipLaunchKernelGGL(ld_internal_kernel<16>, dim3((unsigned)std::min(qgrid, ceil_div(g.n, 16))), dim3(256), 0, cx.s, g.n,
                       g.indptr, g.indices, g.wq, comm, cx.b.total + 1);
  else
    hipLaunchKernelGGL(ld_internal_kernel<64>, dim3((unsigned)std::min(qgrid, ceil_div(g.n, 4))), dim3(256), 0, cx.s, g.n,
                       g.indptr, g.indices, g.wq, comm, cx.b.total + 1);
  SCAMD_LAUNCH_CHECK();
  // (CPM: sum of squared community SIZES, unnormalised)
  hipLaunchKernelGGL(ld_sumsq_kernel, dim3(SUMSQ_BLOCKS), dim3(1024), 0, cx.s, g.n, cx.b.Ktot, cx.cpm ? cx.nw_scale : cx.m2, cx.b.dscratch + 4);
  SCAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(ld_sumsq_final_kernel, dim3(1), dim3(1024), 0, cx.s, cx.b.dscratch + 4, cx.b.dscratch);
  SCAMD_LAUNCH_CHECK();
  unsigned long long internal = 0;
  double sumsq = 0;
  LD_FETCH(&internal, cx.b.total + 1, sizeof(internal), cx.s);
  LD_FETCH(&sumsq, cx.b.dscratch, sizeof(double), cx.s);
  LD_SYNC(cx.s);
  *q = cx.cpm ? ((double)(long long)internal - cx.gamma * WSCALE * sumsq) / cx.m2
              : (double)(long long)internal / cx.m2 - cx.gamma * sumsq;
  return SCAMD_OK;
}

// levels whose rows are short on average (the kNN graph itself) take the four-vertices-per-wave kernels
static_assert(WH_SLOTS / 2 * 3 / 4 == 192 && WH_MAX_DEG == 384,
              "ld_degstats_kernel counts the rows beyond the 128- / 256- / 512-slot tables");
// lanes per vertex of the decision kernels for this level: 16 (four vertices per wave, rows <= 96), 32 (two per wave,
// rows <= 192: the first coarse levels, ~64 entries per row) or 64.  SCAMD_LEIDEN_QUAD = 0 / 1 / 2 forces 64 / 16 / 32.
static int level_lanes(const LevelGraph& g) {
  if (const char* e = getenv("SCAMD_LEIDEN_QUAD")) {
    const int v = atoi(e);
    return v == 1 ? 16 : (v == 2 ? 32 : 64);
  }
  if (g.n <= 0) return 64;
  const int64_t avg = g.nnz / g.n;
  return avg <= 40 ? 16 : (avg <= 110 ? 32 : 64);
}
static bool level_is_short_rowed(const LevelGraph& g) { return level_lanes(g) == 16; }

// class sub-rounds per sweep: 8 everywhere.  (4 / 2 on levels below 16384 / 1024 vertices saved ~1 ms of launch latency
// per call, but the fewer the classes the more neighbours move at once: on the 700-cell fixture one seed in ten then
// ended in a worse optimum, Q 0.8101 against the oracle's minimum 0.8120.)
// Round 6, large levels: FOUR.  At 1M cells (five seeds, profiles/r06zb_leiden_classes.log) 8 / 4 / 2 classes take 457 / 413 / 390 ms
// on the weak graph at Q 0.83960 / 0.83969 / 0.84037 (means), 877 / 794 / 781 ms without structure at Q 0.3318 / 0.3327 / 0.3335,
// 26.6 / 25.2 / 26.5 ms on the planted one (same partition): a vertex of a kNN graph with 10^5 and more vertices has two dozen
// neighbours, half of which deciding on the same snapshot costs nothing the next sweep does not repair -- it is the SMALL levels
// where fewer classes lose quality (see rf_classes; with four classes from 65536 vertices on, the 100k weak sample of the bench
// fell below its agreement gate against the oracle: median ARI over seed pairs 0.02 under the oracle's own).  Four from 262144
// vertices on; SCAMD_LEIDEN_LM_BIG_N moves that size (0: eight everywhere).
static int lm_classes(const LeidenCtx& cx, int n) {
  static const int big_n = [] {
    const char* e = getenv("SCAMD_LEIDEN_LM_BIG_N");
    return e ? atoi(e) : 262144;
  }();
  if (cx.lm_classes > 0) return cx.lm_classes;
  // (the iterations after the first move a few hundred vertices of a large level per sweep: TWO classes there -- weak 1M graph 429 ->
  // 418 ms over three seeds at the same Q, the others unchanged; one class: everybody on one snapshot, far slower to settle)
  static const int later_cls = [] {
    const char* e = getenv("SCAMD_LEIDEN_LM_LATER_CLASSES");
    const int v = e ? atoi(e) : 2;
    return (v == 1 || v == 2 || v == 4 || v == 8) ? v : 2;
  }();
  if (big_n > 0 && n >= big_n) return cx.iter >= 1 ? later_cls : 4;
  return DEF_CLASSES;
}
// (refinement: a singleton cannot join a singleton of its OWN class -- with 8 classes an eighth of the targets the
// sequential algorithm would see are excluded, which on small graphs costs quality: 700-cell fixture over 30 seeds, runs
// ending below Q 0.8115: oracle 0, 32 classes 0, 8 classes 2, 2 classes 5.  Small levels therefore take 32 sub-rounds.)
static int rf_classes(const LeidenCtx& cx, int n) {
  if (cx.rf_classes > 0) return cx.rf_classes;
  return n <= 4096 ? MAX_CLASSES : DEF_CLASSES;
}

static int local_moving(LeidenCtx& cx, const LevelGraph& g, int* total_moves) {
  LeidenBuffers& b = cx.b;
  const double gg = cx.gscale();
  *total_moves = 0;
  const size_t n = (size_t)g.n;
  // (one clear launch: totals, re-queue flags, phase counters ([0] moved, [1] blocked (cumulative), [7] error), the counter
  // area of sweep 0 -- every later sweep's area is cleared by the ld_compact_cls_kernel launch of the sweep before it)
  int rc = compute_totals(cx, g, b.comm, Filler().add(b.flag, sizeof(int) * n).add(b.counters, sizeof(int) * 8).add(b.rcounters, sizeof(int) * CTR_AREA));
  if (rc != SCAMD_OK) return rc;
  const int lanes = level_lanes(g);
  const int thr_mid = lanes == 16 ? (int)G16_MAX : (lanes == 32 ? (int)(WH_SLOTS / 2 * 3 / 4) : -1);
  const int n_cls_level = lm_classes(cx, g.n);
  int n_act_prev = g.n;
  int moved_before = 0, quiet = 0, moved_prev2 = 0;
  for (int sweep = 0; sweep < MAX_LM_SWEEPS; ++sweep) {
    const int n_cls = (sweep > 0 && n_act_prev <= cx.small_sweep_act) ? std::min(n_cls_level, cx.small_sweep_classes) : n_cls_level;
    // [0, MAX_CLASSES): class list lengths of the sweep; then one block per sub-round: hub / overflow counts
    int* sw = b.rcounters + (sweep & 1) * CTR_AREA;
    const unsigned int salt = hash32(cx.seed + 0x85EBCA77u * (unsigned int)(sweep + 1) + 0xC2B2AE3Du * (unsigned int)cx.iter);
    hipLaunchKernelGGL(ld_compact_cls_kernel, dim3((unsigned)ceil_div(g.n, 1024)), dim3(1024), 0, cx.s, g.n,
                       sweep == 0 ? (int*)nullptr : b.flag, b.cls_lists, sw, n_cls, salt, g.indptr, thr_mid, (int)WH_MAX_DEG,
                       b.rcounters + ((sweep + 1) & 1) * CTR_AREA, (int)CTR_AREA);
    SCAMD_LAUNCH_CHECK();
    int hc[CTR_AREA], ht[8];  // class list lengths, then per sub-round [CTR_N_MID] / [CTR_N_HUB]: long rows of the class
    LD_FETCH(hc, sw, sizeof(int) * (MAX_CLASSES + CTR_STRIDE * n_cls), cx.s);
    LD_FETCH(ht, b.counters, sizeof(int) * 8, cx.s);
    LD_SYNC(cx.s);
    SCAMD_REQUIRE(ht[7] == 0, SCAMD_EINTERNAL, "leiden: hub table overflow (local moving)");
    int n_act = 0;
    for (int c = 0; c < n_cls; ++c) n_act += hc[c];
    n_act_prev = n_act;
    const int moved_last = ht[0] - moved_before;  // moves of the previous sweep
    moved_before = ht[0];
    *total_moves = ht[0];
    if (leiden_debug()) {
      int tot_mid = 0, tot_hub = 0;
      for (int c = 0; c < n_cls; ++c) tot_mid += hc[MAX_CLASSES + CTR_STRIDE * c + CTR_N_MID], tot_hub += hc[MAX_CLASSES + CTR_STRIDE * c + CTR_N_HUB];
      fprintf(stderr, "[leiden] lm n=%d sweep=%d classes=%d act=%d moved_prev=%d blocked_total=%d long rows: mid %d hub %d\n", g.n, sweep,
              n_cls, n_act, moved_last, ht[1], tot_mid, tot_hub);
    }
    if (n_act == 0) break;
    ++g_ld_stats[8];
    g_ld_sweep_bytes += (double)n_act * (12.0 * (double)g.nnz / (double)std::max(g.n, 1) + 16.0);
    if (sweep > 0) {
      // (with the direction rule on, blocked vertices stay active without anybody moving: two such sweeps end the level)
      quiet = (moved_last == 0) ? quiet + 1 : 0;
      if (quiet >= 2) break;
      // Vertex-by-vertex merging of whole communities is what the coarser levels are for: once fewer than
      // lm_stop_permille/1000 of the level's vertices move in a sweep, go on to refinement + aggregation (the
      // outer iterations repeat until nothing improves, so no move is lost, it is only made at a cheaper level).
      // (first outer iteration only: the later ones polish, and a level of theirs moves few vertices anyway)
      // The other half of the rule: the move counts of a level fall geometrically while vertices settle (749k, 178k,
      // 55k, 19k, 12k at 1M planted cells) and then RISE again for twenty sweeps (15k ... 74k ... 8k) while fragments
      // coalesce one boundary vertex at a time -- exactly the work that is cheaper one level up.  A sweep that does not
      // move at least a tenth fewer vertices than the one before it ends the level (first outer iteration only).
      if (cx.iter == 0 && sweep >= 2) {
        if ((long long)moved_last * 1000 < (long long)g.n * (n_cls_level < DEF_CLASSES ? cx.lm_stop_permille_big : cx.lm_stop_permille)) break;
        if (sweep >= 3 && (long long)moved_last * 10 > (long long)moved_prev2 * 9) break;
      }
      moved_prev2 = moved_last;
    }
    // the direction rule is the termination guarantee only: synchronous sub-rounds without it converge in a few sweeps
    // on every graph tried, with it (round 2's scheme) about half of the wanted moves of a round were blocked
    const int dir_round = sweep >= LM_DIR_AFTER ? sweep : -1;
    // the non-empty classes of the sweep, in order; the main decision kernel of sub-round i + 1 shares a launch with the
    // re-queue of sub-round i (ld_requeue_move_kernel), the decisions alternate between two buffers
    int cls[MAX_CLASSES], ncl = 0;
    for (int c = 0; c < n_cls; ++c)
      if (hc[c] > 0) cls[ncl++] = c;
    int* tgt[2] = {b.target, b.target2};
    auto move_args = [&](int c, int* decision) {
      return MoveArgs{hc[c], b.cls_lists + (size_t)c * n, nullptr, nullptr, g.indptr, g.indices, g.wq, g.k, b.comm, b.Ktot,
                      b.csize, gg, dir_round, cx.seed, decision, b.mid_list, b.hub_list, sw + MAX_CLASSES + CTR_STRIDE * c};
    };
    auto move_blocks = [&](int cnt) { return lanes == 16 ? ceil_div(cnt, 16) : (lanes == 32 ? ceil_div(cnt, 8) : ceil_div(cnt, 4)); };
    for (int i = 0; i < ncl; ++i) {
      const int c = cls[i];
      const int cnt = hc[c];
      const int* list = b.cls_lists + (size_t)c * n;
      int* ctr = sw + MAX_CLASSES + CTR_STRIDE * c;
      int* tg = tgt[i & 1];
      // long rows among THIS sub-round's vertices (counted by ld_compact_cls_kernel): no launch for an empty tier
      const int n_mid = hc[MAX_CLASSES + CTR_STRIDE * c + CTR_N_MID], n_hub = hc[MAX_CLASSES + CTR_STRIDE * c + CTR_N_HUB];
      if (i == 0) {  // (later sub-rounds: decided in the launch that re-queued the sub-round before)
        const MoveArgs ma = move_args(c, tg);
        if (lanes == 32)
          hipLaunchKernelGGL(ld_move_kernel<32>, dim3((unsigned)move_blocks(cnt)), dim3(256), 0, cx.s, ma.n_act, ma.list, ma.sub_list,
                             ma.sub_count, ma.indptr, ma.indices, ma.wq, ma.k, ma.comm, ma.Ktot, ma.csize, ma.g, ma.round, ma.seed,
                             ma.decision, ma.ovf_list, ma.hub_list, ma.counters);
        else if (lanes == 16)
          hipLaunchKernelGGL(ld_move_kernel<16>, dim3((unsigned)move_blocks(cnt)), dim3(256), 0, cx.s, ma.n_act, ma.list, ma.sub_list,
                             ma.sub_count, ma.indptr, ma.indices, ma.wq, ma.k, ma.comm, ma.Ktot, ma.csize, ma.g, ma.round, ma.seed,
                             ma.decision, ma.ovf_list, ma.hub_list, ma.counters);
        else
          hipLaunchKernelGGL(ld_move_kernel<64>, dim3((unsigned)move_blocks(cnt)), dim3(256), 0, cx.s, ma.n_act, ma.list, ma.sub_list,
                             ma.sub_count, ma.indptr, ma.indices, ma.wq, ma.k, ma.comm, ma.Ktot, ma.csize, ma.g, ma.round, ma.seed,
                             ma.decision, ma.ovf_list, ma.hub_list, ma.counters);
        SCAMD_LAUNCH_CHECK();
      }
      if (lanes != 64 && n_mid > 0) {
        hipLaunchKernelGGL(ld_move_kernel<64>, dim3((unsigned)std::min(2048, ceil_div(n_mid, 4))), dim3(256), 0, cx.s,
                           cnt, list, (const int*)b.mid_list, (const int*)(ctr + 5), g.indptr, g.indices, g.wq, g.k,
                           b.comm, b.Ktot, b.csize, gg, dir_round, cx.seed, tg, b.mid_list, b.hub_list, ctr);
        SCAMD_LAUNCH_CHECK();
      }
      if (n_hub > 0) {
        hipLaunchKernelGGL(ld_move_hub_kernel, dim3((unsigned)std::min(HUB_GRID, n_hub)), dim3(HUB_THREADS), HUB_LDS, cx.s, b.hub_list,
                           ctr, list, g.indptr, g.indices, g.wq, g.k, b.comm, b.Ktot, b.csize, gg, dir_round, cx.seed,
                           tg, b.counters + 7, cx.hub_try_probes);
        SCAMD_LAUNCH_CHECK();
      }
      hipLaunchKernelGGL(ld_apply_kernel, GRID1(cnt), 0, cx.s, cnt, list, tg, g.k, b.comm, b.Ktot, b.csize, b.flag,
                         b.counters);
      SCAMD_LAUNCH_CHECK();
      if (i + 1 < ncl && !cx.no_fuse) {
        const int cn = cls[i + 1];
        const MoveArgs ma = move_args(cn, tgt[(i + 1) & 1]);
        const int nb_rq = ceil_div(cnt, 256);
        const dim3 grid((unsigned)(nb_rq + move_blocks(hc[cn])));
        if (lanes == 32)
          hipLaunchKernelGGL(ld_requeue_move_kernel<32>, grid, dim3(256), 0, cx.s, nb_rq, cnt, list, (const int*)tg, g.indptr,
                             g.indices, (const int*)b.comm, b.flag, ma);
        else if (lanes == 16)
          hipLaunchKernelGGL(ld_requeue_move_kernel<16>, grid, dim3(256), 0, cx.s, nb_rq, cnt, list, (const int*)tg, g.indptr,
                             g.indices, (const int*)b.comm, b.flag, ma);
        else
          hipLaunchKernelGGL(ld_requeue_move_kernel<64>, grid, dim3(256), 0, cx.s, nb_rq, cnt, list, (const int*)tg, g.indptr,
                             g.indices, (const int*)b.comm, b.flag, ma);
        SCAMD_LAUNCH_CHECK();
      } else {
        hipLaunchKernelGGL(ld_requeue_kernel, GRID1(cnt), 0, cx.s, cnt, list, tg, g.indptr, g.indices, b.comm, b.flag);
        SCAMD_LAUNCH_CHECK();
        if (i + 1 < ncl) {  // (SCAMD_LEIDEN_FUSE=0: the next sub-round's decisions as a launch of their own)
          const MoveArgs ma = move_args(cls[i + 1], tgt[(i + 1) & 1]);
          const unsigned nbm = (unsigned)move_blocks(hc[cls[i + 1]]);
          if (lanes == 32)
            hipLaunchKernelGGL(ld_move_kernel<32>, dim3(nbm), dim3(256), 0, cx.s, ma.n_act, ma.list, ma.sub_list, ma.sub_count,
                               ma.indptr, ma.indices, ma.wq, ma.k, ma.comm, ma.Ktot, ma.csize, ma.g, ma.round, ma.seed, ma.decision,
                               ma.ovf_list, ma.hub_list, ma.counters);
          else if (lanes == 16)
            hipLaunchKernelGGL(ld_move_kernel<16>, dim3(nbm), dim3(256), 0, cx.s, ma.n_act, ma.list, ma.sub_list, ma.sub_count,
                               ma.indptr, ma.indices, ma.wq, ma.k, ma.comm, ma.Ktot, ma.csize, ma.g, ma.round, ma.seed, ma.decision,
                               ma.ovf_list, ma.hub_list, ma.counters);
          else
            hipLaunchKernelGGL(ld_move_kernel<64>, dim3(nbm), dim3(256), 0, cx.s, ma.n_act, ma.list, ma.sub_list, ma.sub_count,
                               ma.indptr, ma.indices, ma.wq, ma.k, ma.comm, ma.Ktot, ma.csize, ma.g, ma.round, ma.seed, ma.decision,
                               ma.ovf_list, ma.hub_list, ma.counters);
          SCAMD_LAUNCH_CHECK();
        }
      }
    }
  }
  return SCAMD_OK;
}

// Final polish of the level-0 partition in b.memb (see ld_polish_lock_kernel): full sweeps of lock-arbitrated moves until
// a sweep over ALL vertices finds no improving move -- node optimality by construction.  b.memb is updated in place;
// stats[0] = full sweeps, [1] = rounds, [2] = moves, [3] = communities split off (split_disconnected).  Needs b.Kref (the
// refinement's scratch) as the lock table.
// b.comm: every connected component of a community becomes its own community (ids = smallest member); totals recomputed.
// *n_split = components - communities (0: nothing changed).  Scratch: b.cid, b.counters[2..4].
static int split_disconnected(LeidenCtx& cx, const LevelGraph& g, int* n_split) {
  LeidenBuffers& b = cx.b;
  *n_split = 0;
  hipLaunchKernelGGL(ld_iota_kernel, GRID1(g.n), 0, cx.s, b.cid, g.n);
  SCAMD_LAUNCH_CHECK();
  for (int it = 0; it < g.n; ++it) {
    {
      const int rcf = run_fill(cx, Filler().add(b.counters + 2, sizeof(int) * 3));
      if (rcf != SCAMD_OK) return rcf;
    }
    // (several propagation steps per host round trip: the flag only says whether any of them changed something)
    for (int rep = 0; rep < 4; ++rep) {
      hipLaunchKernelGGL(ld_cc_prop_kernel, dim3((unsigned)ceil_div(g.n, 16)), dim3(256), 0, cx.s, g.n, g.indptr, g.indices,
                         (const int*)b.comm, b.cid, b.counters + 2);
      SCAMD_LAUNCH_CHECK();
    }
    int changed = 0;
    LD_FETCH(&changed, b.counters + 2, sizeof(int), cx.s);
    LD_SYNC(cx.s);
    if (!changed) break;
  }
  hipLaunchKernelGGL(ld_cc_count_kernel, GRID1(g.n), 0, cx.s, g.n, (const int*)b.cid, (const int*)b.csize, b.counters + 3);
  SCAMD_LAUNCH_CHECK();
  int cnt[2] = {0, 0};
  LD_FETCH(cnt, b.counters + 3, sizeof(cnt), cx.s);
  LD_SYNC(cx.s);
  *n_split = cnt[0] - cnt[1];
  if (leiden_debug()) fprintf(stderr, "[leiden] components %d, communities %d\n", cnt[0], cnt[1]);
  if (*n_split > 0) {
    std::swap(b.comm, b.cid);  // (the component labels ARE the new partition; b.cid is scratch)
    return compute_totals(cx, g, b.comm);
  }
  return SCAMD_OK;
}

constexpr int MAX_POLISH_ROUNDS = 1 << 16;
constexpr int MAX_POLISH_PASSES = 6;  // polish -> verifying iteration -> polish ... (each accepted pass raises Q)
static int polish_level0(LeidenCtx& cx, const LevelGraph& g, int* stats) {
  LeidenBuffers& b = cx.b;
  const double gg = cx.gscale();
  const size_t n = (size_t)g.n;
  stats[0] = stats[1] = stats[2] = stats[3] = 0;
  // the polish works on b.comm in place: the partition changes buffers instead of being copied (b.comm <-> b.memb now and
  // back at the end; what b.comm held is scratch between iterations)
  std::swap(b.comm, b.memb);
  unsigned long long* lock = b.Kref;
  int rc = compute_totals(cx, g, b.comm, Filler().add(lock, sizeof(unsigned long long) * n).add(b.flag, sizeof(int) * n).add(b.counters, sizeof(int) * 8)
                                              .add(b.rcounters, sizeof(int) * CTR_AREA));
  if (rc != SCAMD_OK) return rc;  // (cx.b lives for this call only: an error return need not swap back)
  const int lanes = level_lanes(g);
  const int thr_mid = lanes == 16 ? (int)G16_MAX : (lanes == 32 ? (int)(WH_SLOTS / 2 * 3 / 4) : -1);
  unsigned int round = 0, area = 0;  // (the rounds alternate between the two counter areas, as the sweeps of the local moving do)
  int moved_before = 0, moved_at_full = 0;
  int checked_at = 0;  // moves + splits when the communities were last known to be connected (the input is: an iteration's result)
  bool full = true;  // the next round decides for every vertex (else: for the flagged ones)
  for (;;) {
    int* sw = b.rcounters + (area & 1u) * CTR_AREA;
    int* ctr = sw + MAX_CLASSES;  // counter block of the one class
    hipLaunchKernelGGL(ld_compact_cls_kernel, dim3((unsigned)ceil_div(g.n, 1024)), dim3(1024), 0, cx.s, g.n,
                       full ? (int*)nullptr : b.flag, b.cls_lists, sw, 1, 0u, g.indptr, thr_mid, (int)WH_MAX_DEG,
                       b.rcounters + ((area + 1u) & 1u) * CTR_AREA, (int)(MAX_CLASSES + CTR_STRIDE));
    SCAMD_LAUNCH_CHECK();
    ++area;
    int hc[MAX_CLASSES + CTR_STRIDE], ht[8];
    LD_FETCH(hc, sw, sizeof(hc), cx.s);
    LD_FETCH(ht, b.counters, sizeof(ht), cx.s);
    LD_SYNC(cx.s);
    SCAMD_REQUIRE(ht[7] == 0, SCAMD_EINTERNAL, "leiden: hub table overflow (polish)");
    const int cnt = hc[0];
    const int moved_last = ht[0] - moved_before;  // moves of the previous round
    moved_before = ht[0];
    stats[2] = ht[0];
    if (leiden_debug())
      fprintf(stderr, "[leiden] polish round=%u %s act=%d moved_prev=%d lost_total=%d\n", round, full ? "full" : "flagged", cnt,
              moved_last, ht[1]);
    if (round >= MAX_POLISH_ROUNDS) {  // (every round moves at least one vertex and raises Q: a cap, not a rule)
      g_ld_stats[12] = 1;  // ... but one that leaves node optimality unproven: reported (scamd_leiden_last_stats, tl.leiden warns)
      break;
    }
    if (cnt == 0) {
      // nobody is flagged any more.  If nothing moved since the last sweep over ALL vertices, that sweep was the proof
      // of node optimality; otherwise another full sweep has to give it.
      if (ht[0] == moved_at_full) {
        // ... and if anything has moved since the communities were last known to be connected: split what a departure
        // cut in two (the parts are communities of their own then, and the proof has to be given again)
        if (ht[0] + stats[3] == checked_at) break;
        int n_split = 0;
        rc = split_disconnected(cx, g, &n_split);
        if (rc != SCAMD_OK) return rc;
        stats[3] += n_split;
        checked_at = ht[0] + stats[3];
        if (n_split == 0) break;
      }
      full = true;
      continue;
    }
    if (full) {
      ++stats[0];
      moved_at_full = ht[0];
    }
    full = false;
    ++round;
    ++stats[1];
    const int* list = b.cls_lists;
    const int n_mid = hc[MAX_CLASSES + CTR_N_MID], n_hub = hc[MAX_CLASSES + CTR_N_HUB];
    const unsigned nbm = (unsigned)(lanes == 16 ? ceil_div(cnt, 16) : (lanes == 32 ? ceil_div(cnt, 8) : ceil_div(cnt, 4)));
    if (lanes == 32)
      hipLaunchKernelGGL(ld_move_kernel<32>, dim3(nbm), dim3(256), 0, cx.s, cnt, list, (const int*)nullptr, (const int*)nullptr,
                         g.indptr, g.indices, g.wq, g.k, b.comm, b.Ktot, b.csize, gg, -1, cx.seed, b.target, b.mid_list, b.hub_list, ctr);
    else if (lanes == 16)
      hipLaunchKernelGGL(ld_move_kernel<16>, dim3(nbm), dim3(256), 0, cx.s, cnt, list, (const int*)nullptr, (const int*)nullptr,
                         g.indptr, g.indices, g.wq, g.k, b.comm, b.Ktot, b.csize, gg, -1, cx.seed, b.target, b.mid_list, b.hub_list, ctr);
    else
      hipLaunchKernelGGL(ld_move_kernel<64>, dim3(nbm), dim3(256), 0, cx.s, cnt, list, (const int*)nullptr, (const int*)nullptr,
                         g.indptr, g.indices, g.wq, g.k, b.comm, b.Ktot, b.csize, gg, -1, cx.seed, b.target, b.mid_list, b.hub_list, ctr);
    SCAMD_LAUNCH_CHECK();
    if (lanes != 64 && n_mid > 0) {
      hipLaunchKernelGGL(ld_move_kernel<64>, dim3((unsigned)std::min(2048, ceil_div(n_mid, 4))), dim3(256), 0, cx.s, cnt, list,
                         (const int*)b.mid_list, (const int*)(ctr + 5), g.indptr, g.indices, g.wq, g.k, b.comm, b.Ktot, b.csize, gg,
                         -1, cx.seed, b.target, b.mid_list, b.hub_list, ctr);
      SCAMD_LAUNCH_CHECK();
    }
    if (n_hub > 0) {
      hipLaunchKernelGGL(ld_move_hub_kernel, dim3((unsigned)std::min(HUB_GRID, n_hub)), dim3(HUB_THREADS), HUB_LDS, cx.s, b.hub_list,
                         ctr, list, g.indptr, g.indices, g.wq, g.k, b.comm, b.Ktot, b.csize, gg, -1, cx.seed, b.target,
                         b.counters + 7, cx.hub_try_probes);
      SCAMD_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(ld_polish_lock_kernel, GRID1(cnt), 0, cx.s, cnt, list, (const int*)b.target, (const int*)b.comm, lock, round);
    SCAMD_LAUNCH_CHECK();
    hipLaunchKernelGGL(ld_polish_apply_kernel, GRID1(cnt), 0, cx.s, cnt, list, b.target, g.k, b.comm, b.Ktot, b.csize,
                       (const unsigned long long*)lock, round, b.flag, b.counters);
    SCAMD_LAUNCH_CHECK();
    hipLaunchKernelGGL(ld_requeue_kernel, GRID1(cnt), 0, cx.s, cnt, list, (const int*)b.target, g.indptr, g.indices,
                       (const int*)b.comm, b.flag);
    SCAMD_LAUNCH_CHECK();
  }
  std::swap(b.comm, b.memb);  // the polished partition (or, when nothing moved, the one that came in) is b.memb again
  return SCAMD_OK;
}

static int refinement(LeidenCtx& cx, const LevelGraph& g, int* n_merged) {
  LeidenBuffers& b = cx.b;
  const double gg = cx.gscale();
  // fresh merge noise and fresh classes every outer iteration
  const unsigned int rseed = cx.seed + 0x9E3779B9u * (unsigned int)cx.iter;
  const unsigned int salt = hash32(rseed ^ 0x5bd1e995u);
  const size_t n = (size_t)g.n;
  const bool quad = level_is_short_rowed(g);
  if (quad)
    hipLaunchKernelGGL(ld_within_kernel<16>, dim3((unsigned)ceil_div(g.n, 16)), dim3(256), 0, cx.s, g.n, g.indptr,
                       g.indices, g.wq, b.comm, b.a_in);
  else
    hipLaunchKernelGGL(ld_within_kernel<64>, GRIDW(g.n), 0, cx.s, g.n, g.indptr, g.indices, g.wq, b.comm, b.a_in);
  SCAMD_LAUNCH_CHECK();
  const int n_cls = rf_classes(cx, g.n);
  int* rc0 = b.rcounters;  // [0, MAX_CLASSES): class list lengths; then per sub-round c: [0] joiners, [4] hubs, [5] overflow
  // (also clears rc0, the phase counters and the join stamps b.touched: three memsets until round 6)
  hipLaunchKernelGGL(ld_refine_init_kernel, GRID1(g.n), 0, cx.s, g.n, g.k, b.a_in, b.comm, b.ref, b.refsize, b.Kref,
                     b.Eref, b.vrec, b.trec, b.touched, rc0, (int)(MAX_CLASSES + CTR_STRIDE * n_cls), b.counters);
  SCAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(ld_refine_candidates_kernel, dim3((unsigned)ceil_div(g.n, 1024)), dim3(1024), 0, cx.s, g.n, g.k,
                     b.comm, b.Ktot, b.a_in, gg, b.cls_lists, rc0, n_cls, salt, g.indptr,
                     quad ? (int)G16_MAX : -1, (int)WH_MAX_DEG);
  SCAMD_LAUNCH_CHECK();
  LD_DBG_SYNC(cx, "rf candidates n=%d classes=%d", g.n, n_cls);
  int hc[CTR_AREA];  // class list lengths, then per sub-round [CTR_N_MID] / [CTR_N_HUB] (ld_refine_candidates_kernel)
  LD_FETCH(hc, rc0, sizeof(int) * (MAX_CLASSES + CTR_STRIDE * n_cls), cx.s);
  LD_SYNC(cx.s);
  *n_merged = 0;
  // ONE sweep of n_cls sub-rounds, no host round trip in between: every candidate is considered exactly once (see
  // ld_refine_propose_kernel); the joiner counts stay on the device until the end
  for (int c = 0; c < n_cls; ++c) {
    const int cnt = hc[c];
    if (cnt == 0) continue;
    const int* list = b.cls_lists + (size_t)c * n;
    int* ctr = rc0 + MAX_CLASSES + CTR_STRIDE * c;
    const int n_mid = hc[MAX_CLASSES + CTR_STRIDE * c + CTR_N_MID], n_hub = hc[MAX_CLASSES + CTR_STRIDE * c + CTR_N_HUB];
    const unsigned wgrid = (unsigned)std::min(32768, ceil_div(cnt, 4));
    const unsigned tgrid = (unsigned)std::min(32768, ceil_div(cnt, 256));
    const unsigned qgrid = (unsigned)std::min(32768, ceil_div(cnt, 16));
    if (quad) {
      hipLaunchKernelGGL(ld_refine_propose_kernel<16>, dim3(qgrid), dim3(256), 0, cx.s, list, (const int*)nullptr,
                         (const int*)nullptr, g.indptr, g.indices, g.wq, g.k, b.comm, b.Ktot, b.ref, b.refsize, b.Kref,
                         b.Eref, gg, cx.inv_beta, c, n_cls, salt, rseed, b.target, b.mid_list, b.hub_list, ctr, cnt,
                         b.vrec, b.trec);
      SCAMD_LAUNCH_CHECK();
      LD_DBG_SYNC(cx, "rf propose<16> n=%d class=%d cnt=%d", g.n, c, cnt);
      if (n_mid > 0) {
        hipLaunchKernelGGL(ld_refine_propose_kernel<64>, dim3((unsigned)std::min(2048, ceil_div(n_mid, 4))), dim3(256), 0, cx.s, list,
                           (const int*)b.mid_list, (const int*)(ctr + 5), g.indptr, g.indices, g.wq, g.k, b.comm, b.Ktot,
                           b.ref, b.refsize, b.Kref, b.Eref, gg, cx.inv_beta, c, n_cls, salt, rseed, b.target, b.mid_list,
                           b.hub_list, ctr, cnt, b.vrec, b.trec);
        SCAMD_LAUNCH_CHECK();
        LD_DBG_SYNC(cx, "rf propose<64> overflow n=%d class=%d", g.n, c);
      }
    } else {
      hipLaunchKernelGGL(ld_refine_propose_kernel<64>, dim3(wgrid), dim3(256), 0, cx.s, list, (const int*)nullptr,
                         (const int*)nullptr, g.indptr, g.indices, g.wq, g.k, b.comm, b.Ktot, b.ref, b.refsize, b.Kref,
                         b.Eref, gg, cx.inv_beta, c, n_cls, salt, rseed, b.target, b.mid_list, b.hub_list, ctr, cnt,
                         b.vrec, b.trec);
      SCAMD_LAUNCH_CHECK();
      LD_DBG_SYNC(cx, "rf propose<64> n=%d class=%d cnt=%d", g.n, c, cnt);
    }
    if (n_hub > 0) {
      hipLaunchKernelGGL(ld_refine_propose_hub_kernel, dim3((unsigned)std::min(HUB_GRID, n_hub)), dim3(HUB_THREADS), HUB_LDS, cx.s,
                         b.hub_list, ctr, g.indptr, g.indices, g.wq, g.k, b.comm, b.Ktot, b.ref, b.refsize, b.Kref, b.Eref,
                         gg, cx.inv_beta, c, n_cls, salt, rseed, b.target, b.counters + 7, b.vrec, b.trec, cx.hub_try_probes);
      SCAMD_LAUNCH_CHECK();
      LD_DBG_SYNC(cx, "rf propose hub n=%d class=%d", g.n, c);
    }
    hipLaunchKernelGGL(ld_refine_apply_kernel, dim3(tgrid), dim3(256), 0, cx.s, cnt, list, b.target, g.k, b.ref,
                       b.refsize, b.Kref, b.Eref, b.touched, c, b.rlist, ctr, b.vrec, b.trec);
    SCAMD_LAUNCH_CHECK();
    LD_DBG_SYNC(cx, "rf apply n=%d class=%d", g.n, c);
    if (quad)
      hipLaunchKernelGGL(ld_refine_cut_update_kernel<16>, dim3(qgrid), dim3(256), 0, cx.s, cnt, b.rlist, g.indptr,
                         g.indices, g.wq, b.comm, b.ref, b.touched, b.a_in, c, b.Eref, (const int*)ctr,
                         b.vrec, b.trec);
    else
      hipLaunchKernelGGL(ld_refine_cut_update_kernel<64>, dim3(wgrid), dim3(256), 0, cx.s, cnt, b.rlist, g.indptr,
                         g.indices, g.wq, b.comm, b.ref, b.touched, b.a_in, c, b.Eref, (const int*)ctr,
                         b.vrec, b.trec);
    SCAMD_LAUNCH_CHECK();
    LD_DBG_SYNC(cx, "rf cut update n=%d class=%d", g.n, c);
  }
  int hr[CTR_AREA], herr = 0;
  LD_FETCH(hr, rc0, sizeof(int) * (MAX_CLASSES + CTR_STRIDE * n_cls), cx.s);
  LD_FETCH(&herr, b.counters + 7, sizeof(int), cx.s);
  LD_SYNC(cx.s);
  SCAMD_REQUIRE(herr == 0, SCAMD_EINTERNAL, "leiden: hub table overflow (refinement)");
  for (int c = 0; c < n_cls; ++c) {
    *n_merged += hr[MAX_CLASSES + CTR_STRIDE * c];
    if (leiden_debug())
      fprintf(stderr, "[leiden] rf n=%d class=%d/%d cand=%d merges=%d\n", g.n, c, n_cls, hc[c], hr[MAX_CLASSES + CTR_STRIDE * c]);
  }
  return SCAMD_OK;
}

// builds the coarse graph of `g` under b.ref into cb[dst]; updates b.comm (coarse phase-1 partition)
// and b.node_of.  Returns the new node count in *n_new (== g.n means nothing merged: no graph built).
static int aggregate(LeidenCtx& cx, const LevelGraph& g, int n_orig, int dst, LevelGraph* out, int* n_new) {
  LeidenBuffers& b = cx.b;
  hipLaunchKernelGGL(ld_flag_kernel, GRID1(g.n), 0, cx.s, g.n, b.refsize, b.flag, b.rep);  // (+ b.rep = the atomicMin sentinel)
  SCAMD_LAUNCH_CHECK();
  int rc = exclusive_scan_i32_i64(b.flag, g.n, b.newid, b.scan_tmp, cx.s);
  if (rc != SCAMD_OK) return rc;
  int64_t nn = 0;
  LD_FETCH(&nn, b.newid + g.n, sizeof(int64_t), cx.s);
  LD_SYNC(cx.s);
  *n_new = (int)nn;
  if (nn == g.n) return SCAMD_OK;
  hipLaunchKernelGGL(ld_coarse_ids_kernel, GRIDK(g.n), 0, cx.s, g.n, b.ref, b.newid, b.comm, b.cid, b.rep);
  SCAMD_LAUNCH_CHECK();
  // (+ the cursors of the member scatter and the phase counters cleared: two memsets until round 6)
  hipLaunchKernelGGL(ld_coarse_comm_kernel, GRID1(g.n), 0, cx.s, g.n, b.cid, b.comm, b.rep, b.comm_tmp, (int)nn, b.cursor, b.counters);
  SCAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(ld_remap_kernel, GRID1(n_orig), 0, cx.s, n_orig, b.cid, b.node_of);
  SCAMD_LAUNCH_CHECK();
  // group the members of every coarse node, then combine their rows per coarse node in LDS
  hipLaunchKernelGGL(ld_agg_mcount_kernel, GRID1(g.n), 0, cx.s, g.n, b.refsize, b.newid, b.mcount);
  SCAMD_LAUNCH_CHECK();
  rc = exclusive_scan_i32_i64(b.mcount, nn, b.moff, b.scan_tmp, cx.s);
  if (rc != SCAMD_OK) return rc;
  hipLaunchKernelGGL(ld_agg_scatter_kernel, GRID1(g.n), 0, cx.s, g.n, b.cid, b.moff, b.cursor, g.indptr, b.members,
                     b.mdeg);
  SCAMD_LAUNCH_CHECK();
  rc = exclusive_scan_i32_i64(b.mdeg, g.n, b.eoff, b.scan_tmp, cx.s);
  if (rc != SCAMD_OK) return rc;
  const int inn = (int)nn;
  hipLaunchKernelGGL(ld_agg_wave_kernel, GRIDW(inn), 0, cx.s, inn, b.moff, b.eoff, b.members, g.indptr, g.indices, g.wq,
                     b.cid, b.agg_col, b.agg_w, b.rowcnt, b.mid_list, b.big_list, b.counters, cx.agg_wave_max,
                     cx.agg_mid_max, cx.agg_wave_work, cx.agg_mid_work, b.hub_list, agg_split_work());
  SCAMD_LAUNCH_CHECK();
  // workgroup tiers: 512 threads on the 48 KB tables (3 per CU), 1024 threads on the 96 KB table (1 per CU).  Their list
  // lengths are read back first: an empty launch of these shapes costs 40 / 140 us (768 x 512 / 512 x 1024 threads with
  // 48 / 96 KB of LDS each), a host round trip 15 -- and most levels of a clustered graph have no such rows at all.
  int htier[3] = {0, 0, 0};  // rows of the 512-thread tier, of the 1024-thread tier, split rows
  LD_FETCH(htier, b.counters + 4, sizeof(int) * 3, cx.s);
  LD_SYNC(cx.s);
  if (leiden_debug() && (htier[0] || htier[1]))
    fprintf(stderr, "[leiden] aggregate n=%d -> %d: %d rows through the workgroup tier, %d through the 8192-slot tier\n", g.n, inn,
            htier[0], htier[1]);
  if (htier[0] > 0) {
    hipLaunchKernelGGL((ld_agg_block_kernel<AGG_MID_SLOTS, 512>), dim3((unsigned)std::min(cx.agg_mid_grid, htier[0])), dim3(512),
                       (size_t)AGG_MID_SLOTS * 12, cx.s, b.mid_list, b.counters + 4, inn, b.moff, b.eoff, b.members, g.indptr,
                       g.indices, g.wq, b.cid, b.agg_col, b.agg_w, b.rowcnt, b.counters + 7, AGG_MID_MAX, cx.hub_try_probes);
    SCAMD_LAUNCH_CHECK();
  }
  if (htier[1] > 0) {
    hipLaunchKernelGGL((ld_agg_block_kernel<BHUB_SLOTS, 1024>), dim3((unsigned)std::min(cx.agg_big_grid, htier[1])), dim3(1024),
                       HUB_LDS, cx.s, b.big_list, b.counters + 5, inn, b.moff, b.eoff, b.members, g.indptr, g.indices, g.wq,
                       b.cid, b.agg_col, b.agg_w, b.rowcnt, b.counters + 7, cx.agg_pass_keys, cx.hub_try_probes);
    SCAMD_LAUNCH_CHECK();
  }
  if (htier[2] > 0) {
    // split rows: parts -> built as pseudo rows by the 1024-thread builder -> merged (ld_agg_parts_kernel).  The list of
    // split rows borrows b.hub_list, their first part / part count b.rlist / b.touched (refinement scratch, idle here)
    const int64_t chunk = agg_split_chunk();
    if (leiden_debug()) fprintf(stderr, "[leiden] aggregate n=%d -> %d: %d rows split into parts of %lld entries\n", g.n, inn, htier[2], (long long)chunk);
    hipLaunchKernelGGL(ld_agg_parts_kernel, GRID1(htier[2]), 0, cx.s, (const int*)b.hub_list, b.counters, b.moff, b.eoff, chunk,
                       b.pmoff, b.part_list, b.rlist, b.touched);
    SCAMD_LAUNCH_CHECK();
    const int64_t parts_bound = (int64_t)htier[2] + g.nnz / chunk + 1;
    hipLaunchKernelGGL((ld_agg_block_kernel<BHUB_SLOTS, 1024>), dim3((unsigned)std::min<int64_t>(cx.agg_big_grid, parts_bound)), dim3(1024),
                       HUB_LDS, cx.s, (const int*)b.part_list, (const int*)(b.counters + 3), inn, (const int64_t*)b.pmoff, b.eoff,
                       b.members, g.indptr, g.indices, g.wq, b.cid, b.agg_col, b.agg_w, b.part_cnt, b.counters + 7,
                       cx.agg_pass_keys, cx.hub_try_probes);
    SCAMD_LAUNCH_CHECK();
    hipLaunchKernelGGL(ld_agg_merge_kernel, dim3((unsigned)std::min(HUB_GRID, htier[2])), dim3(1024), HUB_LDS, cx.s,
                       (const int*)b.hub_list, (const int*)b.counters, b.moff, b.eoff, (const int64_t*)b.pmoff, (const int*)b.part_cnt,
                       (const int*)b.rlist, (const int*)b.touched, b.agg_col, b.agg_w, b.rowcnt, b.counters + 7);
    SCAMD_LAUNCH_CHECK();
  }
  CoarseBuf& cb = b.cb[dst];
  rc = exclusive_scan_i32_i64(b.rowcnt, nn, cb.indptr, b.scan_tmp, cx.s);
  if (rc != SCAMD_OK) return rc;
  // the order of a row's entries is arbitrary: nothing downstream depends on it (all sums are integer, every
  // choice is an argmax under a total order)
  hipLaunchKernelGGL(ld_agg_compact_kernel, GRIDW(inn), 0, cx.s, inn, b.moff, b.eoff, cb.indptr, b.agg_col, b.agg_w,
                     cb.indices, cb.wq, b.counters + 8);  // (+ clears the row-length statistics ld_degstats_kernel adds to)
  SCAMD_LAUNCH_CHECK();
  int64_t nnz_new = 0;
  int dstat[4] = {0, 0, 0, 0};
  int agg_err = 0;
  LD_FETCH(&agg_err, b.counters + 7, sizeof(int), cx.s);
  hipLaunchKernelGGL(ld_degstats_kernel, GRIDK(nn), 0, cx.s, cb.indptr, (int)nn, b.counters + 8);
  SCAMD_LAUNCH_CHECK();
  LD_FETCH(dstat, b.counters + 8, sizeof(int) * 4, cx.s);
  LD_FETCH(&nnz_new, cb.indptr + nn, sizeof(int64_t), cx.s);
  if (cx.cpm) {  // sizes add up over the members; strengths are the row sums of the coarse graph
    rc = run_fill(cx, Filler().add(cb.k, sizeof(long long) * nn));
    if (rc != SCAMD_OK) return rc;
    hipLaunchKernelGGL(ld_agg_nodeweight_kernel, GRID1(g.n), 0, cx.s, g.n, (const int*)b.cid, g.k, cb.k);
  } else {
    hipLaunchKernelGGL(ld_strength_kernel, GRIDW(nn), 0, cx.s, cb.indptr, cb.wq, (int)nn, cb.k);
  }
  SCAMD_LAUNCH_CHECK();
  std::swap(b.comm, b.comm_tmp);  // (the coarse phase-1 partition ld_coarse_comm_kernel wrote is b.comm from here on)
  LD_SYNC(cx.s);
  SCAMD_REQUIRE(agg_err == 0, SCAMD_EINTERNAL, "leiden: coarse-row table overflow");
  out->n = (int)nn;
  out->max_deg = dstat[0];
  out->n_gt96 = dstat[1];
  out->n_gt192 = dstat[2];
  out->n_gt384 = dstat[3];
  out->nnz = nnz_new;
  out->indptr = cb.indptr;
  out->indices = cb.indices;
  out->wq = cb.wq;
  out->k = cb.k;
  return SCAMD_OK;
}

// one Leiden iteration starting from the level-0 partition in b.memb; result back into b.memb
static double dbg_now(LeidenCtx& cx) {  // debug trace only: drains the stream, then host time in ms
  (void)hipStreamSynchronize(cx.s);
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

// all remaining levels in one workgroup (ld_small_levels_kernel); b.comm[i] = final community of entry node i
static int small_levels(LeidenCtx& cx, const LevelGraph& g, int level) {
  LeidenBuffers& b = cx.b;
  SmallArgs a;
  a.indptr = g.indptr;
  a.indices = g.indices;
  a.wq = g.wq;
  a.k = g.k;
  a.n = g.n;
  for (int i = 0; i < 2; ++i) {
    a.ix[i] = b.cb[i].indices;
    a.w[i] = b.cb[i].wq;
  }
  a.first_dst = level & 1;  // level L >= 1 lives in cb[(L - 1) & 1]
  a.comm = b.comm;
  a.gg = cx.gscale();
  a.inv_beta = cx.inv_beta;
  a.seed = cx.seed;
  a.iter = cx.iter;
  a.lm_stop_permille = cx.lm_stop_permille;
  a.seq_n = cx.small_seq_n;
  a.info = b.counters + 12;
  hipLaunchKernelGGL(ld_small_levels_kernel, dim3(1), dim3(SM_THREADS), sizeof(SmallLds), cx.s, a);
  SCAMD_LAUNCH_CHECK();
  if (leiden_debug()) {
    int h[4];
    LD_FETCH(h, b.counters + 12, sizeof(h), cx.s);
    LD_SYNC(cx.s);
    fprintf(stderr, "[leiden] small levels from level %d (n=%d nnz=%lld): %d levels, %d moves, %d merges -> n=%d\n", level, g.n,
            (long long)g.nnz, h[0], h[1], h[2], h[3]);
  }
  return SCAMD_OK;
}

// (from the partition in b.memb; the result goes to b.memb_work)
static int leiden_iteration(LeidenCtx& cx, const LevelGraph& g0) {
  LeidenBuffers& b = cx.b;
  hipLaunchKernelGGL(ld_iter_init_kernel, GRID1(g0.n), 0, cx.s, g0.n, (const int*)b.memb, b.comm, b.node_of);
  SCAMD_LAUNCH_CHECK();
  LevelGraph g = g0;
  cx.l0_moves = -1;  // (a graph small enough to start in the one-workgroup kernel reports no level-0 count)
  for (int level = 0; level < MAX_LEVELS; ++level) {
    int moves = 0;
    const bool dbg = leiden_debug();
    const double t0 = dbg ? dbg_now(cx) : 0.0;
    if (cx.small_levels && g.n <= SMALL_N && g.nnz <= SMALL_NNZ) {
      const int rcs = small_levels(cx, g, level);
      if (rcs != SCAMD_OK) return rcs;
      if (dbg) fprintf(stderr, "[leiden] small levels %.2f ms\n", dbg_now(cx) - t0);
      break;
    }
    int rc = local_moving(cx, g, &moves);
    if (rc != SCAMD_OK) return rc;
    if (level == 0) cx.l0_moves = moves;
    cx.n_levels = level + 1;
    const double t1 = dbg ? dbg_now(cx) : 0.0;
    int merged = 0;
    rc = refinement(cx, g, &merged);
    if (rc != SCAMD_OK) return rc;
    const double t2 = dbg ? dbg_now(cx) : 0.0;
    if (dbg)
      fprintf(stderr, "[leiden] level %d n=%d nnz=%lld maxdeg=%d: local moving %.2f ms (%d moves), refinement %.2f ms (%d merged)\n",
              level, g.n, (long long)g.nnz, g.max_deg, t1 - t0, moves, t2 - t1, merged);
    if (merged == 0) break;
    LevelGraph gn;
    int n_new = 0;
    rc = aggregate(cx, g, g0.n, level & 1, &gn, &n_new);
    if (rc != SCAMD_OK) return rc;
    if (dbg) fprintf(stderr, "[leiden] level %d aggregate %.2f ms -> n=%d\n", level, dbg_now(cx) - t2, n_new);
    if (n_new == g.n) break;
    g = gn;
  }
  hipLaunchKernelGGL(ld_gather_kernel, GRID1(g0.n), 0, cx.s, g0.n, b.comm, b.node_of, b.memb_work);
  SCAMD_LAUNCH_CHECK();
  return SCAMD_OK;
}

// b.memb (values < n) relabelled to consecutive ids ordered by (size desc, first member asc) -> out (the caller's buffer)
static int renumber(LeidenCtx& cx, int n, int* n_comm, int* out) {
  LeidenBuffers& b = cx.b;
  {
    const int rcf = run_fill(cx, Filler().add(b.minmember, sizeof(int) * n, 0x7f7f7f7fu).add(b.csize, sizeof(int) * n));
    if (rcf != SCAMD_OK) return rcf;
  }
  hipLaunchKernelGGL(ld_minmember_kernel, dim3((unsigned)std::min(REDUCE_GRID, ceil_div(n, 1024))), dim3(1024), 0, cx.s, n, b.memb, b.minmember, b.csize);
  SCAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(ld_flag_kernel, GRID1(n), 0, cx.s, n, b.csize, b.flag, (int*)nullptr);
  SCAMD_LAUNCH_CHECK();
  int rc = exclusive_scan_i32_i64(b.flag, n, b.newid, b.scan_tmp, cx.s);
  if (rc != SCAMD_OK) return rc;
  int64_t nc = 0;
  LD_FETCH(&nc, b.newid + n, sizeof(int64_t), cx.s);
  LD_SYNC(cx.s);
  hipLaunchKernelGGL(ld_commkeys_kernel, GRID1(n), 0, cx.s, n, b.csize, b.minmember, b.newid, b.ckeys, b.cids);
  SCAMD_LAUNCH_CHECK();
  if (nc <= 131072) {
    hipLaunchKernelGGL(ld_rank_kernel, GRID1(nc), 0, cx.s, (int)nc, b.ckeys, b.cids, b.newlabel);
  } else {
    // degenerate partitions (hundreds of thousands of communities): consecutive ids by first member only
    hipLaunchKernelGGL(ld_compact_label_kernel, GRID1(nc), 0, cx.s, (int)nc, b.cids, b.newlabel);
  }
  SCAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(ld_relabel_kernel, GRID1(n), 0, cx.s, n, b.newlabel, (const int*)b.memb, out);
  SCAMD_LAUNCH_CHECK();
  *n_comm = (int)nc;
  return SCAMD_OK;
}

static int setup_level0(LeidenCtx& cx, const int64_t* indptr, const int32_t* indices, const float* weights, int64_t n,
                        int64_t nnz, LevelGraph* g0) {
  LeidenBuffers& b = cx.b;
  if (nnz > 0) {
    hipLaunchKernelGGL(ld_quantize_kernel, dim3((unsigned)ceil_div(nnz, 256)), dim3(256), 0, cx.s, weights, nnz, b.wq0);
    SCAMD_LAUNCH_CHECK();
  }
  {
    const int rcf = run_fill(cx, Filler().add(b.total, sizeof(unsigned long long) * 4).add(b.counters, sizeof(int) * 16));
    if (rcf != SCAMD_OK) return rcf;
  }
  hipLaunchKernelGGL(ld_strength_kernel, GRIDW(n), 0, cx.s, indptr, b.wq0, (int)n, b.k0);
  SCAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(ld_sum_kernel, dim3(256), dim3(256), 0, cx.s, b.k0, (int)n, b.total);
  SCAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(ld_degstats_kernel, GRIDK(n), 0, cx.s, indptr, (int)n, b.counters + 8);
  SCAMD_LAUNCH_CHECK();
  unsigned long long tot = 0;
  int dstat[4] = {0, 0, 0, 0};
  LD_FETCH(&tot, b.total, sizeof(tot), cx.s);
  LD_FETCH(dstat, b.counters + 8, sizeof(int) * 4, cx.s);
  LD_SYNC(cx.s);
  cx.m2 = (double)tot;
  g0->max_deg = dstat[0];
  g0->n_gt96 = dstat[1];
  g0->n_gt192 = dstat[2];
  g0->n_gt384 = dstat[3];
  g0->n = (int)n;
  g0->nnz = nnz;
  g0->indptr = indptr;
  g0->indices = indices;
  g0->wq = b.wq0;
  g0->k = b.k0;
  if (cx.cpm && cx.node_weights) {
    // (b.counters[7] is zero: cleared above, nothing has raised it since)
    hipLaunchKernelGGL(ld_nodeweight_quantize_kernel, GRID1(n), 0, cx.s, cx.node_weights, (int)n, b.k0, b.counters + 7);
    SCAMD_LAUNCH_CHECK();
    int bad = 0;
    LD_FETCH(&bad, b.counters + 7, sizeof(int), cx.s);
    LD_SYNC(cx.s);
    SCAMD_REQUIRE(bad == 0, SCAMD_EINVAL, "leiden: node weights must lie in [0, 1e6]");
    cx.nw_scale = NODE_WEIGHT_SCALE;
  } else if (cx.cpm) {  // the vertex weights of CPM are counts; the strengths above were only needed for 2m
    hipLaunchKernelGGL(ld_fill_i64_kernel, GRID1(n), 0, cx.s, b.k0, (int)n, 1ll);
    SCAMD_LAUNCH_CHECK();
  }
  return SCAMD_OK;
}

}  // namespace scamd

using namespace scamd;

extern "C" size_t scamd_leiden_workspace_bytes(int64_t n, int64_t nnz) {
  if (n <= 0 || nnz < 0) return 0;
  Workspace ws(nullptr, 0);
  LeidenBuffers b;
  leiden_carve(ws, n, nnz, &b);
  return ws.used();
}

// memb[v] = init[v]; *err |= 1 when an id lies outside [0, n)
__global__ void ld_copy_membership_kernel(int n, const int* __restrict__ init, int* __restrict__ memb, int* __restrict__ err) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  const int c = init[v];
  if (c < 0 || c >= n) {
    atomicOr(err, 1);
    memb[v] = v;
  } else {
    memb[v] = c;
  }
}

static int leiden_run(const int64_t* indptr, const int32_t* indices, const float* weights, int64_t n, int64_t nnz,
                      double resolution, int n_iterations, double beta, uint64_t seed, const int32_t* initial_membership,
                      int32_t* membership, double* modularity_host, int32_t* n_communities_host, void* workspace,
                      size_t workspace_bytes, scamd_stream_t stream, int objective, const float* node_weights = nullptr);

extern "C" int scamd_leiden_csr_f32(const int64_t* indptr, const int32_t* indices, const float* weights, int64_t n,
                                    int64_t nnz, double resolution, int n_iterations, double beta, uint64_t seed,
                                    int32_t* membership, double* modularity_host, int32_t* n_communities_host,
                                    void* workspace, size_t workspace_bytes, scamd_stream_t stream) {
  return leiden_run(indptr, indices, weights, n, nnz, resolution, n_iterations, beta, seed, nullptr, membership,
                    modularity_host, n_communities_host, workspace, workspace_bytes, stream, 0);
}

// ... starting from a given partition instead of singletons (`initial_membership` of leidenalg.find_partition /
// igraph community_leiden, passed through `**clustering_args` at src/scanpy/tools/_leiden.py:66, 174-196): ids in [0, n),
// device pointer; n_iterations = 0 returns it renumbered with its modularity.
extern "C" int scamd_leiden_csr_init_f32(const int64_t* indptr, const int32_t* indices, const float* weights, int64_t n,
                                         int64_t nnz, double resolution, int n_iterations, double beta, uint64_t seed,
                                         const int32_t* initial_membership, int32_t* membership,
                                         double* modularity_host, int32_t* n_communities_host, void* workspace,
                                         size_t workspace_bytes, scamd_stream_t stream) {
  SCAMD_REQUIRE(initial_membership, SCAMD_EINVAL, "leiden: null initial membership");
  return leiden_run(indptr, indices, weights, n, nnz, resolution, n_iterations, beta, seed, initial_membership, membership,
                    modularity_host, n_communities_host, workspace, workspace_bytes, stream, 0);
}

// ... with the objective named: 0 = modularity (the two entry points above), 1 = CPM -- igraph's
// `community_leiden(objective_function='CPM')`, reachable through `sc.tl.leiden(flavor='igraph', objective_function='CPM')`
// (src/scanpy/tools/_leiden.py:188-196): every vertex weighs 1, the resolution is NOT divided by 2m, a community pays
// gamma n_C^2.  initial_membership may be NULL.  *modularity_host is the (resolution 1) modularity of the returned partition
// either way -- what the reference stores (`part.modularity`, :219) -- for objective 0 at the given resolution, as before.
extern "C" int scamd_leiden_csr_ex_f32(const int64_t* indptr, const int32_t* indices, const float* weights, int64_t n,
                                       int64_t nnz, double resolution, int n_iterations, double beta, uint64_t seed,
                                       int objective, const int32_t* initial_membership, int32_t* membership,
                                       double* modularity_host, int32_t* n_communities_host, void* workspace,
                                       size_t workspace_bytes, scamd_stream_t stream) {
  SCAMD_REQUIRE(objective == 0 || objective == 1, SCAMD_EINVAL, "leiden: objective %d (0 = modularity, 1 = CPM)", objective);
  return leiden_run(indptr, indices, weights, n, nnz, resolution, n_iterations, beta, seed, initial_membership, membership,
                    modularity_host, n_communities_host, workspace, workspace_bytes, stream, objective);
}

// ... CPM with the caller's vertex weights: igraph's `community_leiden(objective_function='CPM', node_weights=...)`, reachable
// through `sc.tl.leiden(flavor='igraph', objective_function='CPM', node_weights=...)` (`**clustering_args`,
// src/scanpy/tools/_leiden.py:66, 188-196): a community pays gamma (sum of its members' weights)^2.  node_weights: device
// pointer, n floats in [0, 1e6] (held in fixed point with 16 fractional bits); NULL = every vertex weighs 1.  With the
// modularity objective the vertex weights ARE the strengths: node_weights must be NULL there.
extern "C" int scamd_leiden_csr_nw_f32(const int64_t* indptr, const int32_t* indices, const float* weights, int64_t n,
                                       int64_t nnz, double resolution, int n_iterations, double beta, uint64_t seed,
                                       int objective, const float* node_weights, const int32_t* initial_membership,
                                       int32_t* membership, double* modularity_host, int32_t* n_communities_host,
                                       void* workspace, size_t workspace_bytes, scamd_stream_t stream) {
  SCAMD_REQUIRE(objective == 0 || objective == 1, SCAMD_EINVAL, "leiden: objective %d (0 = modularity, 1 = CPM)", objective);
  SCAMD_REQUIRE(objective == 1 || node_weights == nullptr, SCAMD_EUNSUPPORTED,
                "leiden: node weights with the modularity objective (its vertex weights are the strengths)");
  return leiden_run(indptr, indices, weights, n, nnz, resolution, n_iterations, beta, seed, initial_membership, membership,
                    modularity_host, n_communities_host, workspace, workspace_bytes, stream, objective, node_weights);
}

static int leiden_run(const int64_t* indptr, const int32_t* indices, const float* weights, int64_t n, int64_t nnz,
                      double resolution, int n_iterations, double beta, uint64_t seed, const int32_t* initial_membership,
                      int32_t* membership, double* modularity_host, int32_t* n_communities_host, void* workspace,
                      size_t workspace_bytes, scamd_stream_t stream, int objective, const float* node_weights) {
  SCAMD_REQUIRE(indptr && membership && (nnz == 0 || (indices && weights)), SCAMD_EINVAL, "leiden: null pointer");
  SCAMD_REQUIRE(n >= 1 && n < ((int64_t)1 << 31) && nnz >= 0, SCAMD_EINVAL, "leiden: bad shape n=%lld nnz=%lld",
                (long long)n, (long long)nnz);
  SCAMD_REQUIRE(resolution >= 0.0, SCAMD_EINVAL, "leiden: negative resolution");
  {
    const char* e = getenv("SCAMD_LEIDEN_DEBUG");
    g_leiden_debug = e && (e[0] == '1' || e[0] == '2');
    g_leiden_debug_sync = e && e[0] == '2';
  }
  LeidenCtx cx;
  cx.s = stream;
  cx.gamma = resolution;
  cx.inv_beta = beta > 0.0 ? 1.0 / (beta * WSCALE) : 0.0;  // beta <= 0: the greedy limit (largest gain, no "stay")
  cx.seed = (unsigned int)(seed ^ (seed >> 32)) * 0x9E3779B1u + 0x632BE5ABu;
  if (const char* e = getenv("SCAMD_LEIDEN_LM_STOP_PERMILLE")) cx.lm_stop_permille = cx.lm_stop_permille_big = atoi(e);
  cx.lm_classes = classes_env("SCAMD_LEIDEN_LM_CLASSES");
  cx.rf_classes = classes_env("SCAMD_LEIDEN_RF_CLASSES");
  if (const char* e = getenv("SCAMD_LEIDEN_SMALL")) cx.small_levels = e[0] != '0';
  if (const char* e = getenv("SCAMD_LEIDEN_FUSE")) cx.no_fuse = e[0] == '0';
  if (const char* e = getenv("SCAMD_LEIDEN_POLISH")) cx.polish = e[0] != '0';
  cx.cpm = objective == 1;
  cx.node_weights = node_weights;
  if (cx.cpm) cx.small_levels = false;  // (ld_small_levels_kernel derives its coarse vertex weights from row sums: strengths)
  for (int i = 0; i < LD_NSTATS; ++i) g_ld_stats[i] = 0;
  g_ld_sweep_bytes = 0.0;
  if (const char* e = getenv("SCAMD_LEIDEN_SMALL_SEQ")) cx.small_seq_n = atoi(e);
  if (const char* e = getenv("SCAMD_LEIDEN_AGG_WAVE_MAX")) cx.agg_wave_max = std::min(atoi(e), (int)WH_MAX_DEG);
  if (const char* e = getenv("SCAMD_LEIDEN_AGG_MID_MAX")) cx.agg_mid_max = std::min(atoi(e), (int)AGG_MID_MAX);
  if (const char* e = getenv("SCAMD_LEIDEN_HUB_TRY_PROBES")) cx.hub_try_probes = std::max(0, atoi(e));
  if (const char* e = getenv("SCAMD_LEIDEN_SMALL_SWEEP_ACT")) cx.small_sweep_act = std::max(0, atoi(e));
  if (const char* e = getenv("SCAMD_LEIDEN_SMALL_SWEEP_CLASSES")) cx.small_sweep_classes = std::max(1, std::min(atoi(e), (int)MAX_CLASSES));
  if (const char* e = getenv("SCAMD_LEIDEN_AGG_WAVE_WORK")) cx.agg_wave_work = std::max(1, atoi(e));
  if (const char* e = getenv("SCAMD_LEIDEN_AGG_MID_WORK")) cx.agg_mid_work = std::max(1, atoi(e));
  if (const char* e = getenv("SCAMD_LEIDEN_AGG_MID_GRID")) cx.agg_mid_grid = std::max(1, atoi(e));
  if (const char* e = getenv("SCAMD_LEIDEN_AGG_BIG_GRID")) cx.agg_big_grid = std::max(1, atoi(e));
  if (const char* e = getenv("SCAMD_LEIDEN_AGG_PASS_KEYS"))
    cx.agg_pass_keys = std::max(16, std::min(atoi(e), (int)AGG_BIG_PASS));
  Workspace ws(workspace, workspace_bytes);
  leiden_carve(ws, n, nnz, &cx.b);
  SCAMD_REQUIRE(workspace && ws.ok, SCAMD_EWORKSPACE, "leiden: workspace %zu < required %zu", workspace_bytes,
                ws.used());
  SCAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ld_move_hub_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)HUB_LDS));
  SCAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ld_refine_propose_hub_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)HUB_LDS));
  SCAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ld_agg_merge_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)HUB_LDS));
  SCAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ld_agg_block_kernel<BHUB_SLOTS, 1024>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)HUB_LDS));
  SCAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ld_small_levels_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SmallLds)));
  LevelGraph g0;
  int rc = setup_level0(cx, indptr, indices, weights, n, nnz, &g0);
  if (rc != SCAMD_OK) return rc;
  LeidenBuffers& b = cx.b;
  if (initial_membership) {
    // (b.counters[7] is zero: setup_level0 cleared the counters and checked whatever it raised)
    hipLaunchKernelGGL(ld_copy_membership_kernel, GRID1(n), 0, cx.s, (int)n, initial_membership, b.memb, b.counters + 7);
    SCAMD_LAUNCH_CHECK();
    int bad = 0;
    LD_FETCH(&bad, b.counters + 7, sizeof(int), cx.s);
    LD_SYNC(cx.s);
    SCAMD_REQUIRE(bad == 0, SCAMD_EINVAL, "leiden: initial membership ids must lie in [0, n)");
  } else {
    hipLaunchKernelGGL(ld_iota_kernel, GRID1(n), 0, cx.s, b.memb, (int)n);
    SCAMD_LAUNCH_CHECK();
  }
  double q_best = 0.0;
  if (cx.m2 > 0.0) {
    rc = compute_totals(cx, g0, b.memb);
    if (rc == SCAMD_OK) rc = quality(cx, g0, b.memb, &q_best);
    if (rc != SCAMD_OK) return rc;
    // b.memb is ALWAYS the best partition seen (the input of the next iteration); an iteration writes into b.memb_work and
    // the two change places when it improved -- no copies (round 5: two 4-byte-per-vertex blits per iteration)
    int iter_cap = MAX_OUTER_ITERS;
    if (const char* e = getenv("SCAMD_LEIDEN_ITER_CAP"))
      if (atoi(e) > 0) iter_cap = atoi(e);
    g_ld_stats[13] = iter_cap;
    int max_iter = n_iterations < 0 ? iter_cap : n_iterations;
    // (test knob: caps the outer loop of an n_iterations < 0 run, so that the polish meets an unfinished partition)
    if (const char* e = getenv("SCAMD_LEIDEN_MAX_ITERS"))
      if (n_iterations < 0 && atoi(e) > 0) max_iter = std::min(max_iter, atoi(e));
    int bad_iters = 0;
    // true once b.memb has been the INPUT of an iteration whose level-0 local moving found nothing to move: its
    // first sweep decides for every vertex on the final state, so that is a proof of node optimality
    bool best_is_clean = false;
    bool ended_by_cap = true;  // the loop below ran out of iterations (n_iterations < 0: MAX_OUTER_ITERS) instead of converging
    for (int it = 0; it < max_iter; ++it) {
      cx.iter = it;
      g_ld_stats[0] = it + 1;
      rc = leiden_iteration(cx, g0);
      if (rc != SCAMD_OK) return rc;
      if (it == 0) g_ld_stats[7] = cx.n_levels;
      double q = 0.0;
      rc = compute_totals(cx, g0, b.memb_work);
      if (rc == SCAMD_OK) rc = quality(cx, g0, b.memb_work, &q);
      if (rc != SCAMD_OK) return rc;
      const bool improved = q > q_best + 1e-12;
      const bool worse = q < q_best - 1e-12;
      if (leiden_debug()) fprintf(stderr, "[leiden] iteration %d: Q = %.10f (best before %.10f)\n", it, q, q_best);
      if (improved) {
        q_best = q;
        bad_iters = 0;
        best_is_clean = false;
        std::swap(b.memb, b.memb_work);
      } else {
        if (cx.l0_moves == 0) best_is_clean = true;
        // synchronous moves and the randomised refinement are not monotone: the best partition seen stays in b.memb
      }
      // n_iterations < 0: until an iteration changes nothing (leidenalg: `while diff_inc > 0`).  An iteration that
      // reproduces the best quality exactly found nothing to move -> done.  One that came out WORSE was unlucky in its
      // random merges (the next one draws different noise, cx.iter is part of the noise seed): two of those in a row end
      // the run as well -- without this patience a single unlucky iteration right after the first one froze the
      // result of ONE iteration (Q 0.8028 instead of 0.812 on the 700-cell fixture).
      if (n_iterations < 0 && !improved) {
        ended_by_cap = false;
        if (!worse || ++bad_iters >= 2) break;
        ended_by_cap = true;
      }
    }
    // n_iterations < 0 promises a STABLE partition (leidenalg iterates until an iteration changes nothing; such a
    // partition is node optimal and g-separated).  Ours is the best of a run of non-monotone iterations.  Unless the last
    // iteration proved it (best_is_clean): strictly monotone single-vertex moves until a sweep over all vertices finds
    // none (polish_level0), then -- the moved vertices may have made two communities mergeable, or cut one in two -- ONE
    // ordinary iteration from the polished partition: its level-0 moving finds nothing, its refinement and coarse levels
    // re-examine everything else.  No gain: stable, done.  A gain: accepted, and the polish runs again.
    int n_iter_total = g_ld_stats[0];
    ended_by_cap = ended_by_cap && max_iter == iter_cap;  // (not the test knob's cap: that one is followed up)
    g_ld_stats[11] = (n_iterations < 0 && ended_by_cap) ? 1 : 0;
    for (int pr = 0; n_iterations < 0 && cx.polish && pr <= MAX_POLISH_PASSES; ++pr) {
      if (best_is_clean) {
        if (pr == 0) g_ld_stats[6] = 1;
        break;
      }
      int ps[4] = {0, 0, 0, 0};
      rc = polish_level0(cx, g0, ps);
      if (rc != SCAMD_OK) return rc;
      g_ld_stats[3] += ps[0];
      g_ld_stats[4] += ps[1];
      g_ld_stats[5] += ps[2];
      g_ld_stats[10] += ps[3];
      if (ps[2] == 0) break;  // the full sweep found no improving move: node optimal as it stands
      double q = 0.0;
      rc = compute_totals(cx, g0, b.memb);
      if (rc == SCAMD_OK) rc = quality(cx, g0, b.memb, &q);
      if (rc != SCAMD_OK) return rc;
      if (leiden_debug())
        fprintf(stderr, "[leiden] polish %d: %d full sweeps, %d rounds, %d moves: Q %.10f -> %.10f\n", pr, ps[0], ps[1], ps[2], q_best, q);
      SCAMD_REQUIRE(q >= q_best - 1e-12, SCAMD_EINTERNAL, "leiden: the monotone polish lowered the quality (%.12f -> %.12f)", q_best, q);
      q_best = q;
      // (the last pass only polishes what the last accepted iteration left.  A run that the iteration cap ended was still
      // gaining a little with every iteration -- structure-less graphs do that for dozens of iterations: there is no stable
      // partition to verify, the polished one is node optimal and connected, and that is what is returned)
      if (pr == MAX_POLISH_PASSES || ended_by_cap) break;
      cx.iter = n_iter_total++;
      g_ld_stats[0] = n_iter_total;
      rc = leiden_iteration(cx, g0);
      if (rc != SCAMD_OK) return rc;
      rc = compute_totals(cx, g0, b.memb_work);
      if (rc == SCAMD_OK) rc = quality(cx, g0, b.memb_work, &q);
      if (rc != SCAMD_OK) return rc;
      if (leiden_debug()) fprintf(stderr, "[leiden] iteration after polish %d: Q = %.10f (polished %.10f), level-0 moves %d\n", pr, q, q_best, cx.l0_moves);
      if (q > q_best + 1e-12) {
        q_best = q;
        std::swap(b.memb, b.memb_work);
      } else {
        break;  // stable: the polished partition (b.memb) stands
      }
    }
  }
  if (cx.cpm && cx.m2 > 0.0) {
    // what is reported is the modularity of the partition (the reference stores `part.modularity`), not the CPM quality the
    // run maximised: strengths back into k0, totals, resolution 1
    hipLaunchKernelGGL(ld_strength_kernel, GRIDW(n), 0, cx.s, indptr, b.wq0, (int)n, b.k0);
    SCAMD_LAUNCH_CHECK();
    cx.cpm = false;
    cx.gamma = 1.0;
    rc = compute_totals(cx, g0, b.memb);
    if (rc == SCAMD_OK) rc = quality(cx, g0, b.memb, &q_best);
    if (rc != SCAMD_OK) return rc;
  }
  int nc = 0;
  rc = renumber(cx, (int)n, &nc, membership);  // (writes the caller's buffer directly)
  if (rc != SCAMD_OK) return rc;
  LD_SYNC(cx.s);
  if (modularity_host) *modularity_host = q_best;
  if (n_communities_host) *n_communities_host = nc;
  return SCAMD_OK;
}

// Test entry: the component split of the polish (split_disconnected) on a GIVEN membership (ids in [0, n)): every
// connected component of a community becomes a community of its own (id = its smallest vertex); membership is rewritten
// in place when anything was split, *n_split_host = components - communities.
extern "C" int scamd_leiden_debug_split_f32(const int64_t* indptr, const int32_t* indices, const float* weights, int64_t n,
                                            int64_t nnz, int32_t* membership, int32_t* n_split_host, void* workspace,
                                            size_t workspace_bytes, scamd_stream_t stream) {
  SCAMD_REQUIRE(indptr && membership && n_split_host && (nnz == 0 || (indices && weights)), SCAMD_EINVAL, "leiden split: null pointer");
  SCAMD_REQUIRE(n >= 1 && n < ((int64_t)1 << 31) && nnz >= 0, SCAMD_EINVAL, "leiden split: bad shape");
  LeidenCtx cx;
  cx.s = stream;
  cx.gamma = 1.0;
  cx.seed = 0;
  Workspace ws(workspace, workspace_bytes);
  leiden_carve(ws, n, nnz, &cx.b);
  SCAMD_REQUIRE(workspace && ws.ok, SCAMD_EWORKSPACE, "leiden split: workspace %zu < required %zu", workspace_bytes, ws.used());
  LevelGraph g0;
  int rc = setup_level0(cx, indptr, indices, weights, n, nnz, &g0);
  if (rc != SCAMD_OK) return rc;
  hipLaunchKernelGGL(ld_copy_i32_kernel, GRID1(n), 0, cx.s, (int)n, (const int*)membership, cx.b.comm);
  SCAMD_LAUNCH_CHECK();
  rc = compute_totals(cx, g0, cx.b.comm);
  if (rc != SCAMD_OK) return rc;
  int n_split = 0;
  rc = split_disconnected(cx, g0, &n_split);
  if (rc != SCAMD_OK) return rc;
  if (n_split > 0) {
    hipLaunchKernelGGL(ld_copy_i32_kernel, GRID1(n), 0, cx.s, (int)n, (const int*)cx.b.comm, membership);
    SCAMD_LAUNCH_CHECK();
  }
  SCAMD_HIP_CHECK(hipStreamSynchronize(cx.s));
  *n_split_host = n_split;
  return SCAMD_OK;
}

extern "C" void scamd_leiden_last_stats(int32_t* out, int n) {
  g_ld_stats[9] = (int)std::min(2.0e9, g_ld_sweep_bytes / 1.0e6);
  for (int i = 0; i < n && i < LD_NSTATS; ++i) out[i] = g_ld_stats[i];
}

extern "C" int scamd_modularity_csr_f32(const int64_t* indptr, const int32_t* indices, const float* weights, int64_t n,
                                        int64_t nnz, const int32_t* membership, double resolution,
                                        double* modularity_host, void* workspace, size_t workspace_bytes,
                                        scamd_stream_t stream) {
  SCAMD_REQUIRE(indptr && membership && modularity_host && (nnz == 0 || (indices && weights)), SCAMD_EINVAL,
                "modularity: null pointer");
  SCAMD_REQUIRE(n >= 1 && n < ((int64_t)1 << 31) && nnz >= 0, SCAMD_EINVAL, "modularity: bad shape");
  {
    const char* e = getenv("SCAMD_LEIDEN_DEBUG");
    g_leiden_debug = e && e[0] == '1';
  }
  LeidenCtx cx;
  cx.s = stream;
  cx.gamma = resolution;
  cx.seed = 0;
  Workspace ws(workspace, workspace_bytes);
  leiden_carve(ws, n, nnz, &cx.b);
  SCAMD_REQUIRE(workspace && ws.ok, SCAMD_EWORKSPACE, "modularity: workspace %zu < required %zu", workspace_bytes,
                ws.used());
  LevelGraph g0;
  int rc = setup_level0(cx, indptr, indices, weights, n, nnz, &g0);
  if (rc != SCAMD_OK) return rc;
  *modularity_host = 0.0;
  if (cx.m2 <= 0.0) return SCAMD_OK;
  // membership ids must lie in [0, n)
  rc = compute_totals(cx, g0, membership);
  if (rc == SCAMD_OK) rc = quality(cx, g0, membership, modularity_host);
  return rc;
}
