// Exact Gram matrix G = X^T X (+ column sums) of a CSR float32 matrix on gfx950, in 64-bit fixed point.
//
// Replaces the per-row outer-product Gram kernel of the reference's covariance route
// (src/scanpy/preprocessing/_pca/_kernels.py:14-58, `_csr_gram_upper_triangular` / `csr_gram_dense`, used by
// PCAEighDask.fit at _pca/_dask.py:28-132) and, through it, every CSR pass of the PCA solve: once G (g x g) and
// the column sums are on the device, the eigen-solve is dense GEMM work on a 32 MB matrix.
//
// Design (deterministic; what bounds it: see the notes in front of bcast8 below):
//   * genes are cut into tiles of T = 128; a work item = (tile pair a <= b, row chunk); its T x T block of G lives
//     in LDS as int64 (128 KB) and is accumulated with ds_add_u64 -- integer addition is associative, so the
//     result does not depend on the order in which waves/lanes/blocks (or ranks) add: bitwise reproducible;
//   * products are exact: (double)x_ia * (double)x_ib is exact for float32 inputs, scaled by 2^S and rounded
//     once to int64 (S chosen by the host from max|x| and n so that no sum can overflow);
//   * a per-row tile pointer table (uint16, built once) gives each item the entries of a row that fall into
//     gene tiles a and b without searching; 8 lanes work on one row (lane = entry of tile b, loop over the
//     entries of tile a), 8 rows per wave step;
//   * each item flushes its block to global memory with 64-bit integer atomics (distinct addresses except
//     between the row chunks of one tile pair).
// Work: sum_i r_i^2 products (r_i = stored entries of row i) -- 5e9 at 1M x 2k, 5 % dense.
#include "common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace scamd {

constexpr int GT = 128;             // gene tile
constexpr int GRAM_THREADS = 1024;  // 16 waves per block, one block per CU (LDS bound)

__global__ void gram_absmax_kernel(const float* __restrict__ data, int64_t nnz, unsigned int* __restrict__ out_bits) {
  float m = 0.f;
  // 16-byte loads over the aligned middle of the array, scalars at both ends
  const int64_t head = std::min<int64_t>(nnz, (int64_t)((16 - (reinterpret_cast<uintptr_t>(data) & 15)) & 15) / 4);
  const int64_t nvec = (nnz - head) / 4;
  const float4* dv = reinterpret_cast<const float4*>(data + head);
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = tid; i < nvec; i += nthr) {
    const float4 v = dv[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  if (tid < head) m = fmaxf(m, fabsf(data[tid]));
  for (int64_t i = head + nvec * 4 + tid; i < nnz; i += nthr) m = fmaxf(m, fabsf(data[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out_bits, __float_as_uint(m));
}

// ptr[row][t] = position (relative to the row start) of the first entry with column >= t * GT, t = 0..ntile
__global__ __launch_bounds__(256) void gram_tileptr_kernel(const int64_t* __restrict__ indptr,
                                                           const int32_t* __restrict__ indices, int64_t n, int ntile,
                                                           unsigned short* __restrict__ ptr) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const int64_t rb = indptr[row];
  const int len = (int)(indptr[row + 1] - rb);
  unsigned short* out = ptr + row * (ntile + 1);
  int prev_tile = -1;  // tile of the entry before this chunk
  for (int e0 = 0; e0 < len; e0 += 64) {
    const int e = e0 + lane;
    const int t = (e < len) ? indices[rb + e] / GT : ntile;
    int tp = __shfl_up(t, 1);
    if (lane == 0) tp = prev_tile;
    if (e < len)
      for (int u = tp + 1; u <= t; ++u) out[u] = (unsigned short)e;
    prev_tile = __shfl(t, 63);
  }
  // tiles after the last entry (prev_tile is ntile when the last chunk was not full: the loop below is empty then,
  // so recompute the last real tile)
  const int last_tile = (len > 0) ? indices[rb + len - 1] / GT : -1;
  for (int u = last_tile + 1 + lane; u <= ntile; u += 64) out[u] = (unsigned short)len;
}

// What binds this kernel (round-2 measurements, tools/probes/lds_atomic_probe.hip + profiles/r02p_pca_stage_pmc*.csv):
// NOT the 64-bit LDS atomic -- the probe retires 14 ds_add_u64 lanes per CU per ns (3.6e12 /s chip-wide, ds_add_u32 23,
// ds_add_f64 7.5), the 5e9 products of the 1M x 2k matrix would take 1.4 ms at that rate.  The waves were parked on
// s_waitcnt (69 % of the wave-cycles): every step's pointer and entry loads were waited for where they were issued.
// Three fixes, each measured: requesting the next step's loads one step ahead and keeping the RAW loaded values until
// their step (any arithmetic at the load site makes the compiler wait there: 12.4 -> 10.3 ms), prefetching the first
// SIXTEEN entries per row and tile instead of eight (10.3 -> 9.0 ms), column sums from the prefetched registers.  What
// is left is instruction issue: ~45 % of the lanes of a product instruction carry a product (8 lanes per row against
// 6.4 entries per tile, the a-loop runs to the longest of 8 rows).
// A_FROM_MEM: every lane reads the tile-a entry of its row group straight from memory (one address per 8 lanes, L1
// hits) instead of receiving it by a broadcast: slower (17.5 ms), kept behind SCAMD_GRAM_A_FROM_MEM for measurements.
// Broadcast lane PP of every aligned group of 8 lanes to the whole group.  Tried: ds_bpermute (two LDS instructions with
// a lane-address register), DPP (quad broadcast + bank-masked row shift: three VALU with the copy the tied operand
// needs) and ds_swizzle (one instruction): 7.0 ms with DPP, 6.9 with the swizzle once the kernel was issue-bound.
template <int PP>
__device__ __forceinline__ int bcast8(int x) {
  // ds_swizzle, bit-mask mode: lane' = (lane & 0b11000) | PP inside each group of 32 -- one LDS-crossbar instruction
  return __builtin_amdgcn_ds_swizzle(x, 0x18 | (PP << 5));
}

// round-to-nearest-even of a float64 to int64: the 1.5 * 2^52 trick where it is exact, llrint beyond
__device__ __forceinline__ long long fixed_round(double x) {
  if (fabs(x) < 2251799813685248.0) {  // 2^51
    const double t = x + 6755399441055744.0;  // 1.5 * 2^52: the sum's low mantissa bits are round(x) in two's complement
    return __double_as_longlong(t) - 0x4338000000000000ll;
  }
  return llrint(x);
}

// FAST: every product is rounded with the 2^52 trick alone (two instructions) and a product outside its range
// (|x| >= 2^51) only raises `*flag`; the host then runs the !FAST instantiation -- fixed_round with its llrint
// branch -- which returns at once unless the flag is up, after a kernel that clears the (then meaningless) sums.  The
// branchy form cost the common case dearly: the compiler does not jump over the llrint side, its six float64
// instructions issue with an empty exec mask behind EVERY product, plus four scalar instructions of mask bookkeeping.
template <bool A_FROM_MEM, bool FAST>
__global__ __launch_bounds__(GRAM_THREADS) void gram_tile_kernel(
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices, const float* __restrict__ data,
    int64_t n, int ntile, const unsigned short* __restrict__ ptr, int rows_per_chunk, double scale,
    unsigned long long* __restrict__ gram, int64_t ld, unsigned long long* __restrict__ colsum,
    unsigned int* __restrict__ flag) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long tile[];  // [GT][GT] + [GT] column sums
  unsigned long long* csum = tile + GT * GT;
  if constexpr (!FAST) {
    if (*flag == 0u) return;
  }
  // FAST: the range of the products is bounded once per wave from the largest |value| it met in either tile (two
  // v_max_f32 per eight products; tracking the products themselves was two to four instructions behind each of them)
  float amax = 0.f, bmax = 0.f;
  auto round_product = [&](double x) -> unsigned long long {
    if constexpr (FAST) {
      return (unsigned long long)(__double_as_longlong(x + 6755399441055744.0) - 0x4338000000000000ll);
    } else {
      return (unsigned long long)fixed_round(x);
    }
  };
  // blockIdx.x -> (pair index, chunk); pair index -> (a, b), a <= b
  const int chunk = blockIdx.y;
  int a = 0, rem = blockIdx.x;
  while (rem >= ntile - a) {
    rem -= ntile - a;
    ++a;
  }
  const int b = a + rem;
  const bool diag = (a == b);
  for (int i = threadIdx.x; i < GT * GT + GT; i += GRAM_THREADS) tile[i] = 0ull;
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rs = lane >> 3, q = lane & 7;
  const int64_t r0 = (int64_t)chunk * rows_per_chunk;
  const int64_t r1 = std::min<int64_t>(n, r0 + rows_per_chunk);
  const int a0 = a * GT, b0 = b * GT;
  // Two-stage software pipeline over the wave's steps (8 rows each).  PMC (profiles/r02p_pca_stage_pmc1.csv): 69 % of the
  // wave cycles of the straight loop were spent parked on s_waitcnt; a step is a chain of three dependent global loads
  // (tile pointers + indptr -> the row's entries of tile b -> of tile a) in front of ~40 products.  With the pointers of
  // step s + 2 and the first entry chunks of step s + 1 in flight while step s is computed: 13.2 -> 12.4 ms.  The rest of
  // the parking is lgkmcnt: 5.3e9 atomic lanes in 12.4 ms = 0.8 per CU cycle -- the 64-bit ds_add retires about one lane
  // per cycle on scattered addresses, and that, not its operands, is the wall.
  // (the structs hold RAW loaded values: any arithmetic on them at the point of the load makes the compiler wait for
  // the load right there -- `indices[p] - b0` in the first version of this pipeline put an s_waitcnt vmcnt behind every
  // prefetch and the pipeline did nothing)
  struct PtrsRaw {
    unsigned short a_lo, a_hi, b_lo, b_hi;
    int64_t rb;
    bool valid;
  };
  struct EntsRaw {  // lane q: entries q and q + 8 of either tile
    int ja, jb, ja2, jb2;
    float va, vb, va2, vb2;
  };
  const int64_t gstep = (GRAM_THREADS / 64) * 8;
  auto load_ptrs = [&](int64_t grp) -> PtrsRaw {
    PtrsRaw t{0, 0, 0, 0, 0, false};
    const int64_t row = grp + rs;
    if (grp < r1 && row < r1) {
      const unsigned short* pr = ptr + row * (ntile + 1);
      t.rb = indptr[row];
      t.a_lo = pr[a];
      t.a_hi = pr[a + 1];
      t.b_lo = pr[b];
      t.b_hi = pr[b + 1];
      t.valid = true;
    }
    return t;
  };
  // the first sixteen entries of either tile (columns still absolute).  Eight rows of a 5 %-dense matrix have a tile with
  // more than eight entries in four steps out of five, so the second chunk is prefetched with the first: fetched on
  // demand it put two to three serial memory latencies into most steps (2.4 us per step per wave, measured)
  auto load_ents = [&](int na_, int nb_, int64_t pa_, int64_t pb_) -> EntsRaw {
    EntsRaw e{0, 0, 0, 0, 0.f, 0.f, 0.f, 0.f};
    if (q < nb_) {
      e.jb = indices[pb_ + q];
      e.vb = data[pb_ + q];
    }
    if (q < na_) {
      e.ja = indices[pa_ + q];
      e.va = data[pa_ + q];
    }
    if (q + 8 < nb_) {
      e.jb2 = indices[pb_ + q + 8];
      e.vb2 = data[pb_ + q + 8];
    }
    if (q + 8 < na_) {
      e.ja2 = indices[pa_ + q + 8];
      e.va2 = data[pa_ + q + 8];
    }
    return e;
  };
  const int64_t g_first = r0 + (int64_t)wave * 8;
  int na, nb;
  int64_t pa, pb;
  {
    const PtrsRaw t = load_ptrs(g_first);
    na = t.valid ? t.a_hi - t.a_lo : 0;
    nb = t.valid ? t.b_hi - t.b_lo : 0;
    pa = t.rb + t.a_lo;
    pb = t.rb + t.b_lo;
  }
  EntsRaw en_cur = load_ents(na, nb, pa, pb);
  PtrsRaw pt_nxt = load_ptrs(g_first + gstep);
  for (int64_t grp = g_first; grp < r1; grp += gstep) {
    // the pointers of the next step were requested one step ago: turn them into ranges, request that step's entries and
    // the pointers of the step after it -- all before this step's products
    const int na_n = pt_nxt.valid ? pt_nxt.a_hi - pt_nxt.a_lo : 0;
    const int nb_n = pt_nxt.valid ? pt_nxt.b_hi - pt_nxt.b_lo : 0;
    const int64_t pa_n = pt_nxt.rb + pt_nxt.a_lo, pb_n = pt_nxt.rb + pt_nxt.b_lo;
    const EntsRaw en_nxt = load_ents(na_n, nb_n, pa_n, pb_n);
    const PtrsRaw pt_nn = load_ptrs(grp + 2 * gstep);
    int max_na = na, max_nb = nb;
#pragma unroll
    for (int o = 32; o >= 8; o >>= 1) {
      max_na = max(max_na, __shfl_xor(max_na, o));
      max_nb = max(max_nb, __shfl_xor(max_nb, o));
    }
    if (max_na != 0 && max_nb != 0) {
      for (int qc = 0; qc < max_nb; qc += 8) {
        const int myq = qc + q;
        const bool has_b = myq < nb;
        int jb = (qc == 0 ? en_cur.jb : en_cur.jb2) - b0;
        float vbf = qc == 0 ? en_cur.vb : en_cur.vb2;
        if (qc > 8) {  // rows with more than sixteen entries in tile b: later chunks straight from memory
          jb = 0;
          vbf = 0.f;
          if (has_b) {
            jb = indices[pb + myq] - b0;
            vbf = data[pb + myq];
          }
        }
        if constexpr (FAST) bmax = fmaxf(bmax, fabsf(vbf));
        const double vb = (double)vbf * scale;
        for (int pc = 0; pc < max_na; pc += 8) {
          // the group's next 8 entries of tile a: lane q holds entry pc + q
          int ja_l = (pc == 0 ? en_cur.ja : en_cur.ja2) - a0;
          float va_l = pc == 0 ? en_cur.va : en_cur.va2;
          if (pc > 8) {
            ja_l = 0;
            va_l = 0.f;
            if (pc + q < na) {
              ja_l = indices[pa + pc + q] - a0;
              va_l = data[pa + pc + q];
            }
          }
          if constexpr (FAST) amax = fmaxf(amax, fabsf(va_l));
          if (diag && qc == 0 && pc + q < na)  // column sums ride along on the diagonal items
            atomicAdd(&csum[ja_l], (unsigned long long)llrint((double)va_l * scale));
          if constexpr (!A_FROM_MEM) {
            // the 16 permutes of a group of eight tile-a entries are issued back to back and waited for once, the eight
            // (predicated) atomics follow without a wait in between; the float64 -> int64 rounding is the 2^52 trick
            // (exact for |x| < 2^51, llrint otherwise)
            int ja8[8];
            float va8[8];
            const int vai = __float_as_int(va_l);
            ja8[0] = bcast8<0>(ja_l); va8[0] = __int_as_float(bcast8<0>(vai));
            ja8[1] = bcast8<1>(ja_l); va8[1] = __int_as_float(bcast8<1>(vai));
            ja8[2] = bcast8<2>(ja_l); va8[2] = __int_as_float(bcast8<2>(vai));
            ja8[3] = bcast8<3>(ja_l); va8[3] = __int_as_float(bcast8<3>(vai));
            ja8[4] = bcast8<4>(ja_l); va8[4] = __int_as_float(bcast8<4>(vai));
            ja8[5] = bcast8<5>(ja_l); va8[5] = __int_as_float(bcast8<5>(vai));
            ja8[6] = bcast8<6>(ja_l); va8[6] = __int_as_float(bcast8<6>(vai));
            ja8[7] = bcast8<7>(ja_l); va8[7] = __int_as_float(bcast8<7>(vai));
#pragma unroll
            for (int pp = 0; pp < 8; ++pp) {
              if (has_b && pc + pp < na)
                atomicAdd(&tile[ja8[pp] * GT + jb], round_product((double)va8[pp] * vb));
            }
          } else {
            const int cnt = min(8, max_na - pc);
            for (int pp = 0; pp < cnt; ++pp) {
              if (has_b && pc + pp < na) {
                const int ja = indices[pa + pc + pp] - a0;
                const float va = data[pa + pc + pp];
                atomicAdd(&tile[ja * GT + jb], round_product((double)va * vb));
              }
            }
          }
        }
      }
    }
    na = na_n;
    nb = nb_n;
    pa = pa_n;
    pb = pb_n;
    en_cur = en_nxt;
    pt_nxt = pt_nn;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < GT * GT; i += GRAM_THREADS) {
    const unsigned long long v = tile[i];
    if (v) atomicAdd(&gram[(int64_t)(a0 + i / GT) * ld + b0 + (i % GT)], v);
  }
  if (diag)
    for (int i = threadIdx.x; i < GT; i += GRAM_THREADS)
      if (csum[i]) atomicAdd(&colsum[a0 + i], csum[i]);
  if constexpr (FAST) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      amax = fmaxf(amax, __shfl_xor(amax, o));
      bmax = fmaxf(bmax, __shfl_xor(bmax, o));
    }
    if (!((double)amax * ((double)bmax * scale) < 2251799813685248.0) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
  }
}

// the fast pass met a product outside the range of its rounding: clear what it accumulated
__global__ void gram_clear_if_flagged_kernel(unsigned long long* __restrict__ gram, int64_t n_gram,
                                             unsigned long long* __restrict__ colsum, int64_t n_col,
                                             const unsigned int* __restrict__ flag) {
  if (*flag == 0u) return;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_gram; i += (int64_t)gridDim.x * blockDim.x) gram[i] = 0ull;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_col; i += (int64_t)gridDim.x * blockDim.x) colsum[i] = 0ull;
}

// lower triangle <- upper triangle (tile pairs a < b were accumulated into the upper block only)
__global__ void gram_mirror_kernel(long long* __restrict__ gram, int64_t gp, int64_t ld) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i = blockIdx.y;
  if (j < gp && i < gp && (i / GT) > (j / GT)) gram[i * ld + j] = gram[j * ld + i];
}

}  // namespace scamd

using namespace scamd;

extern "C" size_t scamd_csr_gram_workspace_bytes(int64_t n, int64_t g) {
  if (n <= 0 || g <= 0) return 0;
  const int64_t ntile = (g + GT - 1) / GT;
  Workspace ws(nullptr, 0);
  (void)ws.take<unsigned short>((size_t)n * (ntile + 1));
  (void)ws.take<unsigned int>(4);
  return ws.used();
}

extern "C" int scamd_csr_gram_f32(const int64_t* indptr, const int32_t* indices, const float* data, int64_t n,
                                  int64_t g, int64_t nnz, int scale_bits, int64_t* gram, int64_t ld_gram,
                                  int64_t* colsum, float* absmax_host, void* workspace, size_t workspace_bytes,
                                  scamd_stream_t stream) {
  SCAMD_REQUIRE(indptr && (nnz == 0 || (indices && data)), SCAMD_EINVAL, "gram: null pointer");
  SCAMD_REQUIRE(n >= 1 && g >= 1 && nnz >= 0, SCAMD_EINVAL, "gram: bad shape");
  SCAMD_REQUIRE(g <= 65535 && n < ((int64_t)1 << 40), SCAMD_EUNSUPPORTED, "gram: g=%lld exceeds the uint16 tile pointers",
                (long long)g);
  const int ntile = (int)((g + GT - 1) / GT);
  const int64_t gp = (int64_t)ntile * GT;
  Workspace ws(workspace, workspace_bytes);
  unsigned short* ptr = ws.take<unsigned short>((size_t)n * (ntile + 1));
  unsigned int* mx = ws.take<unsigned int>(4);
  SCAMD_REQUIRE(workspace && ws.ok, SCAMD_EWORKSPACE, "gram: workspace %zu < required %zu", workspace_bytes, ws.used());
  hipStream_t s = stream;
  if (absmax_host) {  // phase 1: max |x| (the caller derives scale_bits from it, possibly after a max all-reduce)
    SCAMD_HIP_CHECK(hipMemsetAsync(mx, 0, 16, s));
    if (nnz > 0) {
      hipLaunchKernelGGL(gram_absmax_kernel, dim3(1024), dim3(256), 0, s, data, nnz, mx);
      SCAMD_LAUNCH_CHECK();
    }
    unsigned int bits = 0;
    SCAMD_READBACK_NOW(&bits, mx, 4, s);
    memcpy(absmax_host, &bits, 4);
    if (!gram) return SCAMD_OK;
  }
  SCAMD_REQUIRE(gram && colsum && ld_gram >= gp, SCAMD_EINVAL, "gram: output must be [%lld x ld >= %lld] (g padded to %d)",
                (long long)gp, (long long)gp, GT);
  SCAMD_REQUIRE(scale_bits >= 0 && scale_bits <= 60, SCAMD_EINVAL, "gram: scale_bits=%d", scale_bits);
  SCAMD_HIP_CHECK(hipMemsetAsync(gram, 0, sizeof(int64_t) * gp * ld_gram, s));
  SCAMD_HIP_CHECK(hipMemsetAsync(colsum, 0, sizeof(int64_t) * gp, s));
  hipLaunchKernelGGL(gram_tileptr_kernel, dim3((unsigned)ceil_div(n, 4)), dim3(256), 0, s, indptr, indices, n, ntile, ptr);
  SCAMD_LAUNCH_CHECK();
  const int npair = ntile * (ntile + 1) / 2;
  // ~8 items per CU: enough to balance, few enough that the 128 KB flushes stay negligible
  int n_chunks = std::max(1, std::min<int>((int)((n + 4095) / 4096), (256 * 8 + npair - 1) / npair));
  {
    // one workgroup per CU (the 128 KB tile), equal items: the launch runs in rounds of 256 workgroups, and a last round that
    // is half empty costs half a round -- 136 pairs x 16 chunks = 8.5 rounds: 6.90 ms, x 15 = 7.97 rounds: 6.57 ms
    // (profiles/r05v_gram_chunks.log; 17 chunks = 9.03 rounds: 7.15 ms).  The count near the default that wastes least:
    const int hi = (int)std::min<int64_t>((n + 4095) / 4096, n_chunks + 4);
    double best = 1e9;
    for (int c = std::max(1, n_chunks - 4); c <= hi; ++c) {
      const int64_t items = (int64_t)npair * c, rounds = (items + 255) / 256;
      const double waste = (double)(rounds * 256) / (double)items;
      if (waste < best - 1e-9) best = waste, n_chunks = c;
    }
  }
  if (const char* e = getenv("SCAMD_GRAM_CHUNKS")) n_chunks = std::max(1, std::min<int>((int)((n + 255) / 256), atoi(e)));  // (A/B knob)
  const int rows_per_chunk = (int)((n + n_chunks - 1) / n_chunks);
  n_chunks = (int)((n + rows_per_chunk - 1) / rows_per_chunk);
  const size_t lds = (size_t)(GT * GT + GT) * sizeof(unsigned long long);
  static const bool a_from_mem = [] {
    const char* e = getenv("SCAMD_GRAM_A_FROM_MEM");
    return e && e[0] == '1';
  }();
  auto gram_kernel = a_from_mem ? gram_tile_kernel<true, true> : gram_tile_kernel<false, true>;
  auto gram_exact = gram_tile_kernel<false, false>;
  SCAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gram_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  SCAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gram_exact),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  // column sums are scaled by 2^scale_bits as well (x * 2^S), products by 2^S: x_a * (x_b * 2^S)
  const double scale = std::ldexp(1.0, scale_bits);
  unsigned int* flag = mx + 1;  // products outside the fast rounding's range (see gram_tile_kernel)
  SCAMD_HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(unsigned int), s));
  unsigned long long* gram_u = reinterpret_cast<unsigned long long*>(gram);
  unsigned long long* colsum_u = reinterpret_cast<unsigned long long*>(colsum);
  hipLaunchKernelGGL(gram_kernel, dim3((unsigned)npair, (unsigned)n_chunks), dim3(GRAM_THREADS), lds, s, indptr,
                     indices, data, n, ntile, ptr, rows_per_chunk, scale, gram_u, ld_gram, colsum_u, flag);
  SCAMD_LAUNCH_CHECK();
  // no-ops unless the flag went up (no host round trip: the entry stays stream-ordered)
  hipLaunchKernelGGL(gram_clear_if_flagged_kernel, dim3(1024), dim3(256), 0, s, gram_u, gp * ld_gram, colsum_u, gp, flag);
  SCAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(gram_exact, dim3((unsigned)npair, (unsigned)n_chunks), dim3(GRAM_THREADS), lds, s, indptr,
                     indices, data, n, ntile, ptr, rows_per_chunk, scale, gram_u, ld_gram, colsum_u, flag);
  SCAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(gram_mirror_kernel, dim3((unsigned)ceil_div(gp, 256), (unsigned)gp), dim3(256), 0, s,
                     reinterpret_cast<long long*>(gram), gp, ld_gram);
  SCAMD_LAUNCH_CHECK();
  return SCAMD_OK;
}
