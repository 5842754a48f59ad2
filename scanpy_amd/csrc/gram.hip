// Exact Gram matrix G = X^T X (+ column sums) of a CSR float32 matrix on gfx950, in 64-bit fixed point.
//
// Replaces the per-row outer-product Gram kernel of the reference's covariance route
// (src/scanpy/preprocessing/_pca/_kernels.py:14-58, `_csr_gram_upper_triangular` / `csr_gram_dense`, used by
// PCAEighDask.fit at _pca/_dask.py:28-132) and, through it, every CSR pass of the PCA solve: once G (g x g) and
// the column sums are on the device, the eigen-solve is dense GEMM work on a 32 MB matrix.
//
// Design (deterministic):
//   * genes are cut into tiles of T = 128; a work item = (tile pair a <= b, row chunk); its T x T block of G lives
//     in LDS as int64 (128 KB) and is accumulated with ds_add_u64 -- integer addition is associative, so the
//     result does not depend on the order in which waves/lanes/blocks (or ranks) add: bitwise reproducible;
//   * products are exact: (double)x_ia * (double)x_ib is exact for float32 inputs, scaled by 2^S and rounded
//     once to int64 (S chosen by the host from max|x| and n so that no sum can overflow);
//   * each item flushes its block to global memory with 64-bit integer atomics (distinct addresses except
//     between the row chunks of one tile pair);
//   * round 6 (g <= 8192): the matrix is first rewritten as per-tile record slabs, one 128-byte line per (row, tile),
//     the rows of every block of 1024 ranked by their entry count per tile; four lanes work on one row, sixteen rows
//     per wave step (gram_pack_kernel, gram_rank_kernel, gram_quad_kernel below: 3.7 + 0.8 ms at 1M x 2k);
//   * wider matrices (and SCAMD_GRAM_LEGACY=1) take the round-2 kernel: a per-row tile pointer table (uint16) gives
//     each item the entries of a row that fall into gene tiles a and b without searching; 8 lanes work on one row
//     (lane = entry of tile b, loop over the entries of tile a), 8 rows per wave step (gram_tile_kernel: 6.7 ms).
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


// ---------------------------------------------------------------------------------------------------------------------
// Round 6: the PACKED kernels.  The disassembly of gram_tile_kernel above has ~150 instructions of per-step overhead in
// front of ~120 that make products, its a-loop runs to the longest of the step's eight rows (11 against a mean of 6.4
// entries) and once more for the rows with more than eight entries in tile b (83 % of the steps have one): 20 broadcast
// rounds per step for 328 products; and every visit of a (row, tile) gathers its ~26 bytes of indices and of values out of
// two 128-byte lines.  What was measured on the way (profiles/r06o .. r06w):
//   - rows sorted by entry count, same CSR gathers through a position table (three dependent load levels): a quarter fewer
//     instructions, 9.5 ms instead of 6.7 -- it waited for memory 2.8 x as long;
//   - record slabs + eight lanes per row, broadcasts through ds_swizzle: 4.0 ms, bound by the LDS pipe (2.8 ms busy: the
//     crossbar shares it with the atomics);
//   - record slabs + FOUR lanes per row, broadcasts by DPP quad_perm (VALU): 3.66 ms, LDS 2.0 ms busy (64 % of it bank
//     conflicts of the 64-bit atomics: 50 random addresses over 32 bank pairs), VALU 1.9 ms;
//   - entries 0 .. 7 and 8 .. 15 in two separate slabs of 64-byte half lines: 5.1 ms -- two 128-byte lines per visit.
// As built:
//   * gram_pack_kernel rewrites the matrix once per call as records: per gene tile t a slab [n_pad][16] of 8-byte records
//     {(column - t * GT) << 3 | count << 16, value}, absent entries zero: ONE aligned 128-byte line per (row, tile), read
//     with a base in scalar registers and a 32-bit offset, no position table, no masks -- a zero value adds nothing; the
//     upper half line is only written where a tile has more than eight entries (1.2 GB of stores instead of 2);
//   * gram_rank_kernel ranks the rows of every block of RB = 1024 by their number of entries per tile (counting sort in
//     LDS): the sixteen rows of a step have the SAME number of entries in tile a, the broadcast loop wastes nothing;
//   * the entries 8 .. 15 of tile b are not a second pass of that loop: the roles are exchanged for them (the lane keeps its
//     tile-a entries, the overflow entries of tile b are broadcast: 2-3 rounds instead of another 6-11);
//   * workgroups are dealt to the XCDs by tile neighbourhood (the pairs of a 4 x 4 block of the pair triangle share eight
//     slabs), chunk by chunk: what one workgroup pulled into the XCD's L2 the others find there.
// Rows with more than 16 entries in a tile finish from the CSR arrays.  The products and their rounding are those of
// gram_tile_kernel: the sums are the same integers.
constexpr int RB = 1024;                 // rows per sorting block
constexpr int SORT_BINS = 64;            // entry counts 0 .. 62, 63+ share the last bin
constexpr int SORT_TILES = 16;           // tiles ranked per pass (one wave scans one tile's bins)
constexpr int REC = 16;                  // records per (row, tile): 128 bytes
constexpr int PACK_MAX_TILES = 64;       // (the per-block count table: 64 KB of LDS; wider matrices take the round-2 kernel)

// ent[t][row][k] (row < n_pad = blocks * RB; k < REC): the row's k-th entry in tile t, see above (rows >= n: all zero);
// cnt8[row][t] = min(entries of the row in tile t, 255).  One row per wave, every record slot written (whole 128-byte lines).
constexpr int PACK_WAVES = 4;
__global__ __launch_bounds__(PACK_WAVES * 64) void gram_pack_kernel(const int64_t* __restrict__ indptr,
                                                                   const int32_t* __restrict__ indices,
                                                                   const float* __restrict__ data, int64_t n, int ntile,
                                                                   uint2* __restrict__ ent, unsigned char* __restrict__ cnt8,
                                                                   int64_t n_pad) {
  __shared__ unsigned short pos[PACK_WAVES][PACK_MAX_TILES + 2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned short* wpos = pos[wave];
  const int64_t row = (int64_t)blockIdx.x * PACK_WAVES + wave;
  if (row >= n_pad) return;
  int64_t rb = 0;
  int len = 0;
  if (row < n) {
    rb = indptr[row];
    len = (int)(indptr[row + 1] - rb);
  }
  // wpos[t] = position in the row of its first entry with column >= t * GT (t = 0 .. ntile)
  int prev_tile = -1;  // tile of the entry before this chunk
  for (int e0 = 0; e0 < len; e0 += 64) {
    const int e = e0 + lane;
    const int t = (e < len) ? indices[rb + e] / GT : ntile;
    int tp = __shfl_up(t, 1);
    if (lane == 0) tp = prev_tile;
    if (e < len)
      for (int u = tp + 1; u <= t; ++u) wpos[u] = (unsigned short)e;
    prev_tile = __shfl(t, 63);
  }
  const int last_tile = (len > 0) ? indices[rb + len - 1] / GT : -1;
  for (int u = last_tile + 1 + lane; u <= ntile; u += 64) wpos[u] = (unsigned short)len;
  __builtin_amdgcn_wave_barrier();  // (the wave's LDS writes above are read by other lanes below)
  // the row's record slots, 64 at a time: lane -> (tile, k); first the entries 0 .. 7 of every tile (every slot written), then
  // the entries 8 .. 15, where only the half lines of tiles with more than eight entries are written -- nobody uses the
  // others' contents (gram_quad_kernel masks them by the count): 1.2 GB of stores instead of 2 at 1M x 2k.  (Two separate
  // slabs of 64-byte half lines made the sweep 1.4 x slower: two 128-byte lines per visit instead of one.)
  for (int sl = lane; sl < ntile * (REC / 2); sl += 64) {
    const int t = sl / (REC / 2), k = sl % (REC / 2);
    const int p0 = wpos[t], c = wpos[t + 1] - p0;
    unsigned int col = 0u, vbits = 0u;
    if (k < c) {
      col = (unsigned int)(indices[rb + p0 + k] - t * GT);
      vbits = __float_as_uint(data[rb + p0 + k]);
    }
    ent[((int64_t)t * n_pad + row) * REC + k] = make_uint2((col << 3) | ((unsigned int)min(c, 255) << 16), vbits);
    if (k == 0) cnt8[row * ntile + t] = (unsigned char)min(c, 255);
  }
  for (int sl = lane; sl < ntile * (REC / 2); sl += 64) {
    const int t = sl / (REC / 2), k = sl % (REC / 2) + REC / 2;
    const int p0 = wpos[t], c = wpos[t + 1] - p0;
    if (c > REC / 2) {
      unsigned int col = 0u, vbits = 0u;
      if (k < c) {
        col = (unsigned int)(indices[rb + p0 + k] - t * GT);
        vbits = __float_as_uint(data[rb + p0 + k]);
      }
      ent[((int64_t)t * n_pad + row) * REC + k] = make_uint2((col << 3) | ((unsigned int)min(c, 255) << 16), vbits);
    }
  }
}

// perm[t][blk * RB + i]: block-local number of the row with the i-th fewest entries in tile t (counting sort of a block's
// rows, sixteen tiles per pass)
__global__ __launch_bounds__(RB) void gram_rank_kernel(const unsigned char* __restrict__ cnt8, int ntile,
                                                      unsigned short* __restrict__ perm, int64_t n_pad) {
  __shared__ int hist[SORT_TILES * SORT_BINS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row0 = (int64_t)blockIdx.x * RB;
  const int r = threadIdx.x;
  const unsigned char* cr = cnt8 + (row0 + r) * ntile;
  for (int t0 = 0; t0 < ntile; t0 += SORT_TILES) {
    const int nt = min(SORT_TILES, ntile - t0);
    for (int i = threadIdx.x; i < SORT_TILES * SORT_BINS; i += RB) hist[i] = 0;
    __syncthreads();
    int bin[SORT_TILES], rank[SORT_TILES];
#pragma unroll
    for (int j = 0; j < SORT_TILES; ++j) {
      bin[j] = 0, rank[j] = 0;
      if (j < nt) {
        bin[j] = min((int)cr[t0 + j], SORT_BINS - 1);
        rank[j] = atomicAdd(&hist[j * SORT_BINS + bin[j]], 1);
      }
    }
    __syncthreads();
    {  // exclusive scan of every tile's bins: wave j owns tile j
      const int v = hist[wave * SORT_BINS + lane];
      int incl = v;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
      }
      hist[wave * SORT_BINS + lane] = incl - v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SORT_TILES; ++j)
      if (j < nt) perm[(int64_t)(t0 + j) * n_pad + row0 + hist[j * SORT_BINS + bin[j]] + rank[j]] = (unsigned short)r;
    __syncthreads();
  }
}

// the k-th tile pair (a <= b) in blocked order: the triangle cut into 4 x 4 blocks, blocks row by row, pairs row by row inside
__device__ inline void gram_blocked_pair(int k, int ntile, int* a_out, int* b_out) {
  const int nb4 = (ntile + 3) / 4;
  for (int ba = 0; ba < nb4; ++ba)
    for (int bb = ba; bb < nb4; ++bb)
      for (int a = 4 * ba; a < min(4 * ba + 4, ntile); ++a)
        for (int b = max(a, 4 * bb); b < min(4 * bb + 4, ntile); ++b)
          if (k-- == 0) {
            *a_out = a, *b_out = b;
            return;
          }
  *a_out = -1, *b_out = -1;
}

template <bool FAST>
__global__ __launch_bounds__(GRAM_THREADS) void gram_quad_kernel(
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices, const float* __restrict__ data, int ntile,
    const uint2* __restrict__ ent, const unsigned short* __restrict__ perm, int64_t n_pad, int blocks_per_chunk, int nblk,
    int n_chunks, double scale, unsigned long long* __restrict__ gram, int64_t ld, unsigned long long* __restrict__ colsum,
    unsigned int* __restrict__ flag) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long tile[];  // [GT][GT] + [GT] column sums + the item
  unsigned long long* csum = tile + GT * GT;
  int* s_item = reinterpret_cast<int*>(csum + GT);
  if constexpr (!FAST) {
    if (*flag == 0u) return;
  }
  // launch slot -> (XCD, slot of the XCD) -> (chunk, pair): XCD x owns the pairs [x * npair / 8, (x + 1) * npair / 8) of the
  // blocked order and runs them chunk after chunk (workgroup i is dispatched to XCD i mod 8)
  if (threadIdx.x == 0) {
    const int npair = ntile * (ntile + 1) / 2;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int p_lo = (int)((int64_t)xcd * npair / 8), p_hi = (int)((int64_t)(xcd + 1) * npair / 8);
    const int px = p_hi - p_lo;
    int a = -1, b = -1, chunk = -1;
    if (px > 0 && slot < px * n_chunks) {
      chunk = slot / px;
      gram_blocked_pair(p_lo + slot % px, ntile, &a, &b);
    }
    s_item[0] = a, s_item[1] = b, s_item[2] = chunk;
  }
  __syncthreads();
  const int a = s_item[0], b = s_item[1], chunk = s_item[2];
  if (a < 0) return;
  const bool diag = (a == b);
  for (int i = threadIdx.x; i < GT * GT + GT; i += GRAM_THREADS) tile[i] = 0ull;
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int rs = lane >> 2, q = lane & 3;  // sixteen rows per step, four lanes each
  const int blk0 = chunk * blocks_per_chunk;
  const int blk1 = min(nblk, blk0 + blocks_per_chunk);
  const int steps = (blk1 - blk0) * (RB / 16);  // sixteen rows per wave step
  const int a0 = a * GT, b0 = b * GT;
  const unsigned short* perm_a = perm + (int64_t)a * n_pad + (int64_t)blk0 * RB;
  // slabs of the two tiles from the chunk's first row on, addressed in bytes by a 32-bit offset
  const char* ent_a = reinterpret_cast<const char*>(ent + ((int64_t)a * n_pad + (int64_t)blk0 * RB) * REC);
  const char* ent_b = reinterpret_cast<const char*>(ent + ((int64_t)b * n_pad + (int64_t)blk0 * RB) * REC);
  char* tile_b = reinterpret_cast<char*>(tile);
  float amax = 0.f, bmax = 0.f;
  auto round_product = [&](double x) -> unsigned long long {
    if constexpr (FAST) {
      return (unsigned long long)(__double_as_longlong(x + 6755399441055744.0) - 0x4338000000000000ll);
    } else {
      return (unsigned long long)fixed_round(x);
    }
  };
  auto add_at = [&](int byte_off, unsigned long long v) {
    atomicAdd(reinterpret_cast<unsigned long long*>(tile_b + byte_off), v);
  };

  // two levels of loads, each requested ahead of the level that needs its result (RAW values are kept until then: section
  // 3.0 of DESIGN.md): the sorted rows of the steps s + 2 WS .. and the records of step s + WS are in flight while step s
  // is computed
  struct Ent {  // lane q: records q, q + 4, q + 8, q + 12 of either tile
    uint2 a[4], b[4];
  };
  auto clamp_step = [&](int s) -> int { return min(s, steps - 1); };
  auto load_perm = [&](int s) -> unsigned short { return perm_a[(unsigned int)(clamp_step(s) * 16 + rs)]; };
  auto row_offset = [&](int s, unsigned short prow) -> unsigned int {  // byte offset of the row's records within a slab
    return ((unsigned int)(clamp_step(s) >> 6) * RB + prow) * (REC * 8u) + (unsigned int)q * 8u;
  };
  auto load_ent = [&](unsigned int off) -> Ent {
    Ent e;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      e.a[m] = *reinterpret_cast<const uint2*>(ent_a + off + 32 * m);
      e.b[m] = *reinterpret_cast<const uint2*>(ent_b + off + 32 * m);
    }
    return e;
  };

  // broadcast of lane P of every quad: one VALU instruction (DPP quad_perm), nothing on the LDS pipe the atomics need
#define SCAMD_QB(P, X) __builtin_amdgcn_update_dpp(0, (X), (P) * 0x55, 0xf, 0xf, true)
  auto wave_max_quads = [&](int v) -> int {  // maximum of a non-negative per-quad value over the wave
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true));  // row_shr:4
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true));  // row_shr:8: lanes 12 .. 15 hold the row's maximum
    return max(max(__builtin_amdgcn_readlane(v, 15), __builtin_amdgcn_readlane(v, 31)),
               max(__builtin_amdgcn_readlane(v, 47), __builtin_amdgcn_readlane(v, 63)));
  };
  auto compute = [&](const Ent& e, int s, unsigned short prow) __attribute__((always_inline)) {
    const int na = (int)(e.a[0].x >> 16), nb = (int)(e.b[0].x >> 16);
    // sorted ascending: the last quad of the step has the most entries in tile a (bin 63 holds every count >= 63)
    int namax = __builtin_amdgcn_readlane(na, 63);
    if (namax >= SORT_BINS - 1) namax = wave_max_quads(na);
    const int nbmax = wave_max_quads(nb);
    if (namax == 0 || nbmax == 0) return;
    int ja_o[4], jb_o[4];
    float va[4], vb[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      ja_o[m] = (int)(e.a[m].x & 0x3f8u) << 7;
      jb_o[m] = (int)(e.b[m].x & 0x3f8u);
      va[m] = __uint_as_float(e.a[m].y);
      vb[m] = __uint_as_float(e.b[m].y);
    }
    // the upper half line of a (row, tile) with at most eight entries was never written.  Tile b: its contents only reach
    // the lanes of their own quad, which are switched off (nb <= 8); tile a: masked where it is used (namax > 8, below)
    if (na <= 8) va[2] = 0.f, va[3] = 0.f;
    if (nb <= 8) vb[2] = 0.f, vb[3] = 0.f;
    if constexpr (FAST) {
      amax = fmaxf(amax, fmaxf(fmaxf(fabsf(va[0]), fabsf(va[1])), fmaxf(fabsf(va[2]), fabsf(va[3]))));
      bmax = fmaxf(bmax, fmaxf(fmaxf(fabsf(vb[0]), fabsf(vb[1])), fmaxf(fabsf(vb[2]), fabsf(vb[3]))));
    }
    if (diag) {  // column sums ride along on the diagonal items
#pragma unroll
      for (int m = 0; m < 4; ++m)
        if (q + 4 * m < na) atomicAdd(&csum[ja_o[m] >> 10], (unsigned long long)fixed_round((double)va[m] * scale));
    }
    // eight broadcast entries at a time: byte offset j8[k] and value d8[k] of entry k (entry k sits in lane k & 3, register k >> 2)
    int j8[8];
    double d8[8];
#define SCAMD_GRAM_BCAST2(K0, JLO, VLO)                                                              \
  j8[K0 + 0] = SCAMD_QB(0, JLO); d8[K0 + 0] = (double)__int_as_float(SCAMD_QB(0, __float_as_int(VLO))); \
  j8[K0 + 1] = SCAMD_QB(1, JLO); d8[K0 + 1] = (double)__int_as_float(SCAMD_QB(1, __float_as_int(VLO)))
#define SCAMD_GRAM_BCAST(CNT, JLO, VLO, JHI, VHI)                  \
  do {                                                             \
    SCAMD_GRAM_BCAST2(0, JLO, VLO);                                \
    if ((CNT) > 2) { j8[2] = SCAMD_QB(2, JLO); d8[2] = (double)__int_as_float(SCAMD_QB(2, __float_as_int(VLO))); \
                     j8[3] = SCAMD_QB(3, JLO); d8[3] = (double)__int_as_float(SCAMD_QB(3, __float_as_int(VLO))); } \
    if ((CNT) > 4) { SCAMD_GRAM_BCAST2(4, JHI, VHI); }             \
    if ((CNT) > 6) { j8[6] = SCAMD_QB(2, JHI); d8[6] = (double)__int_as_float(SCAMD_QB(2, __float_as_int(VHI))); \
                     j8[7] = SCAMD_QB(3, JHI); d8[7] = (double)__int_as_float(SCAMD_QB(3, __float_as_int(VHI))); } \
  } while (0)
    // `cnt` of the eight broadcast entries are real (wave-uniform): nested so that the first missing entry leaves the chain with
    // ONE forward branch, the likely side laid out in line (a taken branch costs as much as the product it guards)
#define SCAMD_GRAM_P(K, OFF, VS) add_at(j8[K] + (OFF), round_product(d8[K] * (VS)))
#define SCAMD_GRAM_CHAIN(CNT, OFF, VS)                                   \
  do {                                                                   \
    const int cnt_ = (CNT);                                              \
    if (__builtin_expect(cnt_ > 0, 1)) { SCAMD_GRAM_P(0, OFF, VS);       \
    if (__builtin_expect(cnt_ > 1, 1)) { SCAMD_GRAM_P(1, OFF, VS);       \
    if (__builtin_expect(cnt_ > 2, 1)) { SCAMD_GRAM_P(2, OFF, VS);       \
    if (__builtin_expect(cnt_ > 3, 1)) { SCAMD_GRAM_P(3, OFF, VS);       \
    if (__builtin_expect(cnt_ > 4, 1)) { SCAMD_GRAM_P(4, OFF, VS);       \
    if (__builtin_expect(cnt_ > 5, 1)) { SCAMD_GRAM_P(5, OFF, VS);       \
    if (__builtin_expect(cnt_ > 6, 1)) { SCAMD_GRAM_P(6, OFF, VS);       \
    if (__builtin_expect(cnt_ > 7, 1)) { SCAMD_GRAM_P(7, OFF, VS); } } } } } } } } \
  } while (0)
    // (1) the first eight entries of tile b stay in their lanes (two per lane), the entries of tile a are broadcast
    SCAMD_GRAM_BCAST(namax, ja_o[0], va[0], ja_o[1], va[1]);
    if (q < nb) {
      const double vbs = (double)vb[0] * scale;
      SCAMD_GRAM_CHAIN(namax, jb_o[0], vbs);
    }
    if (q + 4 < nb) {
      const double vbs = (double)vb[1] * scale;
      SCAMD_GRAM_CHAIN(namax, jb_o[1], vbs);
    }
    if (namax > 8) {
      if (na <= 8) ja_o[2] = 0, ja_o[3] = 0;
      SCAMD_GRAM_BCAST(namax - 8, ja_o[2], va[2], ja_o[3], va[3]);
      if (q < nb) {
        const double vbs = (double)vb[0] * scale;
        SCAMD_GRAM_CHAIN(namax - 8, jb_o[0], vbs);
      }
      if (q + 4 < nb) {
        const double vbs = (double)vb[1] * scale;
        SCAMD_GRAM_CHAIN(namax - 8, jb_o[1], vbs);
      }
    }
    // (2) entries 8 .. 15 of tile b: the roles exchanged -- the lane keeps its tile-a entries, these are broadcast
    if (nbmax > 8) {
      const int nb2max = min(nbmax, 16) - 8;
      SCAMD_GRAM_BCAST(nb2max, jb_o[2], vb[2], jb_o[3], vb[3]);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        if (m < 2 || namax > 8) {
          if (q + 4 * m < na && nb > 8) {
            const double vas = (double)va[m] * scale;
            SCAMD_GRAM_CHAIN(nb2max, ja_o[m], vas);
          }
        }
      }
    }
#undef SCAMD_GRAM_CHAIN
#undef SCAMD_GRAM_P
#undef SCAMD_GRAM_BCAST
#undef SCAMD_GRAM_BCAST2
    // (3) rows with more than sixteen entries in either tile: the remaining products from the CSR arrays (every lane of a
    // quad finds the row's two tile ranges by bisection -- a rare path)
    if (namax > REC || nbmax > REC) {
      const int64_t row = ((int64_t)(blk0 + (s >> 6))) * RB + prow;
      int64_t rb = 0;
      int len = 0;
      if (na > REC || nb > REC) {  // (padding rows behind the matrix have no indptr entry)
        rb = indptr[row];
        len = (int)(indptr[row + 1] - rb);
      }
      auto lower = [&](int col) -> int {  // first position of the row with column >= col
        int lo = 0, hi = len;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (indices[rb + mid] < col) lo = mid + 1;
          else hi = mid;
        }
        return lo;
      };
      const int pa_rel = lower(a0), pb_rel = lower(b0);
      const int na_f = lower(a0 + GT) - pa_rel, nb_f = lower(b0 + GT) - pb_rel;  // the full counts (the records cap them at 255)
      const int64_t pa = rb + pa_rel, pb = rb + pb_rel;
      const int na_fmax = wave_max_quads(na_f), nb_fmax = wave_max_quads(nb_f);
      if (diag)
        for (int ia = REC + q; ia < na_fmax; ia += 4)
          if (ia < na_f) atomicAdd(&csum[indices[pa + ia] - a0], (unsigned long long)fixed_round((double)data[pa + ia] * scale));
      for (int cb = 0; cb < nb_fmax; cb += 4) {
        const int ib = cb + q;
        const bool hb = ib < nb_f;
        int jbo = 0;
        double vs = 0.0;
        if (hb) {
          jbo = (indices[pb + ib] - b0) << 3;
          const float v = data[pb + ib];
          if constexpr (FAST) bmax = fmaxf(bmax, fabsf(v));
          vs = (double)v * scale;
        }
        for (int ia = (cb < REC ? REC : 0); ia < na_fmax; ++ia) {
          if (hb && ia < na_f) {
            const float v = data[pa + ia];
            if constexpr (FAST) amax = fmaxf(amax, fabsf(v));
            add_at(((indices[pa + ia] - a0) << 10) + jbo, round_product((double)v * vs));
          }
        }
      }
    }
  };
#undef SCAMD_QB

  if (steps > 0) {
    constexpr int WS = GRAM_THREADS / 64;  // the wave's stride over the steps
    // (the records two steps ahead, the sorted rows four: with one step of lead a third of the wave cycles waited for memory)
    unsigned short pr_cur = load_perm(wave), pr_1 = load_perm(wave + WS), pr_2 = load_perm(wave + 2 * WS),
                   pr_3 = load_perm(wave + 3 * WS);
    Ent en_cur = load_ent(row_offset(wave, pr_cur));
    Ent en_1 = load_ent(row_offset(wave + WS, pr_1));
    for (int s = wave; s < steps; s += WS) {
      const unsigned short pr_4 = load_perm(s + 4 * WS);
      const Ent en_2 = load_ent(row_offset(s + 2 * WS, pr_2));
      compute(en_cur, s, pr_cur);
      en_cur = en_1;
      en_1 = en_2;
      pr_cur = pr_1;
      pr_1 = pr_2;
      pr_2 = pr_3;
      pr_3 = pr_4;
    }
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

// what either kernel family needs: the packed records + per-tile row orders (g <= 64 tiles), or the round-2 tile pointers
struct GramCarve {
  uint2* ent = nullptr;
  unsigned short* perm = nullptr;
  unsigned char* cnt8 = nullptr;
  unsigned short* ptr = nullptr;
  unsigned int* mx = nullptr;
  bool packed = false;
};
static bool gram_use_packed(int64_t n, int64_t ntile) {
  static const bool legacy = [] {
    const char* e = getenv("SCAMD_GRAM_LEGACY");
    return e && e[0] == '1';
  }();
  const int64_t n_pad = (n + RB - 1) / RB * RB;
  return !legacy && ntile <= PACK_MAX_TILES && n_pad * REC * 8 < ((int64_t)1 << 32);
}
static GramCarve gram_carve(Workspace& ws, int64_t n, int64_t ntile) {
  GramCarve c;
  c.packed = gram_use_packed(n, ntile);
  const int64_t n_pad = (n + RB - 1) / RB * RB;
  if (c.packed) {
    c.ent = ws.take<uint2>((size_t)ntile * n_pad * REC);
    c.perm = ws.take<unsigned short>((size_t)ntile * n_pad);
    c.cnt8 = ws.take<unsigned char>((size_t)ntile * n_pad);
  } else {
    c.ptr = ws.take<unsigned short>((size_t)n * (ntile + 1));
  }
  c.mx = ws.take<unsigned int>(4);
  return c;
}
// The packed kernels are built around <= 8 entries per (row, tile) with a tail up to 16: measured at 200k x 2k, 6.4 / 10.2 /
// 15.4 / 25.6 entries per tile (5 / 8 / 12 / 20 % dense): 1.33 / 4.13 / 12.0 / 33.7 ms against 2.47 / 3.57 / 6.49 / 14.1 ms of
// the round-2 kernel (tools/gram_density_probe.py) -- denser matrices keep the round-2 kernel (its uint16 tile pointers fit
// the records' space).
static bool gram_sparse_enough(int64_t n, int64_t ntile, int64_t nnz) { return (double)nnz <= 8.0 * (double)n * (double)ntile; }

extern "C" size_t scamd_csr_gram_workspace_bytes(int64_t n, int64_t g) {
  if (n <= 0 || g <= 0) return 0;
  Workspace ws(nullptr, 0);
  (void)gram_carve(ws, n, (g + GT - 1) / GT);
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
  const int nblk = (int)((n + RB - 1) / RB);
  const int64_t n_pad = (int64_t)nblk * RB;
  Workspace ws(workspace, workspace_bytes);
  GramCarve cv = gram_carve(ws, n, ntile);
  unsigned int* mx = cv.mx;
  if (cv.packed && !gram_sparse_enough(n, ntile, nnz)) {
    cv.packed = false;
    cv.ptr = reinterpret_cast<unsigned short*>(cv.ent);  // n x (ntile + 1) uint16 <= the records' n_pad x ntile x 128 bytes
  }
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
  const int npair = ntile * (ntile + 1) / 2;
  const size_t lds = (size_t)(GT * GT + GT) * sizeof(unsigned long long);
  // column sums are scaled by 2^scale_bits as well (x * 2^S), products by 2^S: x_a * (x_b * 2^S)
  const double scale = std::ldexp(1.0, scale_bits);
  unsigned int* flag = mx + 1;  // products outside the fast rounding's range (see gram_tile_kernel)
  SCAMD_HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(unsigned int), s));
  unsigned long long* gram_u = reinterpret_cast<unsigned long long*>(gram);
  unsigned long long* colsum_u = reinterpret_cast<unsigned long long*>(colsum);
  // one workgroup per CU (the 128 KB tile), equal items: the launch runs in rounds of 256 workgroups, and a last round that
  // is half empty costs half a round -- 136 pairs x 16 chunks = 8.5 rounds: 6.90 ms, x 15 = 7.97 rounds: 6.57 ms
  // (profiles/r05v_gram_chunks.log; 17 chunks = 9.03 rounds: 7.15 ms).  The count near ~8 items per CU that wastes least:
  auto pick_chunks = [&](int64_t most, int per_cu = 8) -> int {  // most: the largest count that leaves a chunk ~4096 rows
    int n_chunks = (int)std::max<int64_t>(1, std::min<int64_t>(most, (256 * per_cu + npair - 1) / npair));
    const int hi = (int)std::min<int64_t>(most, n_chunks + 4);
    double best = 1e9;
    for (int c = std::max(1, n_chunks - 4); c <= hi; ++c) {
      const int64_t items = (int64_t)npair * c, rounds = (items + 255) / 256;
      const double waste = (double)(rounds * 256) / (double)items;
      if (waste < best - 1e-9) best = waste, n_chunks = c;
    }
    return n_chunks;
  };
  if (nnz == 0) {
    // nothing to add: the sums stay zero
  } else if (cv.packed) {
    hipLaunchKernelGGL(gram_pack_kernel, dim3((unsigned)ceil_div(n_pad, PACK_WAVES)), dim3(PACK_WAVES * 64), 0, s, indptr, indices,
                       data, n, ntile, cv.ent, cv.cnt8, n_pad);
    SCAMD_LAUNCH_CHECK();
    hipLaunchKernelGGL(gram_rank_kernel, dim3((unsigned)nblk), dim3(RB), 0, s, cv.cnt8, ntile, cv.perm, n_pad);
    SCAMD_LAUNCH_CHECK();
    // (the packed sweep likes ~24 items per CU, three times the round-2 kernel's: 8 / 15 / 30 / 45 / 60 / 98 / 163 chunks at 1M x 2k
    // take 4.86 / 4.44 / 4.36 / 4.20-4.26 / 4.24 / 4.56 / 5.30 ms for the entry -- shorter items even out the XCD queues' tails,
    // until the 128 KB flush per item shows)
    int n_chunks = pick_chunks(std::max(1, (nblk + 3) / 4), 24);
    if (const char* e = getenv("SCAMD_GRAM_CHUNKS")) n_chunks = std::max(1, std::min(nblk, atoi(e)));  // (A/B knob)
    const int blocks_per_chunk = (nblk + n_chunks - 1) / n_chunks;
    n_chunks = (nblk + blocks_per_chunk - 1) / blocks_per_chunk;
    auto gram_kernel = gram_quad_kernel<true>;
    auto gram_exact = gram_quad_kernel<false>;
    const size_t lds_p = lds + 16;
    SCAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gram_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p));
    SCAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gram_exact),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p));
    // launch slots: 8 XCD queues x (pairs of the XCD) x chunks (see gram_quad_kernel)
    const unsigned grid = 8u * (unsigned)ceil_div(npair, 8) * (unsigned)n_chunks;
    hipLaunchKernelGGL(gram_kernel, dim3(grid), dim3(GRAM_THREADS), lds_p, s, indptr, indices, data, ntile, cv.ent, cv.perm,
                       n_pad, blocks_per_chunk, nblk, n_chunks, scale, gram_u, ld_gram, colsum_u, flag);
    SCAMD_LAUNCH_CHECK();
    // no-ops unless the flag went up (no host round trip: the entry stays stream-ordered)
    hipLaunchKernelGGL(gram_clear_if_flagged_kernel, dim3(1024), dim3(256), 0, s, gram_u, gp * ld_gram, colsum_u, gp, flag);
    SCAMD_LAUNCH_CHECK();
    hipLaunchKernelGGL(gram_exact, dim3(grid), dim3(GRAM_THREADS), lds_p, s, indptr, indices, data, ntile, cv.ent, cv.perm,
                       n_pad, blocks_per_chunk, nblk, n_chunks, scale, gram_u, ld_gram, colsum_u, flag);
    SCAMD_LAUNCH_CHECK();
  } else {
    hipLaunchKernelGGL(gram_tileptr_kernel, dim3((unsigned)ceil_div(n, 4)), dim3(256), 0, s, indptr, indices, n, ntile, cv.ptr);
    SCAMD_LAUNCH_CHECK();
    int n_chunks = pick_chunks((n + 4095) / 4096);
    if (const char* e = getenv("SCAMD_GRAM_CHUNKS")) n_chunks = std::max(1, std::min<int>((int)((n + 255) / 256), atoi(e)));  // (A/B knob)
    const int rows_per_chunk = (int)((n + n_chunks - 1) / n_chunks);
    n_chunks = (int)((n + rows_per_chunk - 1) / rows_per_chunk);
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
    hipLaunchKernelGGL(gram_kernel, dim3((unsigned)npair, (unsigned)n_chunks), dim3(GRAM_THREADS), lds, s, indptr,
                       indices, data, n, ntile, cv.ptr, rows_per_chunk, scale, gram_u, ld_gram, colsum_u, flag);
    SCAMD_LAUNCH_CHECK();
    hipLaunchKernelGGL(gram_clear_if_flagged_kernel, dim3(1024), dim3(256), 0, s, gram_u, gp * ld_gram, colsum_u, gp, flag);
    SCAMD_LAUNCH_CHECK();
    hipLaunchKernelGGL(gram_exact, dim3((unsigned)npair, (unsigned)n_chunks), dim3(GRAM_THREADS), lds, s, indptr,
                       indices, data, n, ntile, cv.ptr, rows_per_chunk, scale, gram_u, ld_gram, colsum_u, flag);
    SCAMD_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(gram_mirror_kernel, dim3((unsigned)ceil_div(gp, 256), (unsigned)gp), dim3(256), 0, s,
                     reinterpret_cast<long long*>(gram), gp, ld_gram);
  SCAMD_LAUNCH_CHECK();
  return SCAMD_OK;
}
