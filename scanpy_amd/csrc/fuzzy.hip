// umap fuzzy simplicial set (connectivities) on gfx950.
//
// Replaces umap.umap_.fuzzy_simplicial_set(set_op_mix_ratio=1, local_connectivity=1) as called at
// src/scanpy/neighbors/_connectivity.py:103-138:
//   per row: rho = first positive distance, sigma by <=64-step bisection so that
//            sum_{j>=1} exp(-max(0, d_ij - rho)/sigma) = log2(k)   (float64 bisection on float32 data)
//   weights: w_ij = 0 (self) | 1 (d_ij - rho <= 0) | exp(-(d_ij - rho)/sigma)   (float32)
//   C = W + W^T - W o W^T, zeros eliminated, CSR with ascending column indices.
// HBM-bound, tiny: n*k*8 B in, <= 2*n*(k-1)*12 B out.
#include "common.h"
#include "scan.h"

#include <algorithm>

namespace scamd {

// Sum of all distances (the only global quantity of smooth_knn_dist: the floor of sigma on rows without a positive
// distance is 1e-3 * the mean of ALL distances), ORDER INDEPENDENT: the distances are added as 64-bit fixed-point
// integers scaled by 2^S, S chosen from the largest distance and the element count so that the sum cannot overflow.
// Integer addition is associative, so the value does not depend on the grid, on the atomics' order or on how the rows
// are sharded over ranks (the row-sharded path computes the same integers with torch and all-reduces them:
// scanpy_amd/_pipeline.py:fixed_point_distance_sum) -- a float64 atomic sum made the sharded rows differ from the
// single-device rows in the last bit whenever the floor was hit.
// (both reductions read 16 B per lane with four loads in flight: the scalar grid-stride loops ran at 0.55 TB/s, one
// dependent load per trip -- 109 us each for the 60 MB of the 1M x 15 problem)
struct FssSpan {  // [0, head) and [tail, total) scalar, [head, tail) as float4 (the pointer need not be 16-byte aligned)
  int64_t head, n4;
};
__device__ __forceinline__ FssSpan fss_span(const float* d, int64_t total) {
  FssSpan sp;
  sp.head = (int64_t)(((16u - (unsigned)(reinterpret_cast<uintptr_t>(d) & 15u)) & 15u) >> 2);
  if (sp.head > total) sp.head = total;
  sp.n4 = (total - sp.head) >> 2;
  return sp;
}
__global__ void fss_max_kernel(const float* __restrict__ d, int64_t total, unsigned int* __restrict__ mx_bits) {
  const FssSpan sp = fss_span(d, total);
  const float4* d4 = reinterpret_cast<const float4*>(d + sp.head);
  float m = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < sp.n4; i += 4 * stride) {
    const float4 a = d4[i], b = d4[i + stride], c = d4[i + 2 * stride], e = d4[i + 3 * stride];
    m = fmaxf(m, fmaxf(fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)), fmaxf(fmaxf(b.x, b.y), fmaxf(b.z, b.w))));
    m = fmaxf(m, fmaxf(fmaxf(fmaxf(c.x, c.y), fmaxf(c.z, c.w)), fmaxf(fmaxf(e.x, e.y), fmaxf(e.z, e.w))));
  }
  for (; i < sp.n4; i += stride) {
    const float4 a = d4[i];
    m = fmaxf(m, fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)));
  }
  if (blockIdx.x == 0) {  // the unaligned ends
    for (int64_t q = threadIdx.x; q < sp.head; q += blockDim.x) m = fmaxf(m, d[q]);
    for (int64_t q = sp.head + 4 * sp.n4 + threadIdx.x; q < total; q += blockDim.x) m = fmaxf(m, d[q]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  // one atomic per WORKGROUP (a same-address atomic per wave -- 8k of them -- was the kernel's duration, not its loads)
  __shared__ float wm[16];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = fmaxf(m, wm[w]);
    // (distances are >= 0: the bit patterns of non-negative floats order like the values)
    if (m > 0.f) atomicMax(mx_bits, __float_as_uint(m));
  }
}
// S = 61 - e - ceil(log2(total)) with max < 2^e: every term is < 2^(61 - ceil(log2 total)), the sum < 2^61
__device__ __forceinline__ int fss_sum_scale_bits(float mx, int64_t total) {
  int e = 0;
  (void)frexpf(mx > 0.f ? mx : 1.f, &e);
  int lg = 0;
  while (((int64_t)1 << lg) < total) ++lg;
  return 61 - e - lg;
}
__global__ void fss_sum_kernel(const float* __restrict__ d, int64_t total, const unsigned int* __restrict__ mx_bits,
                               unsigned long long* __restrict__ isum) {
  const double scale = ldexp(1.0, fss_sum_scale_bits(__uint_as_float(*mx_bits), total));
  const FssSpan sp = fss_span(d, total);
  const float4* d4 = reinterpret_cast<const float4*>(d + sp.head);
  long long s = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
#define FSS_ADD4(v) \
  s += llrint((double)(v).x * scale) + llrint((double)(v).y * scale) + llrint((double)(v).z * scale) + \
       llrint((double)(v).w * scale)
  for (; i + 3 * stride < sp.n4; i += 4 * stride) {
    const float4 a = d4[i], b = d4[i + stride], c = d4[i + 2 * stride], e = d4[i + 3 * stride];
    FSS_ADD4(a);
    FSS_ADD4(b);
    FSS_ADD4(c);
    FSS_ADD4(e);
  }
  for (; i < sp.n4; i += stride) {
    const float4 a = d4[i];
    FSS_ADD4(a);
  }
#undef FSS_ADD4
  if (blockIdx.x == 0) {
    for (int64_t q = threadIdx.x; q < sp.head; q += blockDim.x) s += llrint((double)d[q] * scale);
    for (int64_t q = sp.head + 4 * sp.n4 + threadIdx.x; q < total; q += blockDim.x) s += llrint((double)d[q] * scale);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  __shared__ long long ws[16];
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) s += ws[w];
    if (s != 0) atomicAdd(isum, (unsigned long long)s);
  }
}
__global__ void fss_sum_final_kernel(const unsigned long long* __restrict__ isum, const unsigned int* __restrict__ mx_bits,
                                     int64_t total, double* __restrict__ sum) {
  sum[0] = ldexp((double)(long long)isum[0], -fss_sum_scale_bits(__uint_as_float(*mx_bits), total));
}

// one thread per row
// (row shards: the rows are rows row_begin .. row_begin + n - 1 of an n_total-row problem; idx holds GLOBAL row ids and
// *sum_all is the sum of all n_total * k distances)
//
// The bisection is umap's, decision for decision, but most of its exponentials are float32.  The reference evaluates
// psum(mid) = sum_j exp(-(d_j - rho) / mid) in float64 and looks at two things only: |psum - target| < 1e-5 (stop) and
// psum > target (direction).  A float32 evaluation with float64 accumulation is within E = (k - 1) * 1e-6 of the float64
// one (per term: the rounded 1 / mid and the product move the argument x by 1.2e-7 relative, x e^-x <= 0.37; v_exp_f32 and
// its log2(e) scaling add 2 ulp of a value <= 1: below 2e-7 in all), so whenever |psum32 - target| > 1e-5 + E both
// answers are those of the float64 evaluation.  Phase 1 bisects on float32 sums until a row first comes closer than
// that; phase 2 re-evaluates that step and every later one in float64 exactly as before.  Rows typically take 15-25
// steps, the last one to three of them in phase 2: the kernel spent 0.99 ms in float64 exp at 1M x 15.
// KREG > 0: k <= KREG and the row's d_j - rho live in registers (the loop re-read them through the L1, a 60-byte stride
// between lanes: 30 cache lines per load instruction).
template <int KREG>
__global__ void fss_sigma_kernel(const int* __restrict__ idx, const float* __restrict__ dist, int64_t n,
                                 int k, const double* __restrict__ sum_all, float* __restrict__ sigma_out,
                                 float* __restrict__ rho_out, float* __restrict__ w,
                                 int* __restrict__ outcnt, int64_t row_begin, int64_t n_total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int self_id = (int)(row_begin + i);
  const float* di = dist + i * k;
  const int* ii = idx + i * k;
  float rho = 0.f;
  double rowsum = 0.0;
  bool found = false;
  float dreg[KREG > 0 ? KREG : 1];
  if (KREG > 0) {
#pragma unroll
    for (int j = 0; j < KREG; ++j) {
      const float v = j < k ? di[j] : 0.f;  // (a pad column adds 0 to the sum and is never the first positive one)
      dreg[j] = v;
      rowsum += (double)v;
      if (!found && v > 0.f) {
        rho = v;
        found = true;
      }
    }
  } else {
    for (int j = 0; j < k; ++j) {
      const float v = di[j];
      rowsum += (double)v;
      if (!found && v > 0.f) {
        rho = v;
        found = true;
      }
    }
  }
  float dmax = 0.f;  // largest d_j - rho of the row (j >= 1)
  if (KREG > 0) {
#pragma unroll
    for (int j = 0; j < KREG; ++j) {
      dreg[j] = j < k ? __fsub_rn(dreg[j], rho) : -1.f;  // (pad columns: never > 0, never counted -- see `j < k` below)
      if (j >= 1 && j < k) dmax = fmaxf(dmax, dreg[j]);
    }
  } else {
    for (int j = 1; j < k; ++j) dmax = fmaxf(dmax, __fsub_rn(di[j], rho));
  }
  const double target = log2((double)k);
  const double undecided = 1e-5 + 1e-6 * (double)(k - 1);
  double lo = 0.0, hi = INFINITY, mid = 1.0;
  int it = 0;
  bool done = false;
  // phase 1: float32 terms
  if (dmax < 1e30f) {
    for (; it < 64; ++it) {
      if (!(mid > 1e-30 && mid < 1e30)) break;
      const float inv = (float)(1.0 / mid);
      double psum = 0.0;
      if (KREG > 0) {
#pragma unroll
        for (int j = 1; j < KREG; ++j)
          if (j < k) psum += (double)(dreg[j] > 0.f ? __expf(-(dreg[j] * inv)) : 1.f);
      } else {
        for (int j = 1; j < k; ++j) {
          const float df = __fsub_rn(di[j], rho);
          psum += (double)(df > 0.f ? __expf(-(df * inv)) : 1.f);
        }
      }
      if (fabs(psum - target) <= undecided) break;  // too close for float32: this step again, in float64
      if (psum > target) {
        hi = mid;
        mid = (lo + hi) / 2.0;
      } else {
        lo = mid;
        if (hi == INFINITY) mid *= 2.0;
        else mid = (lo + hi) / 2.0;
      }
    }
  }
  // phase 2: float64, the reference's loop from step `it` on
  for (; it < 64 && !done; ++it) {
    double psum = 0.0;
    if (KREG > 0) {
#pragma unroll
      for (int j = 1; j < KREG; ++j)
        if (j < k) {
          const double dd = (double)dreg[j];
          psum += (dd > 0.0) ? exp(-(dd / mid)) : 1.0;
        }
    } else {
      for (int j = 1; j < k; ++j) {
        const double dd = (double)__fsub_rn(di[j], rho);
        psum += (dd > 0.0) ? exp(-(dd / mid)) : 1.0;
      }
    }
    if (fabs(psum - target) < 1e-5) {
      done = true;
      break;
    }
    if (psum > target) {
      hi = mid;
      mid = (lo + hi) / 2.0;
    } else {
      lo = mid;
      if (hi == INFINITY) mid *= 2.0;
      else mid = (lo + hi) / 2.0;
    }
  }
  float sigma = (float)mid;
  if (rho > 0.f) {
    double mean_i = rowsum / (double)k;
    if ((double)sigma < 1e-3 * mean_i) sigma = (float)(1e-3 * mean_i);
  } else {
    double mean_all = *sum_all / ((double)n_total * (double)k);
    if ((double)sigma < 1e-3 * mean_all) sigma = (float)(1e-3 * mean_all);
  }
  if (sigma_out) sigma_out[i] = sigma;
  if (rho_out) rho_out[i] = rho;
  int cnt = 0;
  for (int j = 0; j < k; ++j) {
    int t = ii[j];
    float val;
    float df = __fsub_rn(di[j], rho);
    if (t == self_id || t < 0) val = 0.f;
    else if (df <= 0.f || sigma == 0.f) val = 1.f;
    else val = (float)exp(-(double)__fdiv_rn(df, sigma));  // correctly rounded, keeps float32 subnormals
    w[i * k + j] = val;
    cnt += (val > 0.f) ? 1 : 0;
  }
  outcnt[i] = cnt;
}
static void launch_sigma(const int* idx, const float* dist, int64_t n, int k, const double* sum_all, float* sigma_out,
                         float* rho_out, float* w, int* outcnt, int64_t row_begin, int64_t n_total, hipStream_t s) {
  if (k <= 16)
    hipLaunchKernelGGL(fss_sigma_kernel<16>, dim3(ceil_div(n, 128)), dim3(128), 0, s, idx, dist, n, k, sum_all, sigma_out,
                       rho_out, w, outcnt, row_begin, n_total);
  else
    hipLaunchKernelGGL(fss_sigma_kernel<0>, dim3(ceil_div(n, 128)), dim3(128), 0, s, idx, dist, n, k, sum_all, sigma_out,
                       rho_out, w, outcnt, row_begin, n_total);
}

// Records of the reverse-edge lookup: row t's neighbour list as KP (id, weight) pairs in ONE aligned block (128 B for
// k <= 16, pad pairs hold id -1).  The lookup "is i among t's neighbours, with which weight" is a random gather per
// directed slot and the kernel's duration was that of its cache-line traffic: the two k-long rows of idx and w, 60 B
// each at a 60 B stride, touch 1.47 lines of 128 B each on average -- 44M lines for the 15M slots of 1M x 15, 0.98 ms;
// one record per slot is a third of that.
template <int KP>
__global__ __launch_bounds__(256) void fss_pack_kernel(const int* __restrict__ idx, const float* __restrict__ w,
                                                       int64_t n, int k, int2* __restrict__ rec) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t row = q / KP;
  const int sub = (int)(q - row * KP);
  if (row >= n) return;
  int2 p = make_int2(-1, 0);
  if (sub < k) p = make_int2(idx[row * k + sub], __float_as_int(w[row * k + sub]));
  rec[q] = p;
}

// KP lanes per group, SPG consecutive directed slots (i, j) per group: weight of the reverse edge, in-only degree of the
// target.  The kernel is bound by latency x occupancy, not by traffic (one 128-byte record per slot instead of two
// 60-byte rows did not change its 1.0 ms): with one slot per group a wave made two dependent round trips (w / idx, then
// the record) for 4 slots, and 3.75M such waves at 8 per SIMD take ~3 us each.  Now the SPG slot loads of a group are
// issued together, then its SPG record gathers: a quarter of the waves, four gathers in flight per lane.
// (Before records: a thread per slot walked the target's list entry by entry -- 15 serial gathers per thread, every one
// of them 64 different cache lines per wave: 1.6 ms at 1M x 15.)
// value of the lane CTRL & 15 places to the left in the same row of 16 lanes (DPP row_ror)
template <int CTRL>
__device__ __forceinline__ float row_ror_f32(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false));
}
template <int KP, int SPG>
__global__ __launch_bounds__(256) void fss_recip_rec_kernel(const int* __restrict__ idx, const float* __restrict__ w,
                                                            const int2* __restrict__ rec, int64_t n, int k,
                                                            float* __restrict__ recw, int* __restrict__ in_only) {
  const int sub = threadIdx.x & (KP - 1);
  const int64_t e0 = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / KP) * SPG;
  const int64_t total = n * k;
  float we[SPG];
  int t[SPG];
#pragma unroll
  for (int q = 0; q < SPG; ++q) {
    const bool live = e0 + q < total;
    we[q] = live ? w[e0 + q] : 0.f;
    t[q] = live ? idx[e0 + q] : 0;
  }
  int2 p[SPG];
#pragma unroll
  for (int q = 0; q < SPG; ++q)  // (absent slots read record 0: no branch around the gather)
    p[q] = rec[(int64_t)(we[q] > 0.f ? t[q] : 0) * KP + sub];
  // row of slot e0 by ONE division (32-bit when the slot count allows: a 64-bit division is ~100 instructions, four of
  // them per lane were a third of the kernel's instruction count), rows of the following slots by counting up
  int i0, j0;
  if (total < ((int64_t)1 << 31)) {
    i0 = (int)((unsigned int)e0 / (unsigned int)k);
    j0 = (int)((unsigned int)e0 - (unsigned int)i0 * (unsigned int)k);
  } else {
    i0 = (int)(e0 / k);
    j0 = (int)(e0 - (int64_t)i0 * k);
  }
  float r[SPG];
#pragma unroll
  for (int q = 0; q < SPG; ++q) {
    int i = i0, j = j0 + q;
    while (j >= k) {  // (k >= 2, q < SPG: at most SPG / 2 + 1 trips)
      j -= k;
      ++i;
    }
    r[q] = (we[q] > 0.f && p[q].x == i) ? __int_as_float(p[q].y) : 0.f;
  }
  // maximum over the KP lanes of the group, in every lane: rotations inside a row of 16 lanes are DPP modifiers of the
  // v_max itself (one instruction each; __shfl_xor is a ds_bpermute round trip through the LDS crossbar)
#pragma unroll
  for (int q = 0; q < SPG; ++q) {
    if (KP == 32) r[q] = fmaxf(r[q], __shfl_xor(r[q], 16));
    r[q] = fmaxf(r[q], row_ror_f32<0x128>(r[q]));
    r[q] = fmaxf(r[q], row_ror_f32<0x124>(r[q]));
    r[q] = fmaxf(r[q], row_ror_f32<0x122>(r[q]));
    r[q] = fmaxf(r[q], row_ror_f32<0x121>(r[q]));
  }
  // lane q of the group writes slot q
  float my_r = 0.f, my_w = 0.f;
  int my_t = 0;
#pragma unroll
  for (int q = 0; q < SPG; ++q)
    if (sub == q) {
      my_r = r[q];
      my_w = we[q];
      my_t = t[q];
    }
  if (sub < SPG && e0 + sub < total) {
    if (my_w > 0.f && my_r == 0.f) atomicAdd(&in_only[my_t], 1);
    recw[e0 + sub] = my_r;
  }
}

// lists longer than a record (k > 32): 16 lanes per slot walk the target's rows of idx and w
__global__ __launch_bounds__(256) void fss_recip_kernel(const int* __restrict__ idx, const float* __restrict__ w,
                                                        int64_t n, int k, float* __restrict__ recw,
                                                        int* __restrict__ in_only) {
  const int sub = threadIdx.x & 15;
  const int64_t e = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const bool live = e < n * k;
  const float we = live ? w[e] : 0.f;
  const int t = live ? idx[e] : 0;
  float r = 0.f;
  if (we > 0.f) {
    const int i = (int)(e / k);
    const int* ti = idx + (int64_t)t * k;
    const float* tw = w + (int64_t)t * k;
    for (int j = sub; j < k; j += 16)
      if (ti[j] == i) r = fmaxf(r, tw[j]);
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) r = fmaxf(r, __shfl_xor(r, o));
  if (live && sub == 0) {
    if (we > 0.f && r == 0.f) atomicAdd(&in_only[t], 1);
    recw[e] = r;
  }
}

__global__ void fss_rowcount_kernel(const int* __restrict__ outcnt, const int* __restrict__ in_only, int64_t n,
                                    int* __restrict__ rowcnt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) rowcnt[i] = outcnt[i] + in_only[i];
}

// where the next in-only entry of row t goes: after the row's own (out) entries.  One 64-bit counter per row, so that
// the slot that appends to row t needs one atomic and nothing else of row t (it used to gather indptr[t], outcnt[t] and
// a 32-bit cursor: three random reads per appended entry).
__global__ void fss_cursor_kernel(const int64_t* __restrict__ indptr, const int* __restrict__ outcnt, int64_t n,
                                  unsigned long long* __restrict__ cursor) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) cursor[i] = (unsigned long long)(indptr[i] + outcnt[i]);
}

// unsorted rows: (column, value bits) pairs, so that an appended entry is one 8-byte store
__global__ void fss_fill_kernel(const int* __restrict__ idx, const float* __restrict__ w,
                                const float* __restrict__ recw, const int64_t* __restrict__ indptr, int64_t n, int k,
                                unsigned long long* __restrict__ cursor, int2* __restrict__ tmp, int mode) {
  // position of the slot among the stored (w > 0) slots of its row: a ballot over the wave's 64 consecutive slots, plus
  // -- for the row cut by the wave's first lane -- a ballot over the 64 slots before them (rows longer than that: the
  // entry-by-entry count).  The count used to be a loop of up to k - 1 serial loads per thread.
  const int lane = threadIdx.x & 63;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t e_wave = e - lane;
  const bool live = e < n * k;
  const float we = live ? w[e] : 0.f;
  const float w_prev = (e >= 64 && e - 64 < n * k) ? w[e - 64] : 0.f;
  const unsigned long long m_cur = __ballot(we > 0.f), m_prev = __ballot(w_prev > 0.f);
  if (!(we > 0.f)) return;
  const int64_t i = e / k;
  const int j = (int)(e - i * k);
  int before;
  if (j <= lane) {  // the row starts inside this wave's slots
    before = __popcll(m_cur & ((1ull << lane) - 1ull) & ~((1ull << (lane - j)) - 1ull));
  } else if (j - lane <= 64) {  // ... inside the 64 slots before them
    const int back = j - lane;  // slots of the row before e_wave
    before = __popcll(m_cur & ((1ull << lane) - 1ull)) + __popcll(back == 64 ? m_prev : m_prev >> (64 - back));
  } else {
    before = __popcll(m_cur & ((1ull << lane) - 1ull));
    for (int64_t q = i * k; q < e_wave; ++q) before += (w[q] > 0.f) ? 1 : 0;
  }
  const int t = idx[e];
  const float r = recw[e];
  const int64_t p = indptr[i] + before;
  // mode 0 (umap): (w + w^T) - w*w^T with one rounding per operation, as scipy's float32 sparse arithmetic does
  // mode 1 (gauss): the weight is symmetric by construction; a missing reverse entry is filled in with it
  // mode 2 (jaccard): (w + w^T) / 2
  const float val = mode == 0 ? __fsub_rn(__fadd_rn(we, r), __fmul_rn(we, r))
                              : (mode == 1 ? fmaxf(we, r) : 0.5f * (we + r));
  tmp[p] = make_int2(t, __float_as_int(val));
  if (r == 0.f) {
    const int64_t pt = (int64_t)atomicAdd(&cursor[t], 1ull);
    tmp[pt] = make_int2((int)i, __float_as_int(mode == 2 ? 0.5f * we : we));
  }
}

// Rows -> ascending column order (columns are unique within a row): every entry counts the entries of its row with a
// smaller column.  A workgroup takes SR_ROWS consecutive rows; when their entries fit the LDS block (all but the
// workgroups that hold a hub row) they are loaded once, coalesced, and ranked a THREAD PER ENTRY against the row in LDS
// (neighbouring lanes sit in the same row and read the same word: broadcasts).  The kernel was a wave per row ranking
// through readlane -- 25 of 64 lanes busy and a million waves of three dependent memory round trips each: 1.36 ms at
// 1M x 15, the longest of the five kernels of the stage.
constexpr int SR_ROWS = 32;
constexpr int SR_CAP = 2048;

// one wave, one row (hub rows: the symmetrised kNN graph of the 1M planted matrix has rows of 1.4k entries): 64 entries
// are ranked at a time against the row read in coalesced chunks of 64, compared from registers
__device__ __forceinline__ void fss_sort_row_wave(int64_t base, int len, int lane, const int2* __restrict__ tmp,
                                                  int* __restrict__ out_col, float* __restrict__ out_val) {
  for (int e0 = 0; e0 < len; e0 += 64) {
    const int e = e0 + lane;
    const int2 ce = e < len ? tmp[base + e] : make_int2(0x7fffffff, 0);
    const int c = ce.x;
    int rank = 0;
    for (int u0 = 0; u0 < len; u0 += 64) {
      const int cu_l = u0 + lane < len ? tmp[base + u0 + lane].x : 0x7fffffff;
      const int cnt = min(64, len - u0);
      for (int t = 0; t < cnt; ++t) {
        const int cu = __builtin_amdgcn_readlane(cu_l, t);
        rank += (cu < c || (cu == c && u0 + t < e)) ? 1 : 0;
      }
    }
    if (e < len) {
      out_col[base + rank] = c;
      out_val[base + rank] = __int_as_float(ce.y);
    }
  }
}

__global__ __launch_bounds__(256) void fss_sortrows_kernel(const int64_t* __restrict__ indptr, int64_t n,
                                                           const int2* __restrict__ tmp, int* __restrict__ out_col,
                                                           float* __restrict__ out_val) {
  __shared__ int s_col[SR_CAP];
  __shared__ int s_val[SR_CAP];
  __shared__ int s_start[SR_ROWS + 1];
  const int64_t r0 = (int64_t)blockIdx.x * SR_ROWS;
  const int nr = n - r0 < SR_ROWS ? (int)(n - r0) : SR_ROWS;
  const int64_t base = indptr[r0];
  const int64_t total64 = indptr[r0 + nr] - base;
  if (total64 <= SR_CAP) {  // (uniform over the workgroup)
    const int total = (int)total64;
    if ((int)threadIdx.x <= nr) s_start[threadIdx.x] = (int)(indptr[r0 + threadIdx.x] - base);
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
      const int2 ce = tmp[base + e];
      s_col[e] = ce.x;
      s_val[e] = ce.y;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
      int lo = 0, hi = nr;  // s_start[lo] <= e < s_start[hi]
      while (hi - lo > 1) {
        const int m = (lo + hi) >> 1;
        if (s_start[m] <= e) lo = m;
        else hi = m;
      }
      const int b = s_start[lo], en = s_start[lo + 1];
      const int c = s_col[e];
      int rank = 0;
      for (int u = b; u < en; ++u) {
        const int cu = s_col[u];
        rank += (cu < c || (cu == c && u < e)) ? 1 : 0;
      }
      out_col[base + b + rank] = c;
      out_val[base + b + rank] = __int_as_float(s_val[e]);
    }
    return;
  }
  const int lane = threadIdx.x & 63;
  for (int r = threadIdx.x >> 6; r < nr; r += (int)(blockDim.x >> 6)) {
    const int64_t rb = indptr[r0 + r];
    fss_sort_row_wave(rb, (int)(indptr[r0 + r + 1] - rb), lane, tmp, out_col, out_val);
  }
}

// ---- method='gauss' (src/scanpy/neighbors/_connectivity.py:21-100, CSR branch) -------------------------------------
// sigma_i^2 = median of the squared distances to the row's stored neighbours (self column excluded)
__global__ void gauss_sigma_kernel(const int* __restrict__ idx, const float* __restrict__ dist, int64_t n, int k,
                                   double* __restrict__ sigma_sq) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int* ri = idx + i * k;
  const float* di = dist + i * k;
  int m = 0;
  for (int j = 0; j < k; ++j) m += (ri[j] != (int)i && ri[j] >= 0) ? 1 : 0;
  // order statistics by rank counting (k is small): numpy's median = mean of the two middle values for even m
  const int lo_rank = (m - 1) / 2, hi_rank = m / 2;
  double lo = 0.0, hi = 0.0;
  for (int j = 0; j < k; ++j) {
    if (ri[j] == (int)i || ri[j] < 0) continue;
    const double v = (double)di[j] * (double)di[j];
    int rank = 0;
    for (int u = 0; u < k; ++u) {
      if (ri[u] == (int)i || ri[u] < 0) continue;
      const double vu = (double)di[u] * (double)di[u];
      rank += (vu < v || (vu == v && u < j)) ? 1 : 0;
    }
    if (rank == lo_rank) lo = v;
    if (rank == hi_rank) hi = v;
  }
  sigma_sq[i] = m > 0 ? 0.5 * (lo + hi) : 0.0;
}

__global__ void gauss_weight_kernel(const int* __restrict__ idx, const float* __restrict__ dist, int64_t n, int k,
                                    const double* __restrict__ sigma_sq, float* __restrict__ w,
                                    int* __restrict__ outcnt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double si2 = sigma_sq[i], si = sqrt(si2);
  int cnt = 0;
  for (int j = 0; j < k; ++j) {
    const int t = idx[i * k + j];
    float val = 0.f;
    if (t != (int)i && t >= 0) {
      const double sj2 = sigma_sq[t], den = si2 + sj2;
      const double d2 = (double)dist[i * k + j] * (double)dist[i * k + j];
      val = den > 0.0 ? (float)(sqrt(2.0 * si * sqrt(sj2) / den) * exp(-d2 / den)) : 0.f;
    }
    w[i * k + j] = val;
    cnt += (val > 0.f) ? 1 : 0;
  }
  outcnt[i] = cnt;
}

// ---- method='jaccard' (src/scanpy/neighbors/_connectivity.py:141-186) ------------------------------------------------
// w_ij = s / (2 (k - 1) - s), s = |N(i) & N(j)| over the neighbour lists without their self columns
__global__ void jaccard_weight_kernel(const int* __restrict__ idx, int64_t n, int k, float* __restrict__ w) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * k) return;
  const int64_t i = e / k;
  const int t = idx[e];
  float val = 0.f;
  if (t != (int)i && t >= 0) {
    const int* ri = idx + i * k;
    const int* rt = idx + (int64_t)t * k;
    int shared = 0;
    for (int a = 0; a < k; ++a) {
      const int u = ri[a];
      if (u == (int)i || u < 0) continue;
      for (int b2 = 0; b2 < k; ++b2) shared += (rt[b2] == u && rt[b2] != t) ? 1 : 0;
    }
    val = (float)((double)shared / (double)(2 * (k - 1) - shared));
  }
  w[e] = val;
}
__global__ void count_positive_rows_kernel(const float* __restrict__ w, int64_t n, int k, int* __restrict__ outcnt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int cnt = 0;
  for (int j = 0; j < k; ++j) cnt += (w[i * k + j] > 0.f) ? 1 : 0;
  outcnt[i] = cnt;
}

struct FuzzyBuffers {
  float* w; float* recw; int* outcnt; int* in_only; unsigned long long* cursor; int* rowcnt; int64_t* scan_tmp;
  int2* tmp; double* sum;
};

// pairs per record of the reverse-edge lookup (0: lists too long for records)
static int fuzzy_record_pairs(int k) { return k <= 16 ? 16 : (k <= 32 ? 32 : 0); }

static void fuzzy_carve(Workspace& ws, int64_t n, int k, FuzzyBuffers* b) {
  // `tmp`: the unsorted rows, <= 2 n (k - 1) entries -- and, before they are written, the n records of the lookup
  const size_t cap = std::max((size_t)2 * n * (k > 1 ? k - 1 : 1), (size_t)n * fuzzy_record_pairs(k));
  b->w = ws.take<float>((size_t)n * k);
  b->recw = ws.take<float>((size_t)n * k);
  b->outcnt = ws.take<int>((size_t)n);
  b->in_only = ws.take<int>((size_t)n);
  b->cursor = ws.take<unsigned long long>((size_t)n);
  b->rowcnt = ws.take<int>((size_t)n);
  b->scan_tmp = ws.take<int64_t>((size_t)scan_num_blocks(n) + 2);
  b->tmp = ws.take<int2>(cap);
  b->sum = ws.take<double>(4);
}

// directed weights w (> 0 = present) on the kNN pattern -> symmetric CSR with sorted rows; mode = combine rule of
// fss_fill_kernel.  b.in_only must be zero.
static int symmetrise(const FuzzyBuffers& b, const int32_t* knn_idx, int64_t n, int k, int mode, int64_t* out_indptr,
                      int32_t* out_indices, float* out_data, int64_t* nnz_host, hipStream_t s) {
  const int64_t total = n * k;
  const int kp = fuzzy_record_pairs(k);
  if (kp == 16) {
    hipLaunchKernelGGL(fss_pack_kernel<16>, dim3(ceil_div(n * 16, 256)), dim3(256), 0, s, knn_idx, b.w, n, k, b.tmp);
    SCAMD_LAUNCH_CHECK();
    hipLaunchKernelGGL((fss_recip_rec_kernel<16, 4>), dim3(ceil_div(ceil_div(total, 4) * (int64_t)16, 256)), dim3(256), 0, s,
                       knn_idx, b.w, b.tmp, n, k, b.recw, b.in_only);
  } else if (kp == 32) {
    hipLaunchKernelGGL(fss_pack_kernel<32>, dim3(ceil_div(n * 32, 256)), dim3(256), 0, s, knn_idx, b.w, n, k, b.tmp);
    SCAMD_LAUNCH_CHECK();
    hipLaunchKernelGGL((fss_recip_rec_kernel<32, 4>), dim3(ceil_div(ceil_div(total, 4) * (int64_t)32, 256)), dim3(256), 0, s,
                       knn_idx, b.w, b.tmp, n, k, b.recw, b.in_only);
  } else {
    hipLaunchKernelGGL(fss_recip_kernel, dim3(ceil_div(total, 16)), dim3(256), 0, s, knn_idx, b.w, n, k, b.recw,
                       b.in_only);
  }
  SCAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(fss_rowcount_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, b.outcnt, b.in_only, n,
                     b.rowcnt);
  SCAMD_LAUNCH_CHECK();
  int rc = exclusive_scan_i32_i64(b.rowcnt, n, out_indptr, b.scan_tmp, s);
  if (rc != SCAMD_OK) return rc;
  hipLaunchKernelGGL(fss_cursor_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, out_indptr, b.outcnt, n, b.cursor);
  SCAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(fss_fill_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, s, knn_idx, b.w, b.recw, out_indptr, n,
                     k, b.cursor, b.tmp, mode);
  SCAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(fss_sortrows_kernel, dim3(ceil_div(n, SR_ROWS)), dim3(256), 0, s, out_indptr, n, b.tmp,
                     out_indices, out_data);
  SCAMD_LAUNCH_CHECK();
  int64_t nnz = 0;
  SCAMD_READBACK_NOW(&nnz, out_indptr + n, sizeof(int64_t), s);
  *nnz_host = nnz;
  return SCAMD_OK;
}

}  // namespace scamd

using namespace scamd;

extern "C" size_t scamd_fuzzy_workspace_bytes(int64_t n, int k) {
  if (n <= 0 || k <= 0) return 0;
  Workspace ws(nullptr, 0);
  FuzzyBuffers b;
  fuzzy_carve(ws, n, k, &b);
  (void)ws.take<double>((size_t)n);  // sigma^2 of the gauss kernel: one size serves the three connectivity kernels
  return ws.used();
}

extern "C" int scamd_fuzzy_simplicial_set_f32(const int32_t* knn_idx, const float* knn_dist, int64_t n, int k,
                                              int64_t* out_indptr, int32_t* out_indices, float* out_data,
                                              int64_t cap, float* out_sigma, float* out_rho,
                                              int64_t* nnz_host, void* workspace, size_t workspace_bytes,
                                              scamd_stream_t stream) {
  SCAMD_REQUIRE(knn_idx && knn_dist && out_indptr && out_indices && out_data && nnz_host, SCAMD_EINVAL,
                "fuzzy: null pointer");
  SCAMD_REQUIRE(n >= 1 && k >= 2 && k <= 1024, SCAMD_EINVAL, "fuzzy: bad shape n=%lld k=%d", (long long)n, k);
  SCAMD_REQUIRE(n < (int64_t)1 << 31, SCAMD_EUNSUPPORTED, "fuzzy: n exceeds int32 ids");
  SCAMD_REQUIRE(cap >= 2 * n * (k - 1), SCAMD_ECAPACITY, "fuzzy: cap %lld < 2*n*(k-1) = %lld", (long long)cap,
                (long long)(2 * n * (k - 1)));
  Workspace ws(workspace, workspace_bytes);
  FuzzyBuffers b;
  fuzzy_carve(ws, n, k, &b);
  SCAMD_REQUIRE(workspace && ws.ok, SCAMD_EWORKSPACE, "fuzzy: workspace %zu < required %zu", workspace_bytes,
                ws.used());
  hipStream_t s = stream;
  const int64_t total = n * k;
  SCAMD_HIP_CHECK(hipMemsetAsync(b.sum, 0, 32, s));  // [0] sum (double), [1] fixed-point sum, [2] bits of the max
  SCAMD_HIP_CHECK(hipMemsetAsync(b.in_only, 0, sizeof(int) * n, s));
  {
    int blocks = (int)std::min<int64_t>((total + 255) / 256, 1024);
    unsigned long long* isum = reinterpret_cast<unsigned long long*>(b.sum + 1);
    unsigned int* mx_bits = reinterpret_cast<unsigned int*>(b.sum + 2);
    hipLaunchKernelGGL(fss_max_kernel, dim3(blocks), dim3(256), 0, s, knn_dist, total, mx_bits);
    SCAMD_LAUNCH_CHECK();
    hipLaunchKernelGGL(fss_sum_kernel, dim3(blocks), dim3(256), 0, s, knn_dist, total, mx_bits, isum);
    SCAMD_LAUNCH_CHECK();
    hipLaunchKernelGGL(fss_sum_final_kernel, dim3(1), dim3(1), 0, s, isum, mx_bits, total, b.sum);
    SCAMD_LAUNCH_CHECK();
  }
  launch_sigma(knn_idx, knn_dist, n, k, b.sum, out_sigma, out_rho, b.w, b.outcnt, (int64_t)0, n, s);
  SCAMD_LAUNCH_CHECK();
  return symmetrise(b, knn_idx, n, k, 0, out_indptr, out_indices, out_data, nnz_host, s);
}

// ---- row-sharded fuzzy set ("symmetrisation needs one more all-to-all of (j, i, w) triples") ----------
// Rank r owns rows [row_begin, row_begin + n_local).  1. scamd_fuzzy_weights_f32: sigma / rho / membership strengths of
// its own rows (the only global quantity is the sum of all distances, one all-reduced double).  2. The caller sends every
// directed edge (i -> j, w_ij > 0) to the owner of row j (all-to-all) and sorts what it receives by (row, source).
// 3. scamd_fuzzy_merge_rows_f32: C[i][j] = w_ij + w_ji - w_ij w_ji over the union of row i's out- and in-edges, float32
// with one rounding per operation -- the same expression, hence the same bits, as the single-device kernel.
extern "C" int scamd_fuzzy_weights_f32(const int32_t* knn_idx, const float* knn_dist, int64_t n_local, int k,
                                       int64_t row_begin, int64_t n_total, const double* sum_all_dev, float* w,
                                       float* out_sigma, float* out_rho, int32_t* out_count, scamd_stream_t stream) {
  SCAMD_REQUIRE(knn_idx && knn_dist && sum_all_dev && w && out_count, SCAMD_EINVAL, "fuzzy_weights: null pointer");
  SCAMD_REQUIRE(n_local >= 0 && k >= 2 && k <= 1024 && row_begin >= 0 && row_begin + n_local <= n_total &&
                    n_total < ((int64_t)1 << 31),
                SCAMD_EINVAL, "fuzzy_weights: bad shape");
  if (n_local == 0) return SCAMD_OK;
  scamd::launch_sigma(knn_idx, knn_dist, n_local, k, sum_all_dev, out_sigma, out_rho, w, out_count, row_begin, n_total,
                      stream);
  SCAMD_LAUNCH_CHECK();
  return SCAMD_OK;
}

namespace scamd {
// entries of the merged row: the out-edges with w > 0, then the in-edges whose source is not among them
__global__ void fss_merge_count_kernel(const int* __restrict__ idx, const float* __restrict__ w, int64_t n, int k,
                                       const int64_t* __restrict__ in_indptr, const int* __restrict__ in_src,
                                       int* __restrict__ rowcnt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int cnt = 0;
  for (int j = 0; j < k; ++j) cnt += (w[i * k + j] > 0.f) ? 1 : 0;
  for (int64_t e = in_indptr[i]; e < in_indptr[i + 1]; ++e) {
    const int src = in_src[e];
    bool found = false;
    for (int j = 0; j < k; ++j) found |= (w[i * k + j] > 0.f && idx[i * k + j] == src);
    cnt += found ? 0 : 1;
  }
  rowcnt[i] = cnt;
}
__global__ void fss_merge_fill_kernel(const int* __restrict__ idx, const float* __restrict__ w, int64_t n, int k,
                                      const int64_t* __restrict__ in_indptr, const int* __restrict__ in_src,
                                      const float* __restrict__ in_w, const int64_t* __restrict__ out_indptr,
                                      int2* __restrict__ tmp) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t p = out_indptr[i];
  const int64_t e0 = in_indptr[i], e1 = in_indptr[i + 1];
  for (int j = 0; j < k; ++j) {
    const float we = w[i * k + j];
    if (!(we > 0.f)) continue;
    const int t = idx[i * k + j];
    float r = 0.f;  // reverse weight: binary search of t among the (ascending) sources of the in-edges
    int64_t lo = e0, hi = e1;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (in_src[mid] < t) lo = mid + 1;
      else hi = mid;
    }
    if (lo < e1 && in_src[lo] == t) r = in_w[lo];
    tmp[p] = make_int2(t, __float_as_int(__fsub_rn(__fadd_rn(we, r), __fmul_rn(we, r))));
    ++p;
  }
  for (int64_t e = e0; e < e1; ++e) {
    const int src = in_src[e];
    bool found = false;
    for (int j = 0; j < k; ++j) found |= (w[i * k + j] > 0.f && idx[i * k + j] == src);
    if (!found) {
      tmp[p] = make_int2(src, __float_as_int(in_w[e]));
      ++p;
    }
  }
}
}  // namespace scamd

extern "C" size_t scamd_fuzzy_merge_workspace_bytes(int64_t n_local, int64_t cap) {
  if (n_local < 0 || cap < 0) return 0;
  Workspace ws(nullptr, 0);
  (void)ws.take<int>((size_t)n_local + 1);
  (void)ws.take<int64_t>((size_t)scan_num_blocks(std::max<int64_t>(n_local, 1)) + 2);
  (void)ws.take<int2>((size_t)cap);
  return ws.used();
}

extern "C" int scamd_fuzzy_merge_rows_f32(const int32_t* knn_idx, const float* w, int64_t n_local, int k,
                                          const int64_t* in_indptr, const int32_t* in_src, const float* in_w,
                                          int64_t* out_indptr, int32_t* out_indices, float* out_data, int64_t cap,
                                          int64_t* nnz_host, void* workspace, size_t workspace_bytes,
                                          scamd_stream_t stream) {
  SCAMD_REQUIRE(n_local >= 0 && k >= 2 && k <= 1024 && cap >= 0, SCAMD_EINVAL, "fuzzy_merge: bad shape");
  if (n_local == 0) {  // a rank that owns no rows (n_total < world size): an empty CSR (empty tensors may be null)
    SCAMD_REQUIRE(out_indptr && nnz_host, SCAMD_EINVAL, "fuzzy_merge: null pointer");
    SCAMD_HIP_CHECK(hipMemsetAsync(out_indptr, 0, sizeof(int64_t), stream));
    *nnz_host = 0;
    return SCAMD_OK;
  }
  SCAMD_REQUIRE(knn_idx && w && in_indptr && out_indptr && out_indices && out_data && nnz_host, SCAMD_EINVAL,
                "fuzzy_merge: null pointer");
  Workspace ws(workspace, workspace_bytes);
  int* rowcnt = ws.take<int>((size_t)n_local + 1);
  int64_t* scan_tmp = ws.take<int64_t>((size_t)scan_num_blocks(n_local) + 2);
  int2* tmp = ws.take<int2>((size_t)cap);
  SCAMD_REQUIRE(workspace && ws.ok, SCAMD_EWORKSPACE, "fuzzy_merge: workspace %zu < required %zu", workspace_bytes, ws.used());
  hipStream_t s = stream;
  hipLaunchKernelGGL(fss_merge_count_kernel, dim3(ceil_div(n_local, 128)), dim3(128), 0, s, knn_idx, w, n_local, k,
                     in_indptr, in_src, rowcnt);
  SCAMD_LAUNCH_CHECK();
  int rc = exclusive_scan_i32_i64(rowcnt, n_local, out_indptr, scan_tmp, s);
  if (rc != SCAMD_OK) return rc;
  int64_t nnz = 0;
  SCAMD_READBACK_NOW(&nnz, out_indptr + n_local, sizeof(int64_t), s);
  SCAMD_REQUIRE(nnz <= cap, SCAMD_ECAPACITY, "fuzzy_merge: %lld entries exceed the capacity %lld", (long long)nnz,
                (long long)cap);
  hipLaunchKernelGGL(fss_merge_fill_kernel, dim3(ceil_div(n_local, 128)), dim3(128), 0, s, knn_idx, w, n_local, k,
                     in_indptr, in_src, in_w, out_indptr, tmp);
  SCAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(fss_sortrows_kernel, dim3(ceil_div(n_local, SR_ROWS)), dim3(256), 0, s, out_indptr, n_local, tmp,
                     out_indices, out_data);
  SCAMD_LAUNCH_CHECK();
  *nnz_host = nnz;
  SCAMD_HIP_CHECK(hipStreamSynchronize(s));
  return SCAMD_OK;
}

static int check_graph_args(const char* what, const int32_t* knn_idx, const void* knn_dist_or_idx, int64_t n, int k,
                            const int64_t* out_indptr, const int32_t* out_indices, const float* out_data, int64_t cap,
                            const int64_t* nnz_host) {
  SCAMD_REQUIRE(knn_idx && knn_dist_or_idx && out_indptr && out_indices && out_data && nnz_host, SCAMD_EINVAL,
                "%s: null pointer", what);
  SCAMD_REQUIRE(n >= 1 && k >= 2 && k <= 256, SCAMD_EINVAL, "%s: bad shape n=%lld k=%d (k <= 256)", what, (long long)n, k);
  SCAMD_REQUIRE(n < (int64_t)1 << 31, SCAMD_EUNSUPPORTED, "%s: n exceeds int32 ids", what);
  SCAMD_REQUIRE(cap >= 2 * n * (k - 1), SCAMD_ECAPACITY, "%s: cap %lld < 2*n*(k-1) = %lld", what, (long long)cap,
                (long long)(2 * n * (k - 1)));
  return SCAMD_OK;
}

extern "C" int scamd_gauss_connectivities_f32(const int32_t* knn_idx, const float* knn_dist, int64_t n, int k,
                                              int64_t* out_indptr, int32_t* out_indices, float* out_data, int64_t cap,
                                              int64_t* nnz_host, void* workspace, size_t workspace_bytes,
                                              scamd_stream_t stream) {
  int rc = check_graph_args("gauss", knn_idx, knn_dist, n, k, out_indptr, out_indices, out_data, cap, nnz_host);
  if (rc != SCAMD_OK) return rc;
  Workspace ws(workspace, workspace_bytes);
  FuzzyBuffers b;
  fuzzy_carve(ws, n, k, &b);
  double* sigma_sq = ws.take<double>((size_t)n);
  SCAMD_REQUIRE(workspace && ws.ok, SCAMD_EWORKSPACE, "gauss: workspace %zu < required %zu", workspace_bytes, ws.used());
  hipStream_t s = stream;
  SCAMD_HIP_CHECK(hipMemsetAsync(b.in_only, 0, sizeof(int) * n, s));
  hipLaunchKernelGGL(gauss_sigma_kernel, dim3(ceil_div(n, 128)), dim3(128), 0, s, knn_idx, knn_dist, n, k, sigma_sq);
  SCAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(gauss_weight_kernel, dim3(ceil_div(n, 128)), dim3(128), 0, s, knn_idx, knn_dist, n, k, sigma_sq, b.w,
                     b.outcnt);
  SCAMD_LAUNCH_CHECK();
  return symmetrise(b, knn_idx, n, k, 1, out_indptr, out_indices, out_data, nnz_host, s);
}

extern "C" int scamd_jaccard_connectivities_f32(const int32_t* knn_idx, int64_t n, int k, int64_t* out_indptr,
                                                int32_t* out_indices, float* out_data, int64_t cap, int64_t* nnz_host,
                                                void* workspace, size_t workspace_bytes, scamd_stream_t stream) {
  int rc = check_graph_args("jaccard", knn_idx, knn_idx, n, k, out_indptr, out_indices, out_data, cap, nnz_host);
  if (rc != SCAMD_OK) return rc;
  Workspace ws(workspace, workspace_bytes);
  FuzzyBuffers b;
  fuzzy_carve(ws, n, k, &b);
  (void)ws.take<double>((size_t)n);  // same layout as the gauss entry point: one workspace size for all three
  SCAMD_REQUIRE(workspace && ws.ok, SCAMD_EWORKSPACE, "jaccard: workspace %zu < required %zu", workspace_bytes, ws.used());
  hipStream_t s = stream;
  const int64_t total = n * k;
  SCAMD_HIP_CHECK(hipMemsetAsync(b.in_only, 0, sizeof(int) * n, s));
  hipLaunchKernelGGL(jaccard_weight_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, s, knn_idx, n, k, b.w);
  SCAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(count_positive_rows_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, b.w, n, k, b.outcnt);
  SCAMD_LAUNCH_CHECK();
  return symmetrise(b, knn_idx, n, k, 2, out_indptr, out_indices, out_data, nnz_host, s);
}
