// Upstream normalisation chain on a CSR float32 matrix: the device passes behind
// scanpy_amd.pp.normalize_total / log1p / highly_variable_genes / scale.
//
// Every pass is HBM-bound (one sweep over indptr/indices/data, 8 B per stored entry, plus what it writes):
//   row sums            reads 4 B/entry (+ 4 B of column index when highly expressed genes are excluded)
//   row divide, log1p   read + write 4 B/entry
//   column statistics   reads 8 B/entry; per-gene sums accumulate in LDS (float64 atomics, one table per workgroup,
//                       flushed once), so the global atomics are g per workgroup, not one per entry
//   scale               CSR in place: read 8 + write 4 B/entry; dense (zero_center): writes n*g*sizeof(out)
// Reference semantics are cited per entry point in include/scanpy_amd.h.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "common.h"

namespace scamd {

namespace {

constexpr int PP_BLOCK = 256;
constexpr int PP_LDS_GENES = 4096;  // column tables up to this many genes live in LDS (80 KB per workgroup)

// Row-wise kernels give G lanes to a row (64 / G rows per wave): a row of a cells x genes matrix holds ~100 entries,
// and a wave that walks ONE row is bound by the latency of its indptr -> data chain, not by HBM; 4-8 rows per wave
// put 4-8x as many loads in flight.  G is chosen on the host from the mean row length.
#define PP_ROW_LOOP(G)                                                                                   \
  const int sub = threadIdx.x % G;                                                                       \
  const int64_t ngroups = (int64_t)gridDim.x * (PP_BLOCK / G);                                           \
  for (int64_t r = (int64_t)blockIdx.x * (PP_BLOCK / G) + threadIdx.x / G; r < n; r += ngroups)

template <int G>
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// out[r] = float( sum of row r in float64 ), entries of columns with col_skip[c] != 0 left out
template <int G>
__global__ __launch_bounds__(PP_BLOCK) void pp_row_sums_kernel(const int64_t* __restrict__ indptr,
                                                               const int32_t* __restrict__ indices,
                                                               const float* __restrict__ data, int64_t n,
                                                               const int32_t* __restrict__ col_skip,
                                                               float* __restrict__ out) {
  PP_ROW_LOOP(G) {
    const int64_t b = indptr[r], e = indptr[r + 1];
    double s = 0.0;
    if (col_skip) {
      for (int64_t p = b + sub; p < e; p += G)
        if (col_skip[indices[p]] == 0) s += (double)data[p];
    } else {
#pragma unroll 4
      for (int64_t p = b + sub; p < e; p += G) s += (double)data[p];
    }
    s = group_sum<G>(s);
    if (sub == 0) out[r] = (float)s;
  }
}

// out[r] = number of stored entries of row r that are > 0  (`data > 0` summed along axis 1, filter_cells(min_genes=))
template <int G>
__global__ __launch_bounds__(PP_BLOCK) void pp_row_npos_kernel(const int64_t* __restrict__ indptr,
                                                               const float* __restrict__ data, int64_t n,
                                                               int32_t* __restrict__ out) {
  PP_ROW_LOOP(G) {
    const int64_t e = indptr[r + 1];
    int c = 0;
#pragma unroll 4
    for (int64_t p = indptr[r] + sub; p < e; p += G) c += data[p] > 0.f ? 1 : 0;
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if (sub == 0) out[r] = c;
  }
}

// col_counts[c] += number of entries of column c with value > max_fraction * row_total[row]
template <int G>
__global__ __launch_bounds__(PP_BLOCK) void pp_count_high_kernel(const int64_t* __restrict__ indptr,
                                                                 const int32_t* __restrict__ indices,
                                                                 const float* __restrict__ data, int64_t n,
                                                                 const float* __restrict__ row_total,
                                                                 float max_fraction, int32_t* __restrict__ col_counts) {
  PP_ROW_LOOP(G) {
    const float thr = max_fraction * row_total[r];
    const int64_t e = indptr[r + 1];
#pragma unroll 4
    for (int64_t p = indptr[r] + sub; p < e; p += G)
      if (data[p] > thr) atomicAdd(&col_counts[indices[p]], 1);
  }
}

// data[p] /= factor[row]   (factor == 0 -> 1: a cell without counts stays all-zero)
template <int G>
__global__ __launch_bounds__(PP_BLOCK) void pp_row_divide_kernel(const int64_t* __restrict__ indptr,
                                                                 float* __restrict__ data, int64_t n,
                                                                 const float* __restrict__ factor) {
  PP_ROW_LOOP(G) {
    float f = factor[r];
    f = (f == 0.f) ? 1.f : f;
    const int64_t e = indptr[r + 1];
#pragma unroll 4
    for (int64_t p = indptr[r] + sub; p < e; p += G) data[p] = data[p] / f;
  }
}

// 16-byte accesses on the aligned body, scalars on the (at most 3 + 3) unaligned head / tail elements
__global__ __launch_bounds__(PP_BLOCK) void pp_log1p_kernel(float* __restrict__ data, int64_t count, float inv_log_base,
                                                            int has_base) {
  const int64_t tid = (int64_t)blockIdx.x * PP_BLOCK + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * PP_BLOCK;
  const int64_t head = std::min<int64_t>(count, (int64_t)((16 - (reinterpret_cast<uintptr_t>(data) & 15)) & 15) / 4);
  const int64_t nvec = (count - head) / 4;
  float4* v4 = reinterpret_cast<float4*>(data + head);
  for (int64_t i = tid; i < nvec; i += stride) {
    float4 v = v4[i];
    v.x = log1pf(v.x);
    v.y = log1pf(v.y);
    v.z = log1pf(v.z);
    v.w = log1pf(v.w);
    if (has_base) {
      v.x *= inv_log_base;
      v.y *= inv_log_base;
      v.z *= inv_log_base;
      v.w *= inv_log_base;
    }
    v4[i] = v;
  }
  // head and tail
  const int64_t tail0 = head + nvec * 4;
  const int64_t n_edge = head + (count - tail0);
  if (tid < n_edge) {
    const int64_t i = tid < head ? tid : tail0 + (tid - head);
    float v = log1pf(data[i]);
    if (has_base) v *= inv_log_base;
    data[i] = v;
  }
}

__device__ __forceinline__ float pp_transform(float v, int transform, float tscale) {
  return transform == 1 ? expm1f(v * tscale) : v;
}

// Per-gene sum, sum of squares (float64) and (npos != NULL) number of positive entries over the rows with
// row_mask != 0.  LDS = true: one table per workgroup in LDS, flushed with g global atomics at the end.
template <bool LDS, int G>
__global__ __launch_bounds__(PP_BLOCK) void pp_col_stats_kernel(const int64_t* __restrict__ indptr,
                                                                const int32_t* __restrict__ indices,
                                                                const float* __restrict__ data, int64_t n, int g,
                                                                const uint8_t* __restrict__ row_mask, int transform,
                                                                float tscale, const double* __restrict__ clip,
                                                                double* __restrict__ sum,
                                                                double* __restrict__ sumsq,
                                                                unsigned long long* __restrict__ npos) {
  extern __shared__ __attribute__((aligned(16))) double pp_smem[];
  double* s_sum = pp_smem;
  double* s_sq = pp_smem + g;
  unsigned int* s_cnt = reinterpret_cast<unsigned int*>(pp_smem + 2 * (size_t)g);
  if (LDS) {
    for (int c = threadIdx.x; c < g; c += PP_BLOCK) {
      s_sum[c] = 0.0;
      s_sq[c] = 0.0;
      s_cnt[c] = 0u;
    }
    __syncthreads();
  }
  PP_ROW_LOOP(G) {
    if (row_mask && !row_mask[r]) continue;
    const int64_t e = indptr[r + 1];
#pragma unroll 2
    for (int64_t p = indptr[r] + sub; p < e; p += G) {
      const int c = indices[p];
      const float v = pp_transform(data[p], transform, tscale);
      double dv = (double)v;
      if (clip) dv = fmin(dv, clip[c]);  // `clip_square_sum` of flavor='seurat_v3' (_highly_variable_genes.py:75-115)
      if (LDS) {
        atomicAdd(&s_sum[c], dv);
        atomicAdd(&s_sq[c], dv * dv);
        if (npos && v > 0.f) atomicAdd(&s_cnt[c], 1u);
      } else {
        atomicAdd(&sum[c], dv);
        atomicAdd(&sumsq[c], dv * dv);
        if (npos && v > 0.f) atomicAdd(&npos[c], 1ull);
      }
    }
  }
  if (LDS) {
    __syncthreads();
    for (int c = threadIdx.x; c < g; c += PP_BLOCK) {
      if (s_sum[c] != 0.0) atomicAdd(&sum[c], s_sum[c]);
      if (s_sq[c] != 0.0) atomicAdd(&sumsq[c], s_sq[c]);
      if (npos && s_cnt[c]) atomicAdd(&npos[c], (unsigned long long)s_cnt[c]);
    }
  }
}

// zero_center = False: data[p] = min(max_value, data[p] / std[col]) on the rows with row_mask != 0
template <int G>
__global__ __launch_bounds__(PP_BLOCK) void pp_scale_csr_kernel(const int64_t* __restrict__ indptr,
                                                                const int32_t* __restrict__ indices,
                                                                float* __restrict__ data, int64_t n,
                                                                const double* __restrict__ std_, double max_value,
                                                                int has_max, const uint8_t* __restrict__ row_mask) {
  PP_ROW_LOOP(G) {
    if (row_mask && !row_mask[r]) continue;
    const int64_t e = indptr[r + 1];
#pragma unroll 4
    for (int64_t p = indptr[r] + sub; p < e; p += G) {
      double v = (double)data[p] / std_[indices[p]];
      if (has_max && v > max_value) v = max_value;
      data[p] = (float)v;
    }
  }
}

// zero_center = True, pass 1: every cell of the dense output gets the value of an implicit zero,
// clip((0 - mean[c]) / std[c]); rows outside the mask get 0 (they keep their original values, pass 2 writes them)
template <typename OutT>
__global__ __launch_bounds__(PP_BLOCK) void pp_scale_dense_fill_kernel(int64_t n, int g,
                                                                       const double* __restrict__ mean,
                                                                       const double* __restrict__ std_,
                                                                       double max_value, int has_max,
                                                                       const uint8_t* __restrict__ row_mask,
                                                                       OutT* __restrict__ out) {
  const int64_t total = n * (int64_t)g;
  const int64_t stride = (int64_t)gridDim.x * PP_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * PP_BLOCK + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / g;
    const int c = (int)(i - r * g);
    double v = 0.0;
    if (!row_mask || row_mask[r]) {
      v = (double)((OutT)(0.0 - mean[c])) / std_[c];  // subtraction rounded to the output dtype, as numpy's in-place ops do
      if (has_max) v = v > max_value ? max_value : (v < -max_value ? -max_value : v);
    }
    out[i] = (OutT)v;
  }
}

// pass 2: the stored entries
template <typename OutT, int G>
__global__ __launch_bounds__(PP_BLOCK) void pp_scale_dense_scatter_kernel(const int64_t* __restrict__ indptr,
                                                                          const int32_t* __restrict__ indices,
                                                                          const float* __restrict__ data, int64_t n,
                                                                          int g, const double* __restrict__ mean,
                                                                          const double* __restrict__ std_,
                                                                          double max_value, int has_max,
                                                                          const uint8_t* __restrict__ row_mask,
                                                                          OutT* __restrict__ out) {
  PP_ROW_LOOP(G) {
    const bool on = !row_mask || row_mask[r];
    OutT* orow = out + r * (int64_t)g;
    const int64_t e = indptr[r + 1];
    for (int64_t p = indptr[r] + sub; p < e; p += G) {
      const int c = indices[p];
      double v = (double)data[p];
      if (on) {
        // numpy computes `x -= mean` in float64 and rounds to the array dtype (float32 dense input stays float32;
        // a sparse input becomes a float64 matrix), then divides the same way
        v = (double)((OutT)(v - mean[c])) / std_[c];
        if (has_max) v = v > max_value ? max_value : (v < -max_value ? -max_value : v);
      }
      orow[c] = (OutT)v;
    }
  }
}

// lanes per row from the mean row length
inline int lanes_per_row(const int64_t* /*indptr (device)*/, int64_t n, int64_t nnz_hint) {
  const int64_t avg = n > 0 ? nnz_hint / n : 0;
  return avg <= 32 ? 8 : (avg <= 160 ? 16 : (avg <= 512 ? 32 : 64));
}

inline unsigned group_grid(int64_t n, int G) {
  return (unsigned)std::min<int64_t>(std::max<int64_t>(ceil_div(n, PP_BLOCK / G), 1), 256 * 32);
}

#define PP_DISPATCH_G(G_, KERNEL, ...)                                                                        \
  do {                                                                                                         \
    switch (G_) {                                                                                              \
      case 8: hipLaunchKernelGGL((KERNEL<8>), dim3(group_grid(n, 8)), dim3(PP_BLOCK), 0, stream, __VA_ARGS__); break;   \
      case 16: hipLaunchKernelGGL((KERNEL<16>), dim3(group_grid(n, 16)), dim3(PP_BLOCK), 0, stream, __VA_ARGS__); break; \
      case 32: hipLaunchKernelGGL((KERNEL<32>), dim3(group_grid(n, 32)), dim3(PP_BLOCK), 0, stream, __VA_ARGS__); break; \
      default: hipLaunchKernelGGL((KERNEL<64>), dim3(group_grid(n, 64)), dim3(PP_BLOCK), 0, stream, __VA_ARGS__); break; \
    }                                                                                                          \
  } while (0)

}  // namespace
}  // namespace scamd

using namespace scamd;

extern "C" int scamd_pp_row_sums_f32(const int64_t* indptr, const int32_t* indices, const float* data, int64_t n,
                                     int64_t nnz, const int32_t* col_skip, float* out, scamd_stream_t stream) {
  SCAMD_REQUIRE(indptr && n >= 0 && nnz >= 0 && (n == 0 || out), SCAMD_EINVAL, "pp_row_sums: null pointer or negative size");
  SCAMD_REQUIRE(!col_skip || indices, SCAMD_EINVAL, "pp_row_sums: col_skip needs indices");
  if (n == 0) return SCAMD_OK;
  PP_DISPATCH_G(lanes_per_row(indptr, n, nnz), pp_row_sums_kernel, indptr, indices, data, n, col_skip, out);
  SCAMD_LAUNCH_CHECK();
  return SCAMD_OK;
}

extern "C" int scamd_pp_row_count_positive_f32(const int64_t* indptr, const float* data, int64_t n, int64_t nnz,
                                               int32_t* out, scamd_stream_t stream) {
  SCAMD_REQUIRE(indptr && n >= 0 && nnz >= 0 && (n == 0 || out), SCAMD_EINVAL, "pp_row_count_positive: bad argument");
  if (n == 0) return SCAMD_OK;
  PP_DISPATCH_G(lanes_per_row(indptr, n, nnz), pp_row_npos_kernel, indptr, data, n, out);
  SCAMD_LAUNCH_CHECK();
  return SCAMD_OK;
}

extern "C" int scamd_pp_count_high_f32(const int64_t* indptr, const int32_t* indices, const float* data, int64_t n,
                                       int64_t g, int64_t nnz, const float* row_total, float max_fraction,
                                       int32_t* col_counts, scamd_stream_t stream) {
  SCAMD_REQUIRE(indptr && n >= 0 && g >= 0 && nnz >= 0 && (n == 0 || row_total) && (g == 0 || col_counts), SCAMD_EINVAL,
                "pp_count_high: bad argument");
  SCAMD_HIP_CHECK(hipMemsetAsync(col_counts, 0, sizeof(int32_t) * (size_t)g, stream));
  if (n == 0) return SCAMD_OK;
  PP_DISPATCH_G(lanes_per_row(indptr, n, nnz), pp_count_high_kernel, indptr, indices, data, n, row_total, max_fraction,
                col_counts);
  SCAMD_LAUNCH_CHECK();
  return SCAMD_OK;
}

extern "C" int scamd_pp_row_divide_f32(const int64_t* indptr, float* data, int64_t n, int64_t nnz, const float* factor,
                                       scamd_stream_t stream) {
  SCAMD_REQUIRE(indptr && n >= 0 && nnz >= 0 && (n == 0 || factor), SCAMD_EINVAL, "pp_row_divide: bad argument");
  if (n == 0) return SCAMD_OK;
  PP_DISPATCH_G(lanes_per_row(indptr, n, nnz), pp_row_divide_kernel, indptr, data, n, factor);
  SCAMD_LAUNCH_CHECK();
  return SCAMD_OK;
}

extern "C" int scamd_pp_log1p_f32(float* data, int64_t count, double base, scamd_stream_t stream) {
  SCAMD_REQUIRE(count >= 0 && (count == 0 || data), SCAMD_EINVAL, "pp_log1p: bad argument");
  SCAMD_REQUIRE(base == 0.0 || (base > 0.0 && base != 1.0), SCAMD_EINVAL, "pp_log1p: base must be > 0 and != 1");
  if (count == 0) return SCAMD_OK;
  const unsigned blocks = (unsigned)std::min<int64_t>(std::max<int64_t>(ceil_div(count / 4 + 8, PP_BLOCK), 1), 256 * 32);
  const float inv = base > 0.0 ? (float)(1.0 / log(base)) : 1.0f;
  hipLaunchKernelGGL(pp_log1p_kernel, dim3(blocks), dim3(PP_BLOCK), 0, stream, data, count, inv, base > 0.0 ? 1 : 0);
  SCAMD_LAUNCH_CHECK();
  return SCAMD_OK;
}

template <int G>
static int launch_col_stats(const int64_t* indptr, const int32_t* indices, const float* data, int64_t n, int g,
                            const uint8_t* row_mask, int transform, float tscale, const double* clip, double* sum,
                            double* sumsq, unsigned long long* npos, hipStream_t stream) {
  if (g <= PP_LDS_GENES) {
    const size_t lds = (size_t)g * (8 + 8 + 4) + 16;
    const unsigned blocks = (unsigned)std::min<int64_t>(std::max<int64_t>(ceil_div(n, (PP_BLOCK / G) * 16), 1), 512);
    hipLaunchKernelGGL((pp_col_stats_kernel<true, G>), dim3(blocks), dim3(PP_BLOCK), lds, stream, indptr, indices, data, n, g,
                       row_mask, transform, tscale, clip, sum, sumsq, npos);
  } else {
    hipLaunchKernelGGL((pp_col_stats_kernel<false, G>), dim3(group_grid(n, G)), dim3(PP_BLOCK), 0, stream, indptr, indices,
                       data, n, g, row_mask, transform, tscale, clip, sum, sumsq, npos);
  }
  SCAMD_LAUNCH_CHECK();
  return SCAMD_OK;
}

extern "C" int scamd_pp_col_stats_f32(const int64_t* indptr, const int32_t* indices, const float* data, int64_t n,
                                      int64_t g, int64_t nnz, const uint8_t* row_mask, int transform, double tscale,
                                      double* sum, double* sumsq, uint64_t* npos, scamd_stream_t stream) {
  SCAMD_REQUIRE(indptr && n >= 0 && g >= 0 && nnz >= 0 && g < ((int64_t)1 << 31) && (g == 0 || (sum && sumsq)),
                SCAMD_EINVAL, "pp_col_stats: bad argument");
  SCAMD_REQUIRE(transform == 0 || transform == 1, SCAMD_EINVAL, "pp_col_stats: unknown transform %d", transform);
  SCAMD_HIP_CHECK(hipMemsetAsync(sum, 0, sizeof(double) * (size_t)g, stream));
  SCAMD_HIP_CHECK(hipMemsetAsync(sumsq, 0, sizeof(double) * (size_t)g, stream));
  if (npos) SCAMD_HIP_CHECK(hipMemsetAsync(npos, 0, sizeof(uint64_t) * (size_t)g, stream));
  if (n == 0 || g == 0) return SCAMD_OK;
  unsigned long long* np = reinterpret_cast<unsigned long long*>(npos);
  switch (lanes_per_row(indptr, n, nnz)) {
    case 8: return launch_col_stats<8>(indptr, indices, data, n, (int)g, row_mask, transform, (float)tscale, nullptr, sum, sumsq, np, stream);
    case 16: return launch_col_stats<16>(indptr, indices, data, n, (int)g, row_mask, transform, (float)tscale, nullptr, sum, sumsq, np, stream);
    case 32: return launch_col_stats<32>(indptr, indices, data, n, (int)g, row_mask, transform, (float)tscale, nullptr, sum, sumsq, np, stream);
    default: return launch_col_stats<64>(indptr, indices, data, n, (int)g, row_mask, transform, (float)tscale, nullptr, sum, sumsq, np, stream);
  }
}

extern "C" int scamd_pp_col_stats_clip_f32(const int64_t* indptr, const int32_t* indices, const float* data, int64_t n,
                                           int64_t g, int64_t nnz, const uint8_t* row_mask, const double* clip,
                                           double* sum, double* sumsq, scamd_stream_t stream) {
  SCAMD_REQUIRE(indptr && clip && n >= 0 && g >= 0 && nnz >= 0 && g < ((int64_t)1 << 31) && (g == 0 || (sum && sumsq)),
                SCAMD_EINVAL, "pp_col_stats_clip: bad argument");
  SCAMD_HIP_CHECK(hipMemsetAsync(sum, 0, sizeof(double) * (size_t)g, stream));
  SCAMD_HIP_CHECK(hipMemsetAsync(sumsq, 0, sizeof(double) * (size_t)g, stream));
  if (n == 0 || g == 0) return SCAMD_OK;
  switch (lanes_per_row(indptr, n, nnz)) {
    case 8: return launch_col_stats<8>(indptr, indices, data, n, (int)g, row_mask, 0, 1.0f, clip, sum, sumsq, nullptr, stream);
    case 16: return launch_col_stats<16>(indptr, indices, data, n, (int)g, row_mask, 0, 1.0f, clip, sum, sumsq, nullptr, stream);
    case 32: return launch_col_stats<32>(indptr, indices, data, n, (int)g, row_mask, 0, 1.0f, clip, sum, sumsq, nullptr, stream);
    default: return launch_col_stats<64>(indptr, indices, data, n, (int)g, row_mask, 0, 1.0f, clip, sum, sumsq, nullptr, stream);
  }
}

extern "C" int scamd_pp_scale_csr_f32(const int64_t* indptr, const int32_t* indices, float* data, int64_t n,
                                      int64_t nnz, const double* std_, double max_value, int has_max,
                                      const uint8_t* row_mask, scamd_stream_t stream) {
  SCAMD_REQUIRE(indptr && n >= 0 && nnz >= 0 && (n == 0 || std_), SCAMD_EINVAL, "pp_scale_csr: bad argument");
  if (n == 0) return SCAMD_OK;
  PP_DISPATCH_G(lanes_per_row(indptr, n, nnz), pp_scale_csr_kernel, indptr, indices, data, n, std_, max_value, has_max,
                row_mask);
  SCAMD_LAUNCH_CHECK();
  return SCAMD_OK;
}

template <typename OutT>
static int launch_scale_dense(const int64_t* indptr, const int32_t* indices, const float* data, int64_t n, int g,
                              int64_t nnz, const double* mean, const double* std_, double max_value, int has_max,
                              const uint8_t* row_mask, OutT* out, hipStream_t stream) {
  const unsigned fblocks = (unsigned)std::min<int64_t>(ceil_div(n * (int64_t)g, PP_BLOCK), 256 * 64);
  hipLaunchKernelGGL(pp_scale_dense_fill_kernel<OutT>, dim3(fblocks), dim3(PP_BLOCK), 0, stream, n, g, mean, std_, max_value,
                     has_max, row_mask, out);
  SCAMD_LAUNCH_CHECK();
  switch (lanes_per_row(indptr, n, nnz)) {
    case 8: hipLaunchKernelGGL((pp_scale_dense_scatter_kernel<OutT, 8>), dim3(group_grid(n, 8)), dim3(PP_BLOCK), 0, stream, indptr, indices, data, n, g, mean, std_, max_value, has_max, row_mask, out); break;
    case 16: hipLaunchKernelGGL((pp_scale_dense_scatter_kernel<OutT, 16>), dim3(group_grid(n, 16)), dim3(PP_BLOCK), 0, stream, indptr, indices, data, n, g, mean, std_, max_value, has_max, row_mask, out); break;
    case 32: hipLaunchKernelGGL((pp_scale_dense_scatter_kernel<OutT, 32>), dim3(group_grid(n, 32)), dim3(PP_BLOCK), 0, stream, indptr, indices, data, n, g, mean, std_, max_value, has_max, row_mask, out); break;
    default: hipLaunchKernelGGL((pp_scale_dense_scatter_kernel<OutT, 64>), dim3(group_grid(n, 64)), dim3(PP_BLOCK), 0, stream, indptr, indices, data, n, g, mean, std_, max_value, has_max, row_mask, out); break;
  }
  SCAMD_LAUNCH_CHECK();
  return SCAMD_OK;
}

extern "C" int scamd_pp_scale_dense_f32(const int64_t* indptr, const int32_t* indices, const float* data, int64_t n,
                                        int64_t g, int64_t nnz, const double* mean, const double* std_,
                                        double max_value, int has_max, const uint8_t* row_mask, void* out,
                                        int out_is_f64, scamd_stream_t stream) {
  SCAMD_REQUIRE(indptr && n >= 0 && g >= 0 && nnz >= 0 && g < ((int64_t)1 << 31) &&
                    (n == 0 || g == 0 || (mean && std_ && out)),
                SCAMD_EINVAL, "pp_scale_dense: bad argument");
  if (n == 0 || g == 0) return SCAMD_OK;
  if (out_is_f64)
    return launch_scale_dense<double>(indptr, indices, data, n, (int)g, nnz, mean, std_, max_value, has_max, row_mask,
                                      static_cast<double*>(out), stream);
  return launch_scale_dense<float>(indptr, indices, data, n, (int)g, nnz, mean, std_, max_value, has_max, row_mask,
                                   static_cast<float*>(out), stream);
}
