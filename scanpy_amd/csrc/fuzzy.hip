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

__global__ void fss_sum_kernel(const float* __restrict__ d, int64_t total, double* __restrict__ sum) {
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x)
    s += (double)d[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(sum, s);
}

// one thread per row
__global__ void fss_sigma_kernel(const int* __restrict__ idx, const float* __restrict__ dist, int64_t n,
                                 int k, const double* __restrict__ sum_all, float* __restrict__ sigma_out,
                                 float* __restrict__ rho_out, float* __restrict__ w,
                                 int* __restrict__ outcnt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* di = dist + i * k;
  const int* ii = idx + i * k;
  float rho = 0.f;
  double rowsum = 0.0;
  bool found = false;
  for (int j = 0; j < k; ++j) {
    float v = di[j];
    rowsum += (double)v;
    if (!found && v > 0.f) {
      rho = v;
      found = true;
    }
  }
  const double target = log2((double)k);
  double lo = 0.0, hi = INFINITY, mid = 1.0;
  for (int it = 0; it < 64; ++it) {
    double psum = 0.0;
    for (int j = 1; j < k; ++j) {
      float df = __fsub_rn(di[j], rho);
      double dd = (double)df;
      psum += (dd > 0.0) ? exp(-(dd / mid)) : 1.0;
    }
    if (fabs(psum - target) < 1e-5) break;
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
    double mean_all = *sum_all / ((double)n * (double)k);
    if ((double)sigma < 1e-3 * mean_all) sigma = (float)(1e-3 * mean_all);
  }
  if (sigma_out) sigma_out[i] = sigma;
  if (rho_out) rho_out[i] = rho;
  int cnt = 0;
  for (int j = 0; j < k; ++j) {
    int t = ii[j];
    float val;
    float df = __fsub_rn(di[j], rho);
    if (t == (int)i || t < 0) val = 0.f;
    else if (df <= 0.f || sigma == 0.f) val = 1.f;
    else val = (float)exp(-(double)__fdiv_rn(df, sigma));  // correctly rounded, keeps float32 subnormals
    w[i * k + j] = val;
    cnt += (val > 0.f) ? 1 : 0;
  }
  outcnt[i] = cnt;
}

// one thread per directed slot (i, j): weight of the reverse edge, in-only degree of the target
__global__ void fss_recip_kernel(const int* __restrict__ idx, const float* __restrict__ w, int64_t n, int k,
                                 float* __restrict__ recw, int* __restrict__ in_only) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * k) return;
  const float we = w[e];
  float r = 0.f;
  if (we > 0.f) {
    const int64_t i = e / k;
    const int t = idx[e];
    const int* ti = idx + (int64_t)t * k;
    const float* tw = w + (int64_t)t * k;
    for (int j = 0; j < k; ++j)
      if (ti[j] == (int)i) r = fmaxf(r, tw[j]);
    if (r == 0.f) atomicAdd(&in_only[t], 1);
  }
  recw[e] = r;
}

__global__ void fss_rowcount_kernel(const int* __restrict__ outcnt, const int* __restrict__ in_only, int64_t n,
                                    int* __restrict__ rowcnt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) rowcnt[i] = outcnt[i] + in_only[i];
}

__global__ void fss_fill_kernel(const int* __restrict__ idx, const float* __restrict__ w,
                                const float* __restrict__ recw, const int* __restrict__ outcnt,
                                const int64_t* __restrict__ indptr, int64_t n, int k, int* __restrict__ cursor,
                                int* __restrict__ tmp_col, float* __restrict__ tmp_val) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * k) return;
  const float we = w[e];
  if (!(we > 0.f)) return;
  const int64_t i = e / k;
  const int j = (int)(e - i * k);
  const int t = idx[e];
  const float r = recw[e];
  int before = 0;
  for (int jj = 0; jj < j; ++jj) before += (w[i * k + jj] > 0.f) ? 1 : 0;
  const int64_t p = indptr[i] + before;
  // (w + w^T) - w*w^T with one rounding per operation, as scipy's float32 sparse arithmetic does
  const float val = __fsub_rn(__fadd_rn(we, r), __fmul_rn(we, r));
  tmp_col[p] = t;
  tmp_val[p] = val;
  if (r == 0.f) {
    const int slot = atomicAdd(&cursor[t], 1);
    const int64_t pt = indptr[t] + outcnt[t] + slot;
    tmp_col[pt] = (int)i;
    tmp_val[pt] = we;
  }
}

// one wave per row: rank sort by column index (columns are unique within a row)
__global__ void fss_sortrows_kernel(const int64_t* __restrict__ indptr, int64_t n, const int* __restrict__ tmp_col,
                                    const float* __restrict__ tmp_val, int* __restrict__ out_col,
                                    float* __restrict__ out_val) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= n) return;
  const int64_t base = indptr[row];
  const int len = (int)(indptr[row + 1] - base);
  for (int e = lane; e < len; e += 64) {
    const int c = tmp_col[base + e];
    const float v = tmp_val[base + e];
    int rank = 0;
    for (int u = 0; u < len; ++u) {
      int cu = tmp_col[base + u];
      rank += (cu < c || (cu == c && u < e)) ? 1 : 0;
    }
    out_col[base + rank] = c;
    out_val[base + rank] = v;
  }
}

struct FuzzyBuffers {
  float* w; float* recw; int* outcnt; int* in_only; int* cursor; int* rowcnt; int64_t* scan_tmp;
  int* tmp_col; float* tmp_val; double* sum;
};

static void fuzzy_carve(Workspace& ws, int64_t n, int k, FuzzyBuffers* b) {
  const size_t cap = (size_t)2 * n * (k > 1 ? k - 1 : 1);
  b->w = ws.take<float>((size_t)n * k);
  b->recw = ws.take<float>((size_t)n * k);
  b->outcnt = ws.take<int>((size_t)n);
  b->in_only = ws.take<int>((size_t)n);
  b->cursor = ws.take<int>((size_t)n);
  b->rowcnt = ws.take<int>((size_t)n);
  b->scan_tmp = ws.take<int64_t>((size_t)scan_num_blocks(n) + 2);
  b->tmp_col = ws.take<int>(cap);
  b->tmp_val = ws.take<float>(cap);
  b->sum = ws.take<double>(2);
}

}  // namespace scamd

using namespace scamd;

extern "C" size_t scamd_fuzzy_workspace_bytes(int64_t n, int k) {
  if (n <= 0 || k <= 0) return 0;
  Workspace ws(nullptr, 0);
  FuzzyBuffers b;
  fuzzy_carve(ws, n, k, &b);
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
  SCAMD_HIP_CHECK(hipMemsetAsync(b.sum, 0, 16, s));
  SCAMD_HIP_CHECK(hipMemsetAsync(b.in_only, 0, sizeof(int) * n, s));
  SCAMD_HIP_CHECK(hipMemsetAsync(b.cursor, 0, sizeof(int) * n, s));
  {
    int blocks = (int)std::min<int64_t>((total + 255) / 256, 2048);
    hipLaunchKernelGGL(fss_sum_kernel, dim3(blocks), dim3(256), 0, s, knn_dist, total, b.sum);
    SCAMD_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(fss_sigma_kernel, dim3(ceil_div(n, 128)), dim3(128), 0, s, knn_idx, knn_dist, n, k, b.sum,
                     out_sigma, out_rho, b.w, b.outcnt);
  SCAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(fss_recip_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, s, knn_idx, b.w, n, k, b.recw,
                     b.in_only);
  SCAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(fss_rowcount_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, b.outcnt, b.in_only, n,
                     b.rowcnt);
  SCAMD_LAUNCH_CHECK();
  int rc = exclusive_scan_i32_i64(b.rowcnt, n, out_indptr, b.scan_tmp, s);
  if (rc != SCAMD_OK) return rc;
  hipLaunchKernelGGL(fss_fill_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, s, knn_idx, b.w, b.recw,
                     b.outcnt, out_indptr, n, k, b.cursor, b.tmp_col, b.tmp_val);
  SCAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(fss_sortrows_kernel, dim3(ceil_div(n, 4)), dim3(256), 0, s, out_indptr, n, b.tmp_col,
                     b.tmp_val, out_indices, out_data);
  SCAMD_LAUNCH_CHECK();
  int64_t nnz = 0;
  SCAMD_HIP_CHECK(hipMemcpyAsync(&nnz, out_indptr + n, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  SCAMD_HIP_CHECK(hipStreamSynchronize(s));
  *nnz_host = nnz;
  return SCAMD_OK;
}
