// Sparse building blocks of the PCA stage on gfx950: everything that touches the n x g CSR matrix.
//
// Replaces the data passes of sklearn PCA(svd_solver='arpack') on sparse input
// (sklearn/decomposition/_pca.py:704-793 called from src/scanpy/preprocessing/_pca/__init__.py:
// 287-308): mean_variance_axis and the implicitly centred operator of
// sklearn/utils/sparsefuncs.py:718-742 / src/scanpy/preprocessing/_pca/_compat.py:43-56,
//     (X - 1 mu^T) Z   = X Z - 1 (mu^T Z)          -> scamd_spmm_csr_f32 with `shift`
//     (X - 1 mu^T)^T Y = X^T Y - mu (1^T Y)        -> scamd_spmm_csr_f32_f64acc on the CSC copy
// applied to BLOCKS of l <= 128 vectors (block Krylov / subspace iteration on the host side)
// instead of ARPACK's one-vector-at-a-time Lanczos.  All kernels are HBM/L2-bound streaming
// kernels; all reductions have a fixed order (bitwise reproducible run to run).
#include "common.h"

#include <cstdlib>
#include "scan.h"

#include <algorithm>

namespace scamd {

// ------------------------------------------------------------------------------------------------
// per-row sum / sum of squares in float64 (applied to the CSC copy it yields per-gene statistics)
// one wave per row; fixed order: lane-strided partials, then xor-butterfly
// ------------------------------------------------------------------------------------------------
__global__ void csr_row_stats_kernel(const int64_t* __restrict__ indptr, const float* __restrict__ data,
                                     int64_t n_rows, double* __restrict__ sum, double* __restrict__ sumsq) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int64_t b = indptr[row], e = indptr[row + 1];
  double s = 0.0, q = 0.0;
  for (int64_t p = b + lane; p < e; p += 64) {
    double v = (double)data[p];
    s += v;
    q = fma(v, v, q);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o);
    q += __shfl_xor(q, o);
  }
  if (lane == 0) {
    sum[row] = s;
    sumsq[row] = q;
  }
}

// ------------------------------------------------------------------------------------------------
// deterministic CSR -> CSC (stable counting sort by column, chunk = contiguous row range per wave)
// ------------------------------------------------------------------------------------------------
// T1: per-chunk column histogram (one wave per chunk, histogram in LDS)
__global__ __launch_bounds__(64) void tr_hist_kernel(const int64_t* __restrict__ indptr,
                                                     const int32_t* __restrict__ indices, int64_t n, int64_t g,
                                                     int rows_per_chunk, unsigned int* __restrict__ hist) {
  extern __shared__ unsigned int lh[];
  const int lane = threadIdx.x;
  const int64_t chunk = blockIdx.x;
  for (int64_t j = lane; j < g; j += 64) lh[j] = 0;
  __syncthreads();
  const int64_t r0 = chunk * rows_per_chunk, r1 = std::min<int64_t>(n, r0 + rows_per_chunk);
  const int64_t b = indptr[r0], e = indptr[r1];
  // columns repeat across rows, so two lanes of one instruction may hit the same counter:
  // integer LDS atomics keep the count exact (and integer addition is order independent)
  for (int64_t p = b + lane; p < e; p += 64) atomicAdd(&lh[indices[p]], 1u);
  __syncthreads();
  unsigned int* out = hist + chunk * g;
  for (int64_t j = lane; j < g; j += 64) out[j] = lh[j];
}

// T2a: partial[grp][j] = sum over the chunks of group grp
__global__ void tr_group_sum_kernel(const unsigned int* __restrict__ hist, int64_t n_chunks, int64_t g,
                                    int chunks_per_group, unsigned long long* __restrict__ partial) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t grp = blockIdx.y;
  if (j >= g) return;
  const int64_t c0 = grp * chunks_per_group, c1 = std::min<int64_t>(n_chunks, c0 + chunks_per_group);
  unsigned long long s = 0;
  for (int64_t c = c0; c < c1; ++c) s += hist[c * g + j];
  partial[grp * g + j] = s;
}

// T2b: per column: exclusive scan over groups (in place) and column total
__global__ void tr_group_scan_kernel(unsigned long long* __restrict__ partial, int64_t n_groups, int64_t g,
                                     int* __restrict__ col_count) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= g) return;
  unsigned long long run = 0;
  for (int64_t grp = 0; grp < n_groups; ++grp) {
    unsigned long long v = partial[grp * g + j];
    partial[grp * g + j] = run;
    run += v;
  }
  col_count[j] = (int)run;  // a column holds at most n < 2^31 entries
}

// T2c: hist[c][j] <- offset of chunk c inside column j
__global__ void tr_chunk_scan_kernel(unsigned int* __restrict__ hist, int64_t n_chunks, int64_t g,
                                     int chunks_per_group, const unsigned long long* __restrict__ partial) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t grp = blockIdx.y;
  if (j >= g) return;
  const int64_t c0 = grp * chunks_per_group, c1 = std::min<int64_t>(n_chunks, c0 + chunks_per_group);
  unsigned int run = (unsigned int)partial[grp * g + j];
  for (int64_t c = c0; c < c1; ++c) {
    unsigned int v = hist[c * g + j];
    hist[c * g + j] = run;
    run += v;
  }
}

// T3: scatter.  One wave per chunk walks its rows IN ORDER, so within a column the row ids come out
// ascending (stable).  Columns are unique inside a CSR row, hence no two lanes of one step collide.
__global__ __launch_bounds__(64) void tr_scatter_kernel(const int64_t* __restrict__ indptr,
                                                        const int32_t* __restrict__ indices,
                                                        const float* __restrict__ data, int64_t n, int64_t g,
                                                        int rows_per_chunk, const unsigned int* __restrict__ hist,
                                                        const int64_t* __restrict__ t_indptr,
                                                        int32_t* __restrict__ t_indices,
                                                        float* __restrict__ t_data) {
  extern __shared__ unsigned int lh[];
  const int lane = threadIdx.x;
  const int64_t chunk = blockIdx.x;
  const unsigned int* in = hist + chunk * g;
  for (int64_t j = lane; j < g; j += 64) lh[j] = in[j];
  __syncthreads();
  const int64_t r0 = chunk * rows_per_chunk, r1 = std::min<int64_t>(n, r0 + rows_per_chunk);
  for (int64_t r = r0; r < r1; ++r) {
    const int64_t b = indptr[r], e = indptr[r + 1];
    for (int64_t p = b + lane; p < e; p += 64) {
      const int c = indices[p];
      const unsigned int rel = lh[c];
      lh[c] = rel + 1;
      const int64_t pos = t_indptr[c] + rel;
      t_indices[pos] = (int32_t)r;
      t_data[pos] = data[p];
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------------------
// Y = A * B - 1 shift^T  (float32).  One wave per CSR row; lane = output column (CPL columns/lane).
// The row's (index, value) pairs are loaded 64 at a time and broadcast with readlane; B rows
// (l*4 bytes, L2-resident panel) are read coalesced.
// Memory-level parallelism is the whole game here: the first version waited for every B row before its fma (one load
// in flight per wave, 80 % of the wave-cycles parked on s_waitcnt: profiles/r02p_pca_stage_pmc1.csv).  Now GU = 8 B rows
// are requested back to back and consumed in the same order (the sum is bit for bit the old one), the next 64 entries
// of the row and the next row's extent are requested before the current ones are used.
// Round 4 tried the other mapping -- a workgroup owns 256 rows (one per thread, accumulators in registers), the rows of B
// staged through LDS in gene tiles, bit-identical sums: 14.05 vs 14.05 ms for the PCA stage (profiles/r04f_bench_spmm_lds*.json).
// A thread per row reads its entries 4 bytes at a time from 64 different cache lines per load instruction, and 12 waves
// of such rows do not fit the 32 KB vector L1: the L2 traffic this kernel spends on B rows came back as CSR sectors.
// Removed again; what would help is a row-major staging of the CSR block itself, i.e. a different storage order.
// ------------------------------------------------------------------------------------------------
template <int CPL>
__global__ __launch_bounds__(256) void spmm_rows_f32_kernel(const int64_t* __restrict__ indptr,
                                                            const int32_t* __restrict__ indices,
                                                            const float* __restrict__ data, int64_t n,
                                                            const float* __restrict__ b, int l,
                                                            const float* __restrict__ shift,
                                                            float* __restrict__ y) {
  constexpr int GU = 8;
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  float sh[CPL];
  bool colok[CPL];
  int colc[CPL];  // lanes past the last column read the last column (no divergence around the gathers), store nothing
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int col = lane + 64 * c;
    colok[c] = col < l;
    colc[c] = min(col, l - 1);
    sh[c] = (shift && colok[c]) ? shift[col] : 0.f;
  }
  if (wave >= n) return;
  int64_t rb = indptr[wave], re = indptr[wave + 1];
  for (int64_t row = wave; row < n; row += nwaves) {
    // the next row's extent (raw: consumed at the bottom of the loop)
    const int64_t nrow = row + nwaves < n ? row + nwaves : row;
    const int64_t nrb = indptr[nrow], nre = indptr[nrow + 1];
    float acc[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) acc[c] = 0.f;
    int ci = 0;
    float cv = 0.f;
    if (rb + lane < re) {
      ci = indices[rb + lane];
      cv = data[rb + lane];
    }
    for (int64_t p0 = rb; p0 < re; p0 += 64) {
      int ci_n = 0;
      float cv_n = 0.f;
      if (p0 + 64 + lane < re) {
        ci_n = indices[p0 + 64 + lane];
        cv_n = data[p0 + 64 + lane];
      }
      const int cnt = (int)std::min<int64_t>(64, re - p0);
      for (int u = 0; u < cnt; u += GU) {
        float bv[GU][CPL], vv[GU];
#pragma unroll
        for (int t = 0; t < GU; ++t) {
          // past the end of the row: the row's last entry again, its product is not added below
          const int uu = min(u + t, cnt - 1);
          const int j = __builtin_amdgcn_readlane(ci, uu);
          vv[t] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cv), uu));
          const float* brow = b + (int64_t)j * l;
#pragma unroll
          for (int c = 0; c < CPL; ++c) bv[t][c] = brow[colc[c]];
        }
#pragma unroll
        for (int t = 0; t < GU; ++t) {
          if (u + t < cnt) {
#pragma unroll
            for (int c = 0; c < CPL; ++c) acc[c] = fmaf(vv[t], bv[t][c], acc[c]);
          }
        }
      }
      ci = ci_n;
      cv = cv_n;
    }
#pragma unroll
    for (int c = 0; c < CPL; ++c)
      if (colok[c]) y[row * l + lane + 64 * c] = acc[c] - sh[c];
    rb = nrb;
    re = nre;
  }
}

// ------------------------------------------------------------------------------------------------
// W = A * B with float64 accumulation, rows split into fixed segments of SEG stored entries so that
// long rows (genes of the CSC copy: ~0.05*n entries) spread over many waves; partial sums are
// combined in segment order (deterministic).
// ------------------------------------------------------------------------------------------------
constexpr int SEG = 2048;

__global__ void seg_count_kernel(const int64_t* __restrict__ indptr, int64_t n_rows, int* __restrict__ nseg) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n_rows) nseg[r] = (int)((indptr[r + 1] - indptr[r] + SEG - 1) / SEG);
}

template <int CPL>
__global__ __launch_bounds__(256) void spmm_seg_f64_kernel(const int64_t* __restrict__ indptr,
                                                           const int32_t* __restrict__ indices,
                                                           const float* __restrict__ data, int64_t n_rows,
                                                           const int64_t* __restrict__ seg_ptr,
                                                           const float* __restrict__ b, int l,
                                                           double* __restrict__ partial) {
  const int lane = threadIdx.x & 63;
  const int64_t seg = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t total = seg_ptr[n_rows];
  if (seg >= total) return;
  // row = last r with seg_ptr[r] <= seg
  int64_t lo = 0, hi = n_rows;
  while (hi - lo > 1) {
    int64_t mid = (lo + hi) >> 1;
    if (seg_ptr[mid] <= seg) lo = mid;
    else hi = mid;
  }
  const int64_t row = lo;
  const int64_t rb = indptr[row] + (seg - seg_ptr[row]) * SEG;
  const int64_t re = std::min<int64_t>(indptr[row + 1], rb + SEG);
  double acc[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) acc[c] = 0.0;
  for (int64_t p0 = rb; p0 < re; p0 += 64) {
    const int64_t p = p0 + lane;
    int ci = 0;
    float cv = 0.f;
    if (p < re) {
      ci = indices[p];
      cv = data[p];
    }
    const int cnt = (int)std::min<int64_t>(64, re - p0);
    for (int u = 0; u < cnt; ++u) {
      const int j = __builtin_amdgcn_readlane(ci, u);
      const double v = (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(cv), u));
      const float* brow = b + (int64_t)j * l;
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        int col = lane + 64 * c;
        if (col < l) acc[c] = fma(v, (double)brow[col], acc[c]);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    int col = lane + 64 * c;
    if (col < l) partial[seg * l + col] = acc[c];
  }
}

__global__ void spmm_seg_reduce_kernel(const int64_t* __restrict__ seg_ptr, int64_t n_rows, int l,
                                       const double* __restrict__ partial, const double* __restrict__ scale,
                                       const double* __restrict__ colsum, double* __restrict__ w) {
  const int64_t row = blockIdx.x;
  const int col = threadIdx.x;
  if (row >= n_rows || col >= l) return;
  double s = 0.0;
  for (int64_t sg = seg_ptr[row]; sg < seg_ptr[row + 1]; ++sg) s += partial[sg * l + col];
  if (scale && colsum) s -= scale[row] * colsum[col];
  w[row * l + col] = s;
}

// ------------------------------------------------------------------------------------------------
// column sums of a dense float32 [n, l] panel in float64 (two fixed-order stages)
// ------------------------------------------------------------------------------------------------
constexpr int COLSUM_BLOCKS = 512;

__global__ __launch_bounds__(256) void colsum_stage1_kernel(const float* __restrict__ y, int64_t n, int l,
                                                            double* __restrict__ partial) {
  // thread (r, c): c = column (threadIdx.x % 128 slots), r = row lane
  const int col = threadIdx.x & 127;
  const int rl = threadIdx.x >> 7;  // 0..1
  const int64_t rows_per_block = (n + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = std::min<int64_t>(n, r0 + rows_per_block);
  double s = 0.0;
  if (col < l)
    for (int64_t r = r0 + rl; r < r1; r += 2) s += (double)y[r * l + col];
  __shared__ double sh[256];
  sh[threadIdx.x] = s;
  __syncthreads();
  if (rl == 0 && col < l) partial[(int64_t)blockIdx.x * l + col] = sh[col] + sh[128 + col];
}

__global__ void colsum_stage2_kernel(const double* __restrict__ partial, int nb, int l, double* __restrict__ out) {
  const int col = threadIdx.x;
  if (col >= l) return;
  double s = 0.0;
  for (int b = 0; b < nb; ++b) s += partial[(int64_t)b * l + col];
  out[col] = s;
}

struct TransposePlan {
  int rows_per_chunk;
  int64_t n_chunks;
  int chunks_per_group;
  int64_t n_groups;
};

static TransposePlan transpose_plan(int64_t n, int64_t g) {
  TransposePlan p;
  // keep the histogram table (n_chunks * g * 4 B) around 64 MB and at least ~2k chunks for parallelism
  int64_t rpc = 256;
  while ((n + rpc - 1) / rpc * g * 4 > ((int64_t)96 << 20)) rpc *= 2;
  p.rows_per_chunk = (int)rpc;
  p.n_chunks = (n + rpc - 1) / rpc;
  p.chunks_per_group = 64;
  p.n_groups = (p.n_chunks + 63) / 64;
  return p;
}

struct TransposeBuffers {
  unsigned int* hist; unsigned long long* partial; int* col_count; int64_t* scan_tmp;
};

static void transpose_carve(Workspace& ws, const TransposePlan& p, int64_t g, TransposeBuffers* b) {
  b->hist = ws.take<unsigned int>((size_t)p.n_chunks * g);
  b->partial = ws.take<unsigned long long>((size_t)p.n_groups * g);
  b->col_count = ws.take<int>((size_t)g);
  b->scan_tmp = ws.take<int64_t>((size_t)scan_num_blocks(g) + 2);
}

struct SegBuffers {
  int* nseg; int64_t* seg_ptr; int64_t* scan_tmp; double* partial;
};

static void seg_carve(Workspace& ws, int64_t n_rows, int64_t nnz, int l, SegBuffers* b) {
  b->nseg = ws.take<int>((size_t)n_rows);
  b->seg_ptr = ws.take<int64_t>((size_t)n_rows + 1);
  b->scan_tmp = ws.take<int64_t>((size_t)scan_num_blocks(n_rows) + 2);
  b->partial = ws.take<double>((size_t)(nnz / SEG + n_rows + 1) * l);
}

}  // namespace scamd

using namespace scamd;

extern "C" int scamd_csr_row_stats_f32(const int64_t* indptr, const float* data, int64_t n_rows, double* row_sum,
                                       double* row_sumsq, scamd_stream_t stream) {
  SCAMD_REQUIRE(indptr && data && row_sum && row_sumsq && n_rows >= 0, SCAMD_EINVAL, "row_stats: bad argument");
  if (n_rows == 0) return SCAMD_OK;
  hipLaunchKernelGGL(csr_row_stats_kernel, dim3(ceil_div(n_rows, 4)), dim3(256), 0, stream, indptr, data, n_rows,
                     row_sum, row_sumsq);
  SCAMD_LAUNCH_CHECK();
  return SCAMD_OK;
}

extern "C" size_t scamd_csr_transpose_workspace_bytes(int64_t n, int64_t g, int64_t nnz) {
  (void)nnz;
  if (n <= 0 || g <= 0) return 0;
  TransposePlan p = transpose_plan(n, g);
  Workspace ws(nullptr, 0);
  TransposeBuffers b;
  transpose_carve(ws, p, g, &b);
  return ws.used();
}

extern "C" int scamd_csr_transpose_f32(const int64_t* indptr, const int32_t* indices, const float* data, int64_t n,
                                       int64_t g, int64_t nnz, int64_t* t_indptr, int32_t* t_indices,
                                       float* t_data, void* workspace, size_t workspace_bytes,
                                       scamd_stream_t stream) {
  SCAMD_REQUIRE(indptr && indices && data && t_indptr && t_indices && t_data, SCAMD_EINVAL, "transpose: null pointer");
  SCAMD_REQUIRE(n >= 1 && g >= 1 && nnz >= 0, SCAMD_EINVAL, "transpose: bad shape");
  SCAMD_REQUIRE(n < ((int64_t)1 << 31) && g < ((int64_t)1 << 31), SCAMD_EUNSUPPORTED, "transpose: dims exceed int32");
  SCAMD_REQUIRE(g * 4 <= 160 * 1024 - 64, SCAMD_EUNSUPPORTED, "transpose: g=%lld exceeds the LDS histogram (max 40944)",
                (long long)g);
  TransposePlan p = transpose_plan(n, g);
  Workspace ws(workspace, workspace_bytes);
  TransposeBuffers b;
  transpose_carve(ws, p, g, &b);
  SCAMD_REQUIRE(workspace && ws.ok, SCAMD_EWORKSPACE, "transpose: workspace %zu < required %zu", workspace_bytes,
                ws.used());
  hipStream_t s = stream;
  const size_t lds = (size_t)g * 4;
  SCAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tr_hist_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  SCAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tr_scatter_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(tr_hist_kernel, dim3((unsigned)p.n_chunks), dim3(64), lds, s, indptr, indices, n, g,
                     p.rows_per_chunk, b.hist);
  SCAMD_LAUNCH_CHECK();
  dim3 grid2(ceil_div(g, 256), (unsigned)p.n_groups);
  hipLaunchKernelGGL(tr_group_sum_kernel, grid2, dim3(256), 0, s, b.hist, p.n_chunks, g, p.chunks_per_group,
                     b.partial);
  SCAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(tr_group_scan_kernel, dim3(ceil_div(g, 256)), dim3(256), 0, s, b.partial, p.n_groups, g,
                     b.col_count);
  SCAMD_LAUNCH_CHECK();
  int rc = exclusive_scan_i32_i64(b.col_count, g, t_indptr, b.scan_tmp, s);
  if (rc != SCAMD_OK) return rc;
  hipLaunchKernelGGL(tr_chunk_scan_kernel, grid2, dim3(256), 0, s, b.hist, p.n_chunks, g, p.chunks_per_group,
                     b.partial);
  SCAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(tr_scatter_kernel, dim3((unsigned)p.n_chunks), dim3(64), lds, s, indptr, indices, data, n, g,
                     p.rows_per_chunk, b.hist, t_indptr, t_indices, t_data);
  SCAMD_LAUNCH_CHECK();
  return SCAMD_OK;
}

namespace scamd {
// Round 6, l <= 64 (the scores of sc.pp.pca): the kernel above issues ONE 200-byte load instruction per stored entry and
// is bound by the rate of those instructions -- halving the bytes (32 columns) did not move its 1.9 ms (tools/spmm_probe.py).
// Here a load instruction fetches the B rows of FOUR entries: lane = (entry slot 0..3, column quad 0..15), 16 bytes per lane;
// the four slots' partial sums meet in a fixed shuffle tree at the end of the row (a different, still fixed, summation order).
// The last quad of a row of B is shifted back to columns [l - 4, l) so that no lane reads past the row.
__global__ __launch_bounds__(256) void spmm_rows_quad_f32_kernel(const int64_t* __restrict__ indptr,
                                                                 const int32_t* __restrict__ indices,
                                                                 const float* __restrict__ data, int64_t n,
                                                                 const float* __restrict__ b, int l,
                                                                 const float* __restrict__ shift, float* __restrict__ y) {
  constexpr int GU = 4;  // loads in flight per lane: 16 entries of the row (8: 1.65 ms against 1.58)
  const int lane = threadIdx.x & 63, slot = lane >> 4, cq = lane & 15;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const int nquad = (l + 3) / 4;
  const int col0 = min(4 * min(cq, nquad - 1), l - 4);  // first of the lane's four columns
  const bool own_quad = cq < nquad;
  float sh[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) sh[c] = shift ? shift[col0 + c] : 0.f;
  if (wave >= n) return;
  int64_t rb = indptr[wave], re = indptr[wave + 1];
  for (int64_t row = wave; row < n; row += nwaves) {
    const int64_t nrow = row + nwaves < n ? row + nwaves : row;
    const int64_t nrb = indptr[nrow], nre = indptr[nrow + 1];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int ci = 0;
    float cv = 0.f;
    if (rb + lane < re) {
      ci = indices[rb + lane];
      cv = data[rb + lane];
    }
    for (int64_t p0 = rb; p0 < re; p0 += 64) {
      int ci_n = 0;
      float cv_n = 0.f;
      if (p0 + 64 + lane < re) {
        ci_n = indices[p0 + 64 + lane];
        cv_n = data[p0 + 64 + lane];
      }
      const int cnt = (int)std::min<int64_t>(64, re - p0);
      for (int u = 0; u < cnt; u += 4 * GU) {
        float4 bv[GU];
        float vv[GU];
#pragma unroll
        for (int t = 0; t < GU; ++t) {
          // the slot's entry of this round; past the end of the row: the row's last entry with value 0
          const int e = u + 4 * t + slot;
          const int ec = min(e, cnt - 1);
          const int j = __shfl(ci, ec);
          const float v = __shfl(cv, ec);
          vv[t] = e < cnt ? v : 0.f;
          const float* bp = b + (int64_t)j * l + col0;
          // (rows of B are l * 4 bytes apart: 8-byte alignment at best, so the 16 bytes come as four dwords the compiler may merge)
          bv[t] = make_float4(bp[0], bp[1], bp[2], bp[3]);
        }
#pragma unroll
        for (int t = 0; t < GU; ++t) {
          acc[0] = fmaf(vv[t], bv[t].x, acc[0]);
          acc[1] = fmaf(vv[t], bv[t].y, acc[1]);
          acc[2] = fmaf(vv[t], bv[t].z, acc[2]);
          acc[3] = fmaf(vv[t], bv[t].w, acc[3]);
        }
      }
      ci = ci_n;
      cv = cv_n;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      acc[c] += __shfl_xor(acc[c], 16);
      acc[c] += __shfl_xor(acc[c], 32);
    }
    if (slot == 0 && own_quad) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (col0 + c >= 4 * cq) y[row * l + col0 + c] = acc[c] - sh[c];
    }
    rb = nrb;
    re = nre;
  }
}

}  // namespace scamd

extern "C" int scamd_spmm_csr_f32(const int64_t* indptr, const int32_t* indices, const float* data, int64_t n,
                                  int64_t g, const float* b, int l, const float* shift, float* y,
                                  scamd_stream_t stream) {
  SCAMD_REQUIRE(indptr && indices && data && b && y, SCAMD_EINVAL, "spmm: null pointer");
  SCAMD_REQUIRE(n >= 0 && g >= 1 && l >= 1 && l <= 256, SCAMD_EINVAL, "spmm: bad shape n=%lld g=%lld l=%d (at most 256 columns)",
                (long long)n, (long long)g, l);
  if (n == 0) return SCAMD_OK;
  const int blocks = (int)std::min<int64_t>((n + 3) / 4, 256 * 32);
  static const bool quad = [] {  // (A/B knob: SCAMD_SPMM_QUAD=0 runs the one-entry-per-load kernel)
    const char* e = getenv("SCAMD_SPMM_QUAD");
    return !(e && e[0] == '0');
  }();
  if (l <= 64 && l >= 4 && quad)
    hipLaunchKernelGGL(spmm_rows_quad_f32_kernel, dim3(blocks), dim3(256), 0, stream, indptr, indices, data, n, b, l, shift, y);
  else if (l <= 64)
    hipLaunchKernelGGL(spmm_rows_f32_kernel<1>, dim3(blocks), dim3(256), 0, stream, indptr, indices, data, n, b, l,
                       shift, y);
  else if (l <= 128)
    hipLaunchKernelGGL(spmm_rows_f32_kernel<2>, dim3(blocks), dim3(256), 0, stream, indptr, indices, data, n, b, l,
                       shift, y);
  else if (l <= 192)  // (round 6: n_comps beyond 128 -- the deflated eigensolver of csrc/dense.hip delivers them)
    hipLaunchKernelGGL(spmm_rows_f32_kernel<3>, dim3(blocks), dim3(256), 0, stream, indptr, indices, data, n, b, l,
                       shift, y);
  else
    hipLaunchKernelGGL(spmm_rows_f32_kernel<4>, dim3(blocks), dim3(256), 0, stream, indptr, indices, data, n, b, l,
                       shift, y);
  SCAMD_LAUNCH_CHECK();
  return SCAMD_OK;
}

extern "C" size_t scamd_spmm_f64acc_workspace_bytes(int64_t n_rows, int64_t nnz, int l) {
  if (n_rows <= 0 || l <= 0) return 0;
  Workspace ws(nullptr, 0);
  SegBuffers b;
  seg_carve(ws, n_rows, nnz, l, &b);
  return ws.used();
}

extern "C" int scamd_spmm_csr_f32_f64acc(const int64_t* indptr, const int32_t* indices, const float* data,
                                         int64_t n_rows, int64_t nnz, const float* b, int l, const double* scale,
                                         const double* colsum, double* w, void* workspace, size_t workspace_bytes,
                                         scamd_stream_t stream) {
  SCAMD_REQUIRE(indptr && indices && data && b && w, SCAMD_EINVAL, "spmm_f64acc: null pointer");
  SCAMD_REQUIRE(n_rows >= 1 && nnz >= 0 && l >= 1 && l <= 128, SCAMD_EINVAL, "spmm_f64acc: bad shape");
  Workspace ws(workspace, workspace_bytes);
  SegBuffers sb;
  seg_carve(ws, n_rows, nnz, l, &sb);
  SCAMD_REQUIRE(workspace && ws.ok, SCAMD_EWORKSPACE, "spmm_f64acc: workspace %zu < required %zu", workspace_bytes,
                ws.used());
  hipStream_t s = stream;
  hipLaunchKernelGGL(seg_count_kernel, dim3(ceil_div(n_rows, 256)), dim3(256), 0, s, indptr, n_rows, sb.nseg);
  SCAMD_LAUNCH_CHECK();
  int rc = exclusive_scan_i32_i64(sb.nseg, n_rows, sb.seg_ptr, sb.scan_tmp, s);
  if (rc != SCAMD_OK) return rc;
  const int64_t max_seg = nnz / SEG + n_rows + 1;
  const int blocks = (int)((max_seg + 3) / 4);
  if (l <= 64)
    hipLaunchKernelGGL(spmm_seg_f64_kernel<1>, dim3(blocks), dim3(256), 0, s, indptr, indices, data, n_rows,
                       sb.seg_ptr, b, l, sb.partial);
  else
    hipLaunchKernelGGL(spmm_seg_f64_kernel<2>, dim3(blocks), dim3(256), 0, s, indptr, indices, data, n_rows,
                       sb.seg_ptr, b, l, sb.partial);
  SCAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(spmm_seg_reduce_kernel, dim3((unsigned)n_rows), dim3(128), 0, s, sb.seg_ptr, n_rows, l,
                     sb.partial, scale, colsum, w);
  SCAMD_LAUNCH_CHECK();
  return SCAMD_OK;
}

extern "C" size_t scamd_colsum_workspace_bytes(int l) { return (size_t)COLSUM_BLOCKS * 128 * sizeof(double) + 256; }

extern "C" int scamd_colsum_f32_f64(const float* y, int64_t n, int l, double* colsum, void* workspace,
                                    size_t workspace_bytes, scamd_stream_t stream) {
  SCAMD_REQUIRE(y && colsum && n >= 1 && l >= 1 && l <= 128, SCAMD_EINVAL, "colsum: bad argument");
  SCAMD_REQUIRE(workspace && workspace_bytes >= scamd_colsum_workspace_bytes(l), SCAMD_EWORKSPACE,
                "colsum: workspace too small");
  double* partial = reinterpret_cast<double*>(workspace);
  const int nb = (int)std::min<int64_t>(COLSUM_BLOCKS, n);
  hipLaunchKernelGGL(colsum_stage1_kernel, dim3(nb), dim3(256), 0, stream, y, n, l, partial);
  SCAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_stage2_kernel, dim3(1), dim3(128), 0, stream, partial, nb, l, colsum);
  SCAMD_LAUNCH_CHECK();
  return SCAMD_OK;
}
