// UMAP layout optimisation: the SGD behind `sc.tl.umap`
// (reference call site src/scanpy/tools/_umap.py:196-216 -> umap-learn `optimize_layout_euclidean`).
//
// MI355X-first formulation: SYNCHRONOUS and race-free.  umap-learn sweeps the graph samples sequentially (or, with
// parallel=True, lets numba threads race on the embedding); a GPU port of that scatters atomics over y.  Here every
// vertex v GATHERS: G lanes walk row v of the symmetric graph's CSR, read the epoch's snapshot of the embedding,
// accumulate the forces on v, and one lane writes y_out[v] -- no atomics, no races, bitwise reproducible.
//   * sample (v, u) of row v fires at epoch n when epoch_of_next_sample <= n (same schedule arrays as the reference,
//     one owner lane per sample); its attractive step moves v, and the mirrored sample (u, v) -- same weight, same
//     schedule -- moves v by the same clipped amount through `move_other`: the step counts twice;
//   * its negative samples are drawn with a counter-based hash of (seed, epoch, sample, p) instead of a per-vertex
//     tau88 state: order independent;
//   * alpha follows the reference (lowered after each epoch).
// Per epoch the kernel reads 12 B/sample of schedule + index, gathers 4*dim B per fired sample and per negative
// sample, and rewrites the embedding: HBM/gather-bound like the Leiden sweeps.
// oracle/umap.c `oracle_umap_synchronous` is the same scheme on the CPU.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "common.h"

namespace scamd {
namespace {

constexpr int UM_BLOCK = 256;
constexpr int UM_MAXD = 8;

__device__ __forceinline__ unsigned int um_hash32(unsigned int x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ int64_t um_neg_vertex(unsigned long long seed, int epoch, int64_t sample, int p,
                                                 int64_t n_vertices) {
  unsigned int h = um_hash32((unsigned int)seed ^ um_hash32((unsigned int)(seed >> 32) + 0x9E3779B9U * (unsigned int)epoch));
  h = um_hash32(h ^ (unsigned int)sample);
  h = um_hash32(h + 0x85EBCA6BU * (unsigned int)(sample >> 32) + 0xC2B2AE35U * (unsigned int)p);
  return (int64_t)(((unsigned long long)h * (unsigned long long)n_vertices) >> 32);
}
// x^b for x > 0 through the hardware exp2 / log2 (v_exp_f32, v_log_f32: ~1 ulp each).  The library powf costs ~200
// instructions per call and, with ~30 % of a wave's lanes inside the "sample fires" branch, made the kernel
// VALU-bound (7 calls per fired sample); the SGD does not need powf's last bit.
__device__ __forceinline__ float um_pow(float x, float b) { return __builtin_amdgcn_exp2f(b * __builtin_amdgcn_logf(x)); }
__device__ __forceinline__ float um_clip4(float v) { return v > 4.0f ? 4.0f : (v < -4.0f ? -4.0f : v); }

// G lanes per vertex (UM_BLOCK / G vertices per workgroup), DIM = embedding dimension (compile time for 2 and 3)
template <int G, int DIM>
__global__ __launch_bounds__(UM_BLOCK) void umap_epoch_kernel(
    int64_t n, int dim_rt, const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
    const float* __restrict__ eps_arr, float* __restrict__ next, float* __restrict__ next_neg,
    const float* __restrict__ yin, float* __restrict__ yout, int epoch, float alpha, float a, float b, float gamma,
    float neg_rate, unsigned long long seed) {
  const int dim = DIM > 0 ? DIM : dim_rt;
  constexpr int D = DIM > 0 ? DIM : UM_MAXD;
  const int sub = threadIdx.x % G;
  const int64_t ngroups = (int64_t)gridDim.x * (UM_BLOCK / G);
  for (int64_t v = (int64_t)blockIdx.x * (UM_BLOCK / G) + threadIdx.x / G; v < n; v += ngroups) {
    float cur[D], delta[D];
#pragma unroll
    for (int t = 0; t < D; ++t) {
      cur[t] = t < dim ? yin[v * dim + t] : 0.f;
      delta[t] = 0.f;
    }
    const int64_t e = indptr[v + 1];
    for (int64_t i = indptr[v] + sub; i < e; i += G) {
      const float eps = eps_arr[i];
      const float nx = next[i];
      if (!(eps > 0.f) || nx > (float)epoch) continue;
      const int64_t u = indices[i];
      float d2 = 0.f;
      float diff[D];
#pragma unroll
      for (int t = 0; t < D; ++t) {
        diff[t] = t < dim ? cur[t] - yin[u * dim + t] : 0.f;
        d2 += diff[t] * diff[t];
      }
      float coeff = 0.f;
      if (d2 > 0.f) {
        const float pw = um_pow(d2, b);  // d^(2b); d^(2(b-1)) = pw / d2
        coeff = (-2.0f * a * b * (pw / d2)) / (a * pw + 1.0f);
      }
#pragma unroll
      for (int t = 0; t < D; ++t) delta[t] += 2.0f * um_clip4(coeff * diff[t]);
      next[i] = nx + eps;
      const float eps_neg = eps / neg_rate;
      const float nn = next_neg[i];
      const int n_neg = (int)(((float)epoch - nn) / eps_neg);
      // negatives in batches of NB: all gathers of a batch are issued before any of them is used (a loop with one
      // dependent gather per iteration is latency-bound: 17 G gathers/s measured, vs 3x that batched)
      constexpr int NB = 6;
      for (int p0 = 0; p0 < n_neg; p0 += NB) {
        int64_t kk[NB];
        float oth[NB][D];
#pragma unroll
        for (int q = 0; q < NB; ++q) {
          kk[q] = (p0 + q < n_neg) ? um_neg_vertex(seed, epoch, i, p0 + q, n) : v;  // v itself = skipped below
#pragma unroll
          for (int t = 0; t < D; ++t) oth[q][t] = t < dim ? yin[kk[q] * dim + t] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
          if (kk[q] == v) continue;
          float e2 = 0.f;
#pragma unroll
          for (int t = 0; t < D; ++t) {
            diff[t] = t < dim ? cur[t] - oth[q][t] : 0.f;
            e2 += diff[t] * diff[t];
          }
          if (e2 > 0.f) {
            const float c2 = (2.0f * gamma * b) / ((0.001f + e2) * (a * um_pow(e2, b) + 1.0f));
#pragma unroll
            for (int t = 0; t < D; ++t) delta[t] += um_clip4(c2 * diff[t]);
          }
        }
      }
      next_neg[i] = nn + (float)n_neg * eps_neg;
    }
    // fixed-shape tree over the G lanes: the sum does not depend on timing
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) {
#pragma unroll
      for (int t = 0; t < D; ++t) delta[t] += __shfl_xor(delta[t], o);
    }
    if (sub == 0) {
#pragma unroll
      for (int t = 0; t < D; ++t)
        if (t < dim) yout[v * dim + t] = cur[t] + alpha * delta[t];
    }
  }
}

__global__ void umap_init_schedule_kernel(int64_t nnz, const float* __restrict__ eps, float neg_rate,
                                          float* __restrict__ next, float* __restrict__ next_neg) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += stride) {
    next[i] = eps[i];
    next_neg[i] = eps[i] / neg_rate;
  }
}

struct UmapBuffers {
  float* next;
  float* next_neg;
  float* y2;
};
void umap_carve(Workspace& ws, int64_t n, int64_t nnz, int dim, UmapBuffers* b) {
  b->next = ws.take<float>((size_t)std::max<int64_t>(nnz, 1));
  b->next_neg = ws.take<float>((size_t)std::max<int64_t>(nnz, 1));
  b->y2 = ws.take<float>((size_t)n * dim);
}

template <int G>
void launch_epoch(int dim, unsigned grid, hipStream_t s, int64_t n, const int64_t* indptr, const int32_t* indices,
                  const float* eps, float* next, float* next_neg, const float* yin, float* yout, int epoch, float alpha,
                  float a, float b, float gamma, float neg_rate, unsigned long long seed) {
  if (dim == 2)
    hipLaunchKernelGGL((umap_epoch_kernel<G, 2>), dim3(grid), dim3(UM_BLOCK), 0, s, n, dim, indptr, indices, eps, next,
                       next_neg, yin, yout, epoch, alpha, a, b, gamma, neg_rate, seed);
  else if (dim == 3)
    hipLaunchKernelGGL((umap_epoch_kernel<G, 3>), dim3(grid), dim3(UM_BLOCK), 0, s, n, dim, indptr, indices, eps, next,
                       next_neg, yin, yout, epoch, alpha, a, b, gamma, neg_rate, seed);
  else
    hipLaunchKernelGGL((umap_epoch_kernel<G, 0>), dim3(grid), dim3(UM_BLOCK), 0, s, n, dim, indptr, indices, eps, next,
                       next_neg, yin, yout, epoch, alpha, a, b, gamma, neg_rate, seed);
}

}  // namespace
}  // namespace scamd

using namespace scamd;

extern "C" size_t scamd_umap_workspace_bytes(int64_t n, int64_t nnz, int dim) {
  if (n <= 0 || nnz < 0 || dim < 1 || dim > UM_MAXD) return 0;
  Workspace ws(nullptr, 0);
  UmapBuffers b;
  umap_carve(ws, n, nnz, dim, &b);
  return ws.used();
}

extern "C" int scamd_umap_optimize_f32(const int64_t* indptr, const int32_t* indices, const float* epochs_per_sample,
                                       int64_t n, int64_t nnz, int dim, int n_epochs, double a, double b, double gamma,
                                       double initial_alpha, double negative_sample_rate, uint64_t seed, float* y,
                                       void* workspace, size_t workspace_bytes, scamd_stream_t stream) {
  SCAMD_REQUIRE(indptr && y && n >= 1 && nnz >= 0 && (nnz == 0 || (indices && epochs_per_sample)), SCAMD_EINVAL,
                "umap: null pointer or bad shape");
  SCAMD_REQUIRE(dim >= 1 && dim <= UM_MAXD, SCAMD_EUNSUPPORTED, "umap: n_components=%d (supported: 1..%d)", dim, UM_MAXD);
  SCAMD_REQUIRE(n_epochs >= 0 && negative_sample_rate > 0.0, SCAMD_EINVAL, "umap: bad n_epochs / negative_sample_rate");
  Workspace ws(workspace, workspace_bytes);
  UmapBuffers bf;
  umap_carve(ws, n, nnz, dim, &bf);
  SCAMD_REQUIRE(workspace && ws.ok, SCAMD_EWORKSPACE, "umap: workspace %zu < required %zu", workspace_bytes, ws.used());
  if (n_epochs == 0 || nnz == 0) return SCAMD_OK;
  hipLaunchKernelGGL(umap_init_schedule_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(nnz, 256), 4096)), dim3(256), 0,
                     stream, nnz, epochs_per_sample, (float)negative_sample_rate, bf.next, bf.next_neg);
  SCAMD_LAUNCH_CHECK();
  const int64_t avg = nnz / n;
  const int G = avg <= 12 ? 4 : (avg <= 24 ? 8 : (avg <= 96 ? 16 : 32));
  const unsigned grid = (unsigned)std::min<int64_t>(std::max<int64_t>(ceil_div(n, UM_BLOCK / G), 1), 256 * 32);
  float* yin = y;
  float* yout = bf.y2;
  for (int ep = 0; ep < n_epochs; ++ep) {
    // the reference lowers alpha after epoch n to initial_alpha (1 - n / n_epochs)
    const float alpha = (float)(initial_alpha * (1.0 - (double)(ep > 0 ? ep - 1 : 0) / (double)n_epochs));
    switch (G) {
      case 4: launch_epoch<4>(dim, grid, stream, n, indptr, indices, epochs_per_sample, bf.next, bf.next_neg, yin, yout, ep, alpha, (float)a, (float)b, (float)gamma, (float)negative_sample_rate, seed); break;
      case 8: launch_epoch<8>(dim, grid, stream, n, indptr, indices, epochs_per_sample, bf.next, bf.next_neg, yin, yout, ep, alpha, (float)a, (float)b, (float)gamma, (float)negative_sample_rate, seed); break;
      case 16: launch_epoch<16>(dim, grid, stream, n, indptr, indices, epochs_per_sample, bf.next, bf.next_neg, yin, yout, ep, alpha, (float)a, (float)b, (float)gamma, (float)negative_sample_rate, seed); break;
      default: launch_epoch<32>(dim, grid, stream, n, indptr, indices, epochs_per_sample, bf.next, bf.next_neg, yin, yout, ep, alpha, (float)a, (float)b, (float)gamma, (float)negative_sample_rate, seed); break;
    }
    SCAMD_LAUNCH_CHECK();
    std::swap(yin, yout);
  }
  if (yin != y) SCAMD_HIP_CHECK(hipMemcpyAsync(y, yin, sizeof(float) * (size_t)n * dim, hipMemcpyDeviceToDevice, stream));
  return SCAMD_OK;
}
