// Device exclusive prefix sum (int32 counts -> int64 offsets), three small kernels.
#pragma once
#include "common.h"

namespace scamd {

constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 8;                      // per thread
constexpr int SCAN_CHUNK = SCAN_BLOCK * SCAN_ITEMS;  // per block

static __device__ inline int64_t block_exclusive_scan_i64(int64_t v, int64_t* total, int64_t* sh /*[5]*/) {
  // inclusive scan inside each wave, then across the 4 waves of the block
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int64_t inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int64_t t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  if (lane == 63) sh[w] = inc;
  __syncthreads();
  int64_t woff = 0;
  for (int i = 0; i < w; ++i) woff += sh[i];
  if (threadIdx.x == SCAN_BLOCK - 1) sh[4] = woff + inc;
  __syncthreads();
  *total = sh[4];
  __syncthreads();
  return woff + inc - v;
}

static __global__ __launch_bounds__(SCAN_BLOCK) void scan_block_sums_kernel(const int* __restrict__ in, int64_t n,
                                                                    int64_t* __restrict__ block_sums) {
  __shared__ int64_t sh[5];
  const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_ITEMS;
  int64_t s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i)
    if (base + i < n) s += in[base + i];
  int64_t total;
  block_exclusive_scan_i64(s, &total, sh);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

static __global__ __launch_bounds__(SCAN_BLOCK) void scan_top_kernel(int64_t* __restrict__ block_sums, int nb,
                                                             int64_t* __restrict__ grand_total) {
  __shared__ int64_t sh[5];
  int64_t carry = 0;
  for (int base = 0; base < nb; base += SCAN_BLOCK) {
    int i = base + threadIdx.x;
    int64_t v = (i < nb) ? block_sums[i] : 0;
    int64_t total;
    int64_t ex = block_exclusive_scan_i64(v, &total, sh);
    if (i < nb) block_sums[i] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0 && grand_total) *grand_total = carry;
}

// out has n+1 entries; out[n] = total.
static __global__ __launch_bounds__(SCAN_BLOCK) void scan_apply_kernel(const int* __restrict__ in, int64_t n,
                                                               const int64_t* __restrict__ block_offs,
                                                               int64_t* __restrict__ out) {
  __shared__ int64_t sh[5];
  const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_ITEMS;
  int v[SCAN_ITEMS];
  int64_t s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    v[i] = (base + i < n) ? in[base + i] : 0;
    s += v[i];
  }
  int64_t total;
  int64_t ex = block_exclusive_scan_i64(s, &total, sh) + block_offs[blockIdx.x];
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    if (base + i < n) out[base + i] = ex;
    ex += v[i];
    if (base + i == n - 1) out[n] = ex;
  }
}

inline int scan_num_blocks(int64_t n) { return (int)((n + SCAN_CHUNK - 1) / SCAN_CHUNK); }

// exclusive scan of int32 `in[n]` into int64 `out[n+1]`; `block_tmp` needs scan_num_blocks(n)+1 int64.
inline int exclusive_scan_i32_i64(const int* in, int64_t n, int64_t* out, int64_t* block_tmp, hipStream_t s) {
  if (n <= 0) {
    SCAMD_HIP_CHECK(hipMemsetAsync(out, 0, sizeof(int64_t), s));
    return SCAMD_OK;
  }
  const int nb = scan_num_blocks(n);
  hipLaunchKernelGGL(scan_block_sums_kernel, dim3(nb), dim3(SCAN_BLOCK), 0, s, in, n, block_tmp);
  SCAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(SCAN_BLOCK), 0, s, block_tmp, nb, block_tmp + nb);
  SCAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(scan_apply_kernel, dim3(nb), dim3(SCAN_BLOCK), 0, s, in, n, block_tmp, out);
  SCAMD_LAUNCH_CHECK();
  return SCAMD_OK;
}

}  // namespace scamd
