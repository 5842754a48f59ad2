// Microbenchmark: issue rate of v_mfma_f32_32x32x2_f32 as a function of (waves per SIMD, independent accumulators).
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NACC>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
  f32x16 acc[NACC];
  for (int j = 0; j < NACC; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = (float)(threadIdx.x + j + r);
  float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 25; ++s) {
#pragma unroll
      for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int j = 0; j < NACC; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int blocks_per_cu, int iters) {
  float* out;
  hipMalloc(&out, sizeof(float) * 256 * 256 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grid = 256 * blocks_per_cu;
  probe<NACC><<<grid, 256>>>(out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<NACC><<<grid, 256>>>(out, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double nmfma = (double)grid * 4 * iters * 25 * NACC;  // wave-level instructions
  double flops = nmfma * 2.0 * 32 * 32 * 2;
  // cycles per MFMA per SIMD at 2.4 GHz: each SIMD executed nmfma / 1024 instructions
  double cyc = ms * 1e-3 * 2.4e9 / (nmfma / 1024.0);
  printf("NACC=%d waves/SIMD=%d: %.2f ms  %.1f TFLOP/s  %.1f cyc/MFMA/SIMD (at 2.4 GHz)\n", NACC, blocks_per_cu, ms,
         flops / ms / 1e9, cyc);
  hipFree(out);
}

int main() {
  const int iters = 20000;
  run<1>(1, iters);
  run<2>(1, iters);
  run<4>(1, iters);
  run<1>(2, iters);
  run<2>(2, iters / 2);
  run<1>(4, iters / 2);
  return 0;
}
