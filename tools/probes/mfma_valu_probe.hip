// Microbenchmark: cost of VALU / LDS fillers issued in the shadow of dependent v_mfma_f32_32x32x2_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;

// KIND 0: v_add_f32 fillers (independent regs); 1: v_cmp_lt_f32 to sgpr; 2: v_add reading the OTHER accumulator set;
// 3: ds_read_b128 fillers
template <int K, int KIND>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (float)i;
  __syncthreads();
  f32x16 acc, other;
  for (int r = 0; r < 16; ++r) { acc[r] = (float)(threadIdx.x + r); other[r] = (float)(r * 3 + threadIdx.x); }
  float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
  float f0 = 1.f, f1 = 2.f, f2 = 3.f, f3 = 4.f;
  float4 l0 = {0, 0, 0, 0};
  const float* lp = lds + (threadIdx.x & 63) * 4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 25; ++s) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      if constexpr (KIND == 0) {
        if (K > 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f0) : "v"(b));
        if (K > 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f1) : "v"(b));
        if (K > 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f2) : "v"(b));
        if (K > 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f3) : "v"(b));
        if (K > 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f0) : "v"(a));
        if (K > 5) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f1) : "v"(a));
        if (K > 6) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f2) : "v"(a));
        if (K > 7) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f3) : "v"(a));
      } else if constexpr (KIND == 2) {
        if (K > 0) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f0) : "v"(other[s & 15]), "v"(b));
        if (K > 1) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f1) : "v"(other[(s + 1) & 15]), "v"(b));
        if (K > 2) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f2) : "v"(other[(s + 2) & 15]), "v"(b));
        if (K > 3) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f3) : "v"(other[(s + 3) & 15]), "v"(b));
      } else if constexpr (KIND == 3) {
        if (K > 0) asm volatile("ds_read_b128 %0, %1" : "=v"(l0) : "v"((unsigned)(size_t)lp));
        if (K > 1) asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(l0) : "v"((unsigned)(size_t)lp));
      }
    }
    if constexpr (KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  float s = f0 + f1 + f2 + f3 + l0.x;
  for (int r = 0; r < 16; ++r) s += acc[r] + other[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int K, int KIND>
void run(int blocks_per_cu, int iters) {
  float* out;
  (void)hipMalloc(&out, sizeof(float) * 256 * 256 * 8);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int grid = 256 * blocks_per_cu;
  probe<K, KIND><<<grid, 256>>>(out, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  probe<K, KIND><<<grid, 256>>>(out, iters);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  double nmfma = (double)grid * 4 * iters * 25;
  double cyc = ms * 1e-3 * 2.4e9 / (nmfma / 1024.0);
  printf("KIND=%d K=%d waves/SIMD=%d: %.2f ms  %.1f TFLOP/s  %.1f cyc/MFMA/SIMD (at 2.4 GHz)\n", KIND, K, blocks_per_cu, ms,
         nmfma * 4096 / ms / 1e9, cyc);
  (void)hipFree(out);
}

int main() {
  const int iters = 10000;
  run<0, 0>(1, iters); run<1, 0>(1, iters); run<2, 0>(1, iters); run<4, 0>(1, iters); run<8, 0>(1, iters);
  run<0, 0>(2, iters); run<2, 0>(2, iters); run<4, 0>(2, iters); run<8, 0>(2, iters);
  run<2, 2>(1, iters); run<4, 2>(1, iters); run<2, 2>(2, iters); run<4, 2>(2, iters);
  run<1, 3>(1, iters); run<2, 3>(1, iters); run<1, 3>(2, iters); run<2, 3>(2, iters);
  return 0;
}
