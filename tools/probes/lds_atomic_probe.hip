// LDS atomic rate probe (gfx950): ds_add_u32 vs ds_add_u64 vs ds_add_f64 on scattered addresses of a 128 KB table,
// 1024 threads per workgroup, one workgroup per CU.  hipcc --offload-arch=gfx950 -O3 lds_atomic_probe.hip -o lds_atomic_probe.out
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(1024) void probe(int iters, unsigned long long* out) {
  extern __shared__ unsigned long long tab[];  // 16384 x 8 B
  for (int i = threadIdx.x; i < 16384; i += 1024) tab[i] = 0;
  __syncthreads();
  unsigned int h = threadIdx.x * 2654435761u + blockIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      h = h * 1664525u + 1013904223u;
      const unsigned int slot = (h >> 10) & 16383u;
      if (MODE == 0) atomicAdd(reinterpret_cast<unsigned int*>(tab) + slot, 1u);
      else if (MODE == 1) atomicAdd(tab + slot, 1ull);
      else atomicAdd(reinterpret_cast<double*>(tab) + slot, 1.0);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = tab[0];
}
int main() {
  unsigned long long* out;
  hipMalloc(&out, 8 * 1024);
  const int iters = 2000;
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEvent_t a, b;
      hipEventCreate(&a);
      hipEventCreate(&b);
      hipEventRecord(a);
      if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(1024), 131072, 0, iters, out);
      if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(1024), 131072, 0, iters, out);
      if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(256), dim3(1024), 131072, 0, iters, out);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      const double lanes = 256.0 * 1024 * iters * 8;
      if (rep) printf("mode %d (%s): %.3f ms, %.2f atomic lanes per CU per ns, %.3e lanes/s\n", mode,
                      mode == 0 ? "ds_add_u32" : (mode == 1 ? "ds_add_u64" : "ds_add_f64"), ms, lanes / 256 / (ms * 1e6), lanes / (ms * 1e-3));
    }
  }
  return 0;
}
