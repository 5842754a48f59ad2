// What does one host round trip of the Leiden / kNN orchestration cost?  The pattern in the library: a small kernel, one or two
// hipMemcpyAsync(device -> PAGEABLE host, a few ints), hipStreamSynchronize.  Variants: the same into PINNED host memory;
// no copy at all (the kernel writes into mapped pinned memory, the host only synchronises).
//   hipcc -O2 --offload-arch=gfx950 tools/probes/host_roundtrip_probe.hip -o /tmp/rt && /tmp/rt
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

__global__ void bump(int* c, int n) {
  if (threadIdx.x < n) c[threadIdx.x] += 1;
}
__global__ void bump_and_publish(int* c, int n, volatile int* host) {
  if (threadIdx.x < n) {
    c[threadIdx.x] += 1;
    host[threadIdx.x] = c[threadIdx.x];
  }
}

static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  hipStream_t s;
  hipStreamCreate(&s);
  int *d, *pin, *pin_dev;
  hipMalloc(&d, 4096);
  hipMemset(d, 0, 4096);
  hipHostMalloc(&pin, 4096, hipHostMallocMapped);
  hipHostGetDevicePointer(reinterpret_cast<void**>(&pin_dev), pin, 0);
  const int reps = 2000;
  int pageable[64];
  for (int variant = 0; variant < 5; ++variant) {
    for (int warm = 0; warm < 2; ++warm) {
      const double t0 = now();
      for (int i = 0; i < reps; ++i) {
        if (variant == 4) hipLaunchKernelGGL(bump_and_publish, dim3(1), dim3(64), 0, s, d, 56, pin_dev);
        else hipLaunchKernelGGL(bump, dim3(1), dim3(64), 0, s, d, 56);
        if (variant == 0) {  // the library today: two copies into pageable memory, then the synchronisation
          hipMemcpyAsync(pageable, d, 48 * 4, hipMemcpyDeviceToHost, s);
          hipMemcpyAsync(pageable + 48, d + 48, 8 * 4, hipMemcpyDeviceToHost, s);
        } else if (variant == 1) {  // one copy into pageable memory
          hipMemcpyAsync(pageable, d, 56 * 4, hipMemcpyDeviceToHost, s);
        } else if (variant == 2) {  // two copies into pinned memory
          hipMemcpyAsync(pin, d, 48 * 4, hipMemcpyDeviceToHost, s);
          hipMemcpyAsync(pin + 48, d + 48, 8 * 4, hipMemcpyDeviceToHost, s);
        } else if (variant == 3) {  // one copy into pinned memory
          hipMemcpyAsync(pin, d, 56 * 4, hipMemcpyDeviceToHost, s);
        }
        hipStreamSynchronize(s);
      }
      const double t1 = now();
      static const char* names[] = {"2 copies -> pageable + sync", "1 copy -> pageable + sync", "2 copies -> pinned + sync",
                                    "1 copy -> pinned + sync", "kernel writes mapped pinned memory + sync"};
      if (warm) printf("%-44s %7.2f us per round trip (value %d)\n", names[variant], (t1 - t0) / reps, variant == 4 ? pin[0] : (variant >= 2 ? pin[0] : pageable[0]));
    }
  }
  // launch only (no synchronisation): the floor of a launch
  {
    const double t0 = now();
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(bump, dim3(1), dim3(64), 0, s, d, 56);
    hipStreamSynchronize(s);
    printf("%-44s %7.2f us per launch\n", "back-to-back launches, one sync at the end", (now() - t0) / reps);
  }
  return 0;
}
