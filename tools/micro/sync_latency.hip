// Developer micro-benchmark: what one host look costs.  (a) launch + hipStreamSynchronize round trip of an empty kernel; (b) the same with the host polling a sequence number the
// kernel writes to page-locked host memory (what the persistent kernel's status mirror does); (c) with a 4 KB hipMemcpyAsync in front of the launch (the argument arena upload).
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/sync_latency.hip -o tools/micro/sync_latency.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
__global__ void k_seq(volatile uint32_t* flag, uint32_t v) { if (threadIdx.x == 0) { __threadfence_system(); *flag = v; } }
int main() {
  hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  uint32_t* flag; hipHostMalloc((void**)&flag, 64, hipHostMallocDefault); *flag = 0;
  char *hsrc, *ddst; hipHostMalloc((void**)&hsrc, 4096, hipHostMallocDefault); hipMalloc((void**)&ddst, 4096);
  const int N = 2000;
  for (int mode = 0; mode < 4; mode++) {
    for (int w = 0; w < 50; w++) { hipLaunchKernelGGL(k_seq, dim3(1), dim3(64), 0, s, flag, 0u); hipStreamSynchronize(s); }
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 1; i <= N; i++) {
      if (mode >= 2) hipMemcpyAsync(ddst, hsrc, 4096, hipMemcpyHostToDevice, s);
      hipLaunchKernelGGL(k_seq, dim3(1), dim3(64), 0, s, flag, (uint32_t)(i + mode * 100000));
      if (mode == 0 || mode == 2) hipStreamSynchronize(s);
      else { while (*(volatile uint32_t*)flag != (uint32_t)(i + mode * 100000)) __builtin_ia32_pause(); }
    }
    hipStreamSynchronize(s);
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
    printf("SYNC mode %d (%s%s): %.2f us per round trip\n", mode, mode >= 2 ? "4 KB H2D + " : "", (mode & 1) ? "launch + poll pinned flag" : "launch + hipStreamSynchronize", us);
  }
  return 0;
}
