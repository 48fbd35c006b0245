// micro-benchmark: cost of a dependent kernel boundary on one stream (empty kernels / 196 x 512-thread blocks touching memory)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_empty() {}
__global__ void k_touch(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }
int main() {
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  float* p; hipMalloc(&p, 1 << 22); hipMemset(p, 0, 1 << 22);
  for (int mode = 0; mode < 2; mode++) for (int rep = 0; rep < 3; rep++) {
    const int N = 2000;
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; i++) { if (mode == 0) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); else hipLaunchKernelGGL(k_touch, dim3(196), dim3(512), 0, s, p, 100000); }
    auto t1 = std::chrono::steady_clock::now();
    hipStreamSynchronize(s);
    auto t2 = std::chrono::steady_clock::now();
    printf("mode %d: enqueue %.2f us/launch, total %.2f us/launch\n", mode, std::chrono::duration<double, std::micro>(t1 - t0).count() / N, std::chrono::duration<double, std::micro>(t2 - t0).count() / N);
  }
  for (int rep = 0; rep < 3; rep++) {
    const int N = 500;
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; i++) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); hipStreamSynchronize(s); }
    auto t1 = std::chrono::steady_clock::now();
    printf("launch + hipStreamSynchronize: %.2f us\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
  }
  for (int rep = 0; rep < 3; rep++) {
    const int N = 500;
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; i++) { hipMemsetAsync(p, 0, 1600000, s); hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); }
    hipStreamSynchronize(s);
    auto t1 = std::chrono::steady_clock::now();
    printf("memsetAsync(1.6 MB) + kernel: %.2f us\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
  }
  return 0;
}
