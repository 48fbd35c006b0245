// micro-benchmark: what makes a dependent kernel boundary expensive?  196 x 512-thread blocks, variants: static LDS, scratch, big kernarg block,
// a cross-kernel data dependency (each launch reads what the previous one wrote, like the tick's partial rows)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { float* p; int n; double pad[46]; };
__global__ void k_plain(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }
__global__ void k_lds(float* p, int n) { __shared__ float sh[12000]; int i = blockIdx.x * blockDim.x + threadIdx.x; sh[threadIdx.x * 23 % 12000] = (float)i; __syncthreads(); if (i < n) p[i] += sh[(threadIdx.x * 7) % 12000] * 0.f + 1.f; }
__global__ void k_scratch(float* p, int n) { float loc[64]; int i = blockIdx.x * blockDim.x + threadIdx.x;
  for (int k = 0; k < 64; k++) loc[k] = (float)(k + i); float v = loc[(i * 7 + n) & 63] + loc[(i * 3 + n) & 63]; if (i < n) p[i] += v * 0.f + 1.f; }
__global__ void k_big(Big b) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < b.n) b.p[i] += 1.f + (float)b.pad[i & 31] * 0.f; }
// every block reads all 196 rows of 28 doubles the previous launch wrote, then writes its own row
__global__ void k_rows(const double* in, double* out) { __shared__ double acc[512]; double v = 0; for (int r = threadIdx.x; r < 196 * 28; r += 512) v += in[r]; acc[threadIdx.x] = v; __syncthreads();
  if (threadIdx.x < 28) out[blockIdx.x * 28 + threadIdx.x] = acc[threadIdx.x] * 1e-9 + 1.0; }
int main() {
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  float* p; hipMalloc(&p, 1 << 22); hipMemset(p, 0, 1 << 22);
  double* rows[2]; for (int i = 0; i < 2; i++) { hipMalloc(&rows[i], 196 * 28 * 8); hipMemset(rows[i], 0, 196 * 28 * 8); }
  Big b{}; b.p = p; b.n = 100000;
  const char* names[] = {"plain", "48 KB LDS", "scratch 256 B/lane", "384 B kernarg", "rows dependency"};
  for (int mode = 0; mode < 5; mode++) for (int rep = 0; rep < 2; rep++) {
    const int N = 2000;
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; i++) {
      switch (mode) {
        case 0: hipLaunchKernelGGL(k_plain, dim3(196), dim3(512), 0, s, p, 100000); break;
        case 1: hipLaunchKernelGGL(k_lds, dim3(196), dim3(512), 0, s, p, 100000); break;
        case 2: hipLaunchKernelGGL(k_scratch, dim3(196), dim3(512), 0, s, p, 100000); break;
        case 3: hipLaunchKernelGGL(k_big, dim3(196), dim3(512), 0, s, b); break;
        case 4: hipLaunchKernelGGL(k_rows, dim3(196), dim3(512), 0, s, rows[i & 1], rows[(i + 1) & 1]); break;
      }
    }
    auto t1 = std::chrono::steady_clock::now();
    hipStreamSynchronize(s);
    auto t2 = std::chrono::steady_clock::now();
    printf("%-20s enqueue %.2f us/launch, total %.2f us/launch\n", names[mode], std::chrono::duration<double, std::micro>(t1 - t0).count() / N, std::chrono::duration<double, std::micro>(t2 - t0).count() / N);
  }
  return 0;
}
