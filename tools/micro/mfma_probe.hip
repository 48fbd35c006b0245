// v_mfma_f32_32x32x16_f16 on gfx950: fragment layout (asymmetric operands against a CPU product), f16 subnormal inputs (kept or flushed?),
// and how the 16 products of one instruction are accumulated (wider than f32 or f32-rounded per addition).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(const _Float16* A /*[32][16]*/, const _Float16* B /*[16][32]*/, float* D /*[32][32]*/) {
  const int l = threadIdx.x;
  h8 a, b;
  for (int j = 0; j < 8; j++) { a[j] = A[(l & 31) * 16 + 8 * (l >> 5) + j]; b[j] = B[(8 * (l >> 5) + j) * 32 + (l & 31)]; }
  f16v c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; r++) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
int main() {
  std::vector<_Float16> A(32 * 16), B(16 * 32); std::vector<float> D(32 * 32), R(32 * 32);
  _Float16 *dA, *dB; float* dD; hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dD, D.size() * 4);
  auto run = [&]() { hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD); hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost); };
  // 1. layout
  for (int i = 0; i < 32; i++) for (int kk = 0; kk < 16; kk++) A[i * 16 + kk] = (_Float16)((i * 7 + kk * 3) % 11 - 5);
  for (int kk = 0; kk < 16; kk++) for (int j = 0; j < 32; j++) B[kk * 32 + j] = (_Float16)((kk * 5 + j * 13) % 17 - 8);
  run();
  int bad = 0;
  for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) { float s = 0; for (int kk = 0; kk < 16; kk++) s += (float)A[i * 16 + kk] * (float)B[kk * 32 + j]; if (s != D[i * 32 + j]) bad++; }
  printf("layout: %d mismatches of 1024\n", bad);
  // 2. subnormal f16 inputs
  for (auto& x : A) x = 0; for (auto& x : B) x = 0;
  A[0] = (_Float16)9.5367431640625e-07f /* 2^-20, subnormal in f16 */; B[0] = (_Float16)1024.f;
  A[16 + 1] = (_Float16)6.103515625e-05f /* 2^-14, smallest normal */; B[32 + 1] = (_Float16)1024.f;
  run();
  printf("subnormal a = 2^-20 x 1024 -> %.10g (kept: 0.0009765625, flushed: 0);  normal 2^-14 x 1024 -> %.10g\n", D[0], D[32 + 1]);
  // 3. accumulation: 2^24 + 1 - 2^24 inside one instruction (row 0), and across the products order
  for (auto& x : A) x = 0; for (auto& x : B) x = 0;
  A[0] = (_Float16)4096.f; B[0] = (_Float16)4096.f;            // k = 0:  2^24
  A[1] = (_Float16)1.f;    B[32] = (_Float16)1.f;              // k = 1:  1
  A[2] = (_Float16)-4096.f; B[64] = (_Float16)4096.f;          // k = 2: -2^24
  A[16 + 0] = (_Float16)4096.f; B[1] = (_Float16)4096.f;       // row 1 col 1: 2^24 (k=0) + 1 (k=15) - 2^24 (k=8)
  A[16 + 15] = (_Float16)1.f;   B[15 * 32 + 1] = (_Float16)1.f;
  A[16 + 8] = (_Float16)-4096.f; B[8 * 32 + 1] = (_Float16)4096.f;
  run();
  printf("2^24 + 1 - 2^24: k = 0,1,2 -> %.3g ; k = 0,15,8 -> %.3g   (1 = accumulated wider than f32, 0 = f32 rounding per addition)\n", D[0], D[32 + 1]);
  // 4. many small terms against a big one: 2^24 + 15 x 1
  for (auto& x : A) x = 0; for (auto& x : B) x = 0;
  A[0] = (_Float16)4096.f; B[0] = (_Float16)4096.f;
  for (int kk = 1; kk < 16; kk++) { A[kk] = (_Float16)1.f; B[kk * 32] = (_Float16)1.f; }
  run();
  printf("2^24 + 15 x 1 -> %.10g (exact 16777231; f32 round-per-add gives 16777216)\n", D[0]);
  return 0;
}
