// Probe: operand / result lane layout of v_mfma_f32_16x16x32_f16 on gfx950 (round 3, split-precision experiment).
// D[m][n] = sum_k A[m][k] B[k][n]; assumed: lane l holds A[m = l % 16][k = 8 (l / 16) + j], B[k = 8 (l / 16) + j][n = l % 16], j = 0..7,
// and D[4 (l / 16) + i][l % 16], i = 0..3.  Prints the max deviation from a host GEMM under that assumption.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const _Float16* A, const _Float16* B, float* D) {
  const int l = threadIdx.x, r = l % 16, kg = l / 16;
  h8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = A[r * 32 + 8 * kg + j]; b[j] = B[(8 * kg + j) * 16 + r]; }
  f4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) D[(4 * kg + i) * 16 + r] = c[i];
}
int main() {
  std::vector<_Float16> A(16 * 32), B(32 * 16);
  for (int i = 0; i < 16 * 32; ++i) { A[i] = (_Float16)(((i * 37) % 17 - 8) / 8.0f); B[i] = (_Float16)(((i * 53) % 19 - 9) / 16.0f); }
  _Float16 *dA, *dB; float* dD;
  hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 1024);
  hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  std::vector<float> D(256);
  hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
    double s = 0; for (int kk = 0; kk < 32; ++kk) s += (double)(float)A[m * 32 + kk] * (double)(float)B[kk * 16 + n];
    worst = fmax(worst, fabs(s - D[m * 16 + n]));
  }
  printf("mfma_f32_16x16x32_f16 layout check: max deviation %.3e (%s)\n", worst, worst < 1e-3 ? "layout OK" : "layout WRONG");
  return 0;
}
