// Micro-benchmark: do VALU / SALU / LDS instructions of one wave issue while OTHER waves of the same SIMD stream fp32 MFMAs?
// (design question behind conv_wino4p.hip).  Build: hipcc -O3 --offload-arch=gfx950 tools/probes/coissue.hip -o gpurun_out/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// mode bits: 1 = MFMA waves active (waves 0..nm-1), 2 = side waves active (waves 8..8+ns-1); side work type in `kind`:
// 0 = 64 independent v_fma per iteration, 1 = 64 s_add per iteration, 2 = 32 ds_read_b32 per iteration, 3 = mixed own-wave: MFMA waves
// themselves interleave 3 v_fma after every MFMA (no side waves)
template <int KIND>
__global__ void __launch_bounds__(768) probe(float* out, int iters, int mode, int nm, int ns) {
  extern __shared__ float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  lds[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  if (wave < 8) {
    if (!(mode & 1) || wave >= nm) return;
    f32x4 acc[27];
    for (int i = 0; i < 27; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = (float)lane, b = 1.0f / (1 + lane);
    float x0 = a, x1 = b, x2 = a + b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 27; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        if (KIND == 3) {
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x1) : "v"(a), "v"(b));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x2) : "v"(a), "v"(b));
        }
      }
    }
    float s = x0 + x1 + x2;
    for (int i = 0; i < 27; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 768 + threadIdx.x] = s;
  } else {
    if (!(mode & 2) || wave - 8 >= ns) return;
    __builtin_amdgcn_s_setprio(3);
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = (float)(lane + i);
    float a = 1.0001f, b = 0.5f;
    int sacc = 0;
    for (int it = 0; it < iters; ++it) {
      if (KIND == 0) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
      } else if (KIND == 1) {
#pragma unroll
        for (int r = 0; r < 64; ++r) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
      } else if (KIND == 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int i = 0; i < 8; ++i) x[i] += lds[(lane + 64 * i + r * 7 + it) & 767];
        }
      }
    }
    float s = (float)sacc;
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 768 + threadIdx.x] = s;
  }
}

template <int KIND>
float run(float* out, int iters, int mode, int nm, int ns) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)probe<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  probe<KIND><<<256, 768, 100 * 1024>>>(out, 10, mode, nm, ns);
  hipEventRecord(e0);
  probe<KIND><<<256, 768, 100 * 1024>>>(out, iters, mode, nm, ns);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f;
}

int main() {
  float* out; hipMalloc(&out, 256 * 768 * 4);
  const int iters = 2000;
  const char* names[] = {"side = 64 v_fma/iter", "side = 64 s_add/iter", "side = 32 ds_read_b32/iter", "own-wave: 3 v_fma after every MFMA"};
  printf("per iteration an MFMA wave issues 27 v_mfma_f32_16x16x4_f32 (ideal 27*32 = 864 clk alone, 2 waves/SIMD: 1728 clk)\n");
  for (int nm : {4, 8}) {
    float m = run<0>(out, iters, 1, nm, 0);
    printf("MFMA waves only, %d waves (%d per SIMD): %.1f us  = %.0f ns/iter\n", nm, nm / 4, m, m * 1e3 / iters);
  }
  { float m = run<3>(out, iters, 1, 8, 0); printf("%s, 8 waves: %.1f us = %.0f ns/iter\n", names[3], m, m * 1e3 / iters); }
  { float m = run<3>(out, iters, 1, 4, 0); printf("%s, 4 waves: %.1f us = %.0f ns/iter\n", names[3], m, m * 1e3 / iters); }
  float s0 = run<0>(out, iters, 2, 0, 4), s1 = run<1>(out, iters, 2, 0, 4), s2 = run<2>(out, iters, 2, 0, 4);
  printf("side waves only (4 waves): %s %.0f ns/iter | %s %.0f | %s %.0f\n", names[0], s0 * 1e3 / iters, names[1], s1 * 1e3 / iters, names[2], s2 * 1e3 / iters);
  for (int nm : {4, 8}) {
    float b0 = run<0>(out, iters, 3, nm, 4), b1 = run<1>(out, iters, 3, nm, 4), b2 = run<2>(out, iters, 3, nm, 4);
    printf("%d MFMA waves + 4 side waves: %s %.0f ns/iter | %s %.0f | %s %.0f\n", nm, names[0], b0 * 1e3 / iters, names[1], b1 * 1e3 / iters, names[2], b2 * 1e3 / iters);
  }
  return 0;
}
