// Micro-benchmark 2: SIMD time taken away from an fp32 MFMA stream by other instruction types, issued by a SIDE wave on the
// same SIMD or by the MFMA wave itself.  hipcc -O3 --offload-arch=gfx950 tools/probes/coissue2.hip -o tools/probes/bin/coissue2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ float4 gsrc[64 * 64];

// side kinds: 0 none, 1 32x ds_read_b32 (one wait at the end), 2 32x ds_read_b128, 3 16x ds_write_b128, 4 10x LDS-DMA 1 KiB pieces,
// 5 64 v_fma, 6 32 v_pk_fma_f32, 7 32x ds_read_b64
// own kinds (MFMA waves themselves, per 27 MFMAs): 0 none, 1 12x ds_read_b128 up front, 2 12x ds_read_b128 interleaved, 3 36 ds_read_b32 interleaved
template <int SIDE, int OWN>
__global__ void __launch_bounds__(768) probe(float* out, int iters, int nm, int ns, const float4* gs) {
  extern __shared__ float4 lds4[];
  float* lds = (float*)lds4;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += 768) lds4[i] = make_float4(i, 1, 2, 3);
  __syncthreads();
  const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) float4*)lds4;
  if (wave < 8) {
    if (wave >= nm) return;
    f32x4 acc[27];
    for (int i = 0; i < 27; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = (float)lane, b = 1.0f / (1 + lane);
    f32x4 r[12];
    float q[36];
    for (int i = 0; i < 12; ++i) r[i] = (f32x4){0, 0, 0, 0};
    for (int i = 0; i < 36; ++i) q[i] = 0.f;
    const unsigned addr = lbase + lane * 16 + wave * 1024;
    for (int it = 0; it < iters; ++it) {
      if (OWN == 1) {
#pragma unroll
        for (int i = 0; i < 12; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[i]) : "v"(addr), "n"(0) );
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
#pragma unroll
      for (int i = 0; i < 27; ++i) {
        if (OWN == 2 && i < 24 && (i & 1) == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(r[i >> 1]) : "v"(addr));
        if (OWN == 3) asm volatile("ds_read_b32 %0, %1" : "=v"(q[i]) : "v"(addr));
        if (OWN == 3 && i < 9) asm volatile("ds_read_b32 %0, %1" : "=v"(q[27 + i]) : "v"(addr));
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
      }
      if (OWN >= 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = 0;
    for (int i = 0; i < 12; ++i) s += r[i][0];
    for (int i = 0; i < 36; ++i) s += q[i];
    for (int i = 0; i < 27; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 768 + threadIdx.x] = s;
  } else {
    if (wave - 8 >= ns) return;
    f32x4 r[8];
    float x[8];
    for (int i = 0; i < 8; ++i) { r[i] = (f32x4){(float)lane, 1, 2, 3}; x[i] = lane + i; }
    const unsigned addr = lbase + lane * 16 + (wave - 8) * 1024;
    const unsigned addr4 = lbase + lane * 4 + (wave - 8) * 1024;
    float a = 1.0001f, b = 0.5f;
    f32x2 pa = (f32x2){1.0001f, 1.0002f}, pb = (f32x2){0.5f, 0.25f};
    f32x2 px[8];
    for (int i = 0; i < 8; ++i) px[i] = (f32x2){(float)lane, (float)i};
    for (int it = 0; it < iters; ++it) {
      if (SIDE == 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("ds_read_b32 %0, %1" : "=v"(x[i]) : "v"(addr4));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      } else if (SIDE == 2) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(r[i]) : "v"(addr));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      } else if (SIDE == 7) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("ds_read_b64 %0, %1" : "=v"(px[i]) : "v"(addr));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      } else if (SIDE == 3) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(r[i]) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      } else if (SIDE == 4) {
#pragma unroll
        for (int k = 0; k < 10; ++k) {
          const void* src = gs + k * 64;
          unsigned keep;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"((unsigned)lane * 16u), "s"(src), "s"((unsigned)__builtin_amdgcn_readfirstlane((int)(lbase + 32768u + (wave - 8) * 10240u + k * 1024u))) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else if (SIDE == 8 || SIDE == 9) {      // 10 global_load_dwordx4 (1 KiB per wave instruction) into VGPRs (+ 10 ds_write_b128 of last round's data)
        f32x4 g[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) {
          const float4* src = gs + k * 64;
          asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(g[k]) : "v"((unsigned)lane * 16u), "s"(src) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (SIDE == 9) {
#pragma unroll
          for (int k = 0; k < 10; ++k) asm volatile("ds_write_b128 %0, %1" :: "v"(addr + 16384u * 0 + k * 0), "v"(g[k]) : "memory");
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
#pragma unroll
          for (int k = 0; k < 10; ++k) r[k & 7] += g[k];
        }
      } else if (SIDE == 5) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
      } else if (SIDE == 6) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(px[i]) : "v"(pa), "v"(pb));
      }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i] + r[i][0] + px[i][0] + px[i][1];
    out[blockIdx.x * 768 + threadIdx.x] = s;
  }
}

static float4* gptr;
template <int SIDE, int OWN>
float run(float* out, int iters, int nm, int ns) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  (void)hipFuncSetAttribute((const void*)probe<SIDE, OWN>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  probe<SIDE, OWN><<<256, 768, 100 * 1024>>>(out, 10, nm, ns, gptr);
  (void)hipEventRecord(e0);
  probe<SIDE, OWN><<<256, 768, 100 * 1024>>>(out, iters, nm, ns, gptr);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6f / iters;   // ns per iteration
}

int main() {
  float* out; (void)hipMalloc(&out, 256 * 768 * 4);
  (void)hipMalloc(&gptr, 64 * 64 * 16); (void)hipMemset(gptr, 0, 64 * 64 * 16);
  const int it = 2000;
  const float base = run<0, 0>(out, it, 8, 0);
  printf("8 MFMA waves (2/SIMD), 27 MFMAs per wave and iteration: %.0f ns/iter (= %.1f clk per MFMA at 2.19 GHz)\n", base, base * 2.19 / 54);
  printf("side wave (1 per SIMD) instruction streams, alone | next to the MFMA waves | extra ns | extra clk per side instruction:\n");
#define SIDE_CASE(K, N, NAME) { float a = run<K, 0>(out, it, 0, 4), b = run<K, 0>(out, it, 8, 4); \
    printf("  %-34s %6.0f | %6.0f | %+6.0f | %5.1f\n", NAME, a, b, b - base, (b - base) * 2.19 / N); }
  SIDE_CASE(1, 32, "32 ds_read_b32");
  SIDE_CASE(7, 32, "32 ds_read_b64");
  SIDE_CASE(2, 32, "32 ds_read_b128");
  SIDE_CASE(3, 16, "16 ds_write_b128");
  SIDE_CASE(4, 10, "10 LDS-DMA pieces (1 KiB)");
  SIDE_CASE(8, 10, "10 global_load_dwordx4 -> VGPR (+10 v_pk_add)");
  SIDE_CASE(9, 10, "10 global_load_dwordx4 + 10 ds_write_b128");
  SIDE_CASE(5, 64, "64 v_fma_f32");
  SIDE_CASE(6, 32, "32 v_pk_fma_f32");
  printf("the MFMA waves' own LDS reads per 27 MFMAs: ns/iter | extra | extra clk per read (per SIMD: 2 waves)\n");
#define OWN_CASE(K, N, NAME) { float b = run<0, K>(out, it, 8, 0); printf("  %-34s %6.0f | %+6.0f | %5.1f\n", NAME, b, b - base, (b - base) * 2.19 / (2 * N)); }
  OWN_CASE(1, 12, "12 ds_read_b128 up front + wait");
  OWN_CASE(2, 12, "12 ds_read_b128 interleaved");
  OWN_CASE(3, 36, "36 ds_read_b32 interleaved");
  return 0;
}
