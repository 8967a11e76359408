// Pieces shared by the two Winograd F(4x4,3x3) kernels (conv_wino4.hip = ALG 7, conv_wino4p.hip = ALG 8):
// transform matrices [Lavin & Gray 2016], slab / patch geometry, the LDS-DMA primitive, U = G g G^T on the host.
#pragma once
#include "conv_mfma_types.h"
#include <vector>

namespace w4 {

// row R of B^T applied to six values (input transform)
template <int R>
__device__ __forceinline__ float bt_row(float a, float b, float c, float d, float e, float f) {
  if constexpr (R == 0) return 4.f * a - 5.f * c + e;
  else if constexpr (R == 1) return -4.f * (b + c) + d + e;
  else if constexpr (R == 2) return 4.f * (b - c) - d + e;
  else if constexpr (R == 3) return 2.f * (d - b) - c + e;
  else if constexpr (R == 4) return 2.f * (b - d) - c + e;
  else return 4.f * b - 5.f * d + f;
}
// A^T[i][k] (output transform)
__host__ __device__ constexpr float at_c(int i, int k) {
  constexpr float A[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 0}, {0, 1, 1, 4, 4, 0}, {0, 1, -1, 8, -8, 1}};
  return A[i][k];
}

// 64 lanes x 16 B from global memory straight into LDS at (wave-uniform) byte address lds_dst + lane * 16.
// Inline asm so that hipcc neither tracks it in its waitcnt insertion nor drains it at every barrier.
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

// Same, wave-uniform base (SGPR pair) + per-lane 32-bit byte offset: no 64-bit VALU address arithmetic.
__device__ __forceinline__ void dma16_sv(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}

// N consecutive one-KiB pieces of one contiguous stream -> LDS, in ONE asm block: source = wave-uniform base + per-lane offsets
// voff[k] (lane * 16 + k * step bytes, loop-invariant VGPRs), LDS destination m0 = dst0 + k * step.  Per piece that is
// s_add_u32 m0 / s_nop / global_load_lds_dwordx4 instead of the eight instructions of a dma16_sv call with its own scalar address
// arithmetic and m0 save / restore (an LDS-DMA request cost its issuing wave ~100 clk in the first traces of this kernel).
#define W4_DMA1 "s_add_u32 m0, m0, %[st]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 "
template <int N>
__device__ __forceinline__ void dma_stream(const void* sbase, const unsigned (&voff)[7], unsigned dst0, unsigned step) {
  static_assert(N >= 1 && N <= 7, "1..7 pieces per call");
  unsigned keep;
  if constexpr (N == 1)
    asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[d]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v0], %[b]\n\ts_mov_b32 m0, %[k]"
                 : [k] "=&s"(keep) : [d] "s"(dst0), [st] "s"(step), [b] "s"(sbase), [v0] "v"(voff[0]) : "memory", "scc");
  else if constexpr (N == 2)
    asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[d]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v0], %[b]\n\t" W4_DMA1 "%[v1], %[b]\n\ts_mov_b32 m0, %[k]"
                 : [k] "=&s"(keep) : [d] "s"(dst0), [st] "s"(step), [b] "s"(sbase), [v0] "v"(voff[0]), [v1] "v"(voff[1]) : "memory", "scc");
  else if constexpr (N == 3)
    asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[d]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v0], %[b]\n\t" W4_DMA1 "%[v1], %[b]\n\t" W4_DMA1 "%[v2], %[b]\n\ts_mov_b32 m0, %[k]"
                 : [k] "=&s"(keep) : [d] "s"(dst0), [st] "s"(step), [b] "s"(sbase), [v0] "v"(voff[0]), [v1] "v"(voff[1]), [v2] "v"(voff[2]) : "memory", "scc");
  else if constexpr (N == 4)
    asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[d]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v0], %[b]\n\t" W4_DMA1 "%[v1], %[b]\n\t" W4_DMA1 "%[v2], %[b]\n\t" W4_DMA1
                 "%[v3], %[b]\n\ts_mov_b32 m0, %[k]"
                 : [k] "=&s"(keep) : [d] "s"(dst0), [st] "s"(step), [b] "s"(sbase), [v0] "v"(voff[0]), [v1] "v"(voff[1]), [v2] "v"(voff[2]), [v3] "v"(voff[3]) : "memory", "scc");
  else if constexpr (N == 5)
    asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[d]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v0], %[b]\n\t" W4_DMA1 "%[v1], %[b]\n\t" W4_DMA1 "%[v2], %[b]\n\t" W4_DMA1
                 "%[v3], %[b]\n\t" W4_DMA1 "%[v4], %[b]\n\ts_mov_b32 m0, %[k]"
                 : [k] "=&s"(keep) : [d] "s"(dst0), [st] "s"(step), [b] "s"(sbase), [v0] "v"(voff[0]), [v1] "v"(voff[1]), [v2] "v"(voff[2]), [v3] "v"(voff[3]), [v4] "v"(voff[4]) : "memory", "scc");
  else if constexpr (N == 6)
    asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[d]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v0], %[b]\n\t" W4_DMA1 "%[v1], %[b]\n\t" W4_DMA1 "%[v2], %[b]\n\t" W4_DMA1
                 "%[v3], %[b]\n\t" W4_DMA1 "%[v4], %[b]\n\t" W4_DMA1 "%[v5], %[b]\n\ts_mov_b32 m0, %[k]"
                 : [k] "=&s"(keep) : [d] "s"(dst0), [st] "s"(step), [b] "s"(sbase), [v0] "v"(voff[0]), [v1] "v"(voff[1]), [v2] "v"(voff[2]), [v3] "v"(voff[3]), [v4] "v"(voff[4]), [v5] "v"(voff[5]) : "memory", "scc");
  else
    asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[d]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v0], %[b]\n\t" W4_DMA1 "%[v1], %[b]\n\t" W4_DMA1 "%[v2], %[b]\n\t" W4_DMA1
                 "%[v3], %[b]\n\t" W4_DMA1 "%[v4], %[b]\n\t" W4_DMA1 "%[v5], %[b]\n\t" W4_DMA1 "%[v6], %[b]\n\ts_mov_b32 m0, %[k]"
                 : [k] "=&s"(keep) : [d] "s"(dst0), [st] "s"(step), [b] "s"(sbase), [v0] "v"(voff[0]), [v1] "v"(voff[1]), [v2] "v"(voff[2]), [v3] "v"(voff[3]), [v4] "v"(voff[4]), [v5] "v"(voff[5]), [v6] "v"(voff[6]) : "memory", "scc");
}
#undef W4_DMA1

// N one-KiB GATHER pieces (per-lane byte offsets voff[k] from one wave-uniform base; lanes outside mask[k] - padding positions of a
// patch - request nothing and leave their LDS slot alone) -> LDS destinations dst0 + k * step, in ONE asm block: per piece
// s_add_u32 m0 / s_mov_b64 exec / s_nop / global_load_lds_dwordx4 instead of a dma16_sv call inside a compiler-made exec branch.
// A piece whose mask is empty still issues (a no-op that counts in vmcnt): the caller counts N requests.
template <int N>
__device__ __forceinline__ void dma_gather(const void* sbase, const unsigned (&voff)[8], const unsigned long long (&mask)[8], unsigned dst0,
                                           unsigned step) {
  static_assert(N >= 1 && N <= 8, "1..8 pieces per call");
  unsigned keep;
  unsigned long long sexec;
  if constexpr (N == 1)
    asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b64 %[se], exec\n\ts_mov_b32 m0, %[d]\n\t"
                 "s_mov_b64 exec, %[m0]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v0], %[b]\n\t"
                 "s_mov_b64 exec, %[se]\n\ts_mov_b32 m0, %[k]"
                 : [k] "=&s"(keep), [se] "=&s"(sexec) : [d] "s"(dst0), [st] "s"(step), [b] "s"(sbase), [m0] "s"(mask[0]), [v0] "v"(voff[0]) : "memory", "scc");
  else if constexpr (N == 2)
    asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b64 %[se], exec\n\ts_mov_b32 m0, %[d]\n\t"
                 "s_mov_b64 exec, %[m0]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v0], %[b]\n\t"
                 "s_add_u32 m0, m0, %[st]\n\t"
                 "s_mov_b64 exec, %[m1]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v1], %[b]\n\t"
                 "s_mov_b64 exec, %[se]\n\ts_mov_b32 m0, %[k]"
                 : [k] "=&s"(keep), [se] "=&s"(sexec) : [d] "s"(dst0), [st] "s"(step), [b] "s"(sbase), [m0] "s"(mask[0]), [v0] "v"(voff[0]), [m1] "s"(mask[1]), [v1] "v"(voff[1]) : "memory", "scc");
  else if constexpr (N == 3)
    asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b64 %[se], exec\n\ts_mov_b32 m0, %[d]\n\t"
                 "s_mov_b64 exec, %[m0]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v0], %[b]\n\t"
                 "s_add_u32 m0, m0, %[st]\n\t"
                 "s_mov_b64 exec, %[m1]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v1], %[b]\n\t"
                 "s_add_u32 m0, m0, %[st]\n\t"
                 "s_mov_b64 exec, %[m2]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v2], %[b]\n\t"
                 "s_mov_b64 exec, %[se]\n\ts_mov_b32 m0, %[k]"
                 : [k] "=&s"(keep), [se] "=&s"(sexec) : [d] "s"(dst0), [st] "s"(step), [b] "s"(sbase), [m0] "s"(mask[0]), [v0] "v"(voff[0]), [m1] "s"(mask[1]), [v1] "v"(voff[1]), [m2] "s"(mask[2]), [v2] "v"(voff[2]) : "memory", "scc");
  else if constexpr (N == 4)
    asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b64 %[se], exec\n\ts_mov_b32 m0, %[d]\n\t"
                 "s_mov_b64 exec, %[m0]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v0], %[b]\n\t"
                 "s_add_u32 m0, m0, %[st]\n\t"
                 "s_mov_b64 exec, %[m1]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v1], %[b]\n\t"
                 "s_add_u32 m0, m0, %[st]\n\t"
                 "s_mov_b64 exec, %[m2]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v2], %[b]\n\t"
                 "s_add_u32 m0, m0, %[st]\n\t"
                 "s_mov_b64 exec, %[m3]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v3], %[b]\n\t"
                 "s_mov_b64 exec, %[se]\n\ts_mov_b32 m0, %[k]"
                 : [k] "=&s"(keep), [se] "=&s"(sexec) : [d] "s"(dst0), [st] "s"(step), [b] "s"(sbase), [m0] "s"(mask[0]), [v0] "v"(voff[0]), [m1] "s"(mask[1]), [v1] "v"(voff[1]), [m2] "s"(mask[2]), [v2] "v"(voff[2]), [m3] "s"(mask[3]), [v3] "v"(voff[3]) : "memory", "scc");
  else if constexpr (N == 5)
    asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b64 %[se], exec\n\ts_mov_b32 m0, %[d]\n\t"
                 "s_mov_b64 exec, %[m0]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v0], %[b]\n\t"
                 "s_add_u32 m0, m0, %[st]\n\t"
                 "s_mov_b64 exec, %[m1]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v1], %[b]\n\t"
                 "s_add_u32 m0, m0, %[st]\n\t"
                 "s_mov_b64 exec, %[m2]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v2], %[b]\n\t"
                 "s_add_u32 m0, m0, %[st]\n\t"
                 "s_mov_b64 exec, %[m3]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v3], %[b]\n\t"
                 "s_add_u32 m0, m0, %[st]\n\t"
                 "s_mov_b64 exec, %[m4]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v4], %[b]\n\t"
                 "s_mov_b64 exec, %[se]\n\ts_mov_b32 m0, %[k]"
                 : [k] "=&s"(keep), [se] "=&s"(sexec) : [d] "s"(dst0), [st] "s"(step), [b] "s"(sbase), [m0] "s"(mask[0]), [v0] "v"(voff[0]), [m1] "s"(mask[1]), [v1] "v"(voff[1]), [m2] "s"(mask[2]), [v2] "v"(voff[2]), [m3] "s"(mask[3]), [v3] "v"(voff[3]), [m4] "s"(mask[4]), [v4] "v"(voff[4]) : "memory", "scc");
  else if constexpr (N == 6)
    asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b64 %[se], exec\n\ts_mov_b32 m0, %[d]\n\t"
                 "s_mov_b64 exec, %[m0]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v0], %[b]\n\t"
                 "s_add_u32 m0, m0, %[st]\n\t"
                 "s_mov_b64 exec, %[m1]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v1], %[b]\n\t"
                 "s_add_u32 m0, m0, %[st]\n\t"
                 "s_mov_b64 exec, %[m2]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v2], %[b]\n\t"
                 "s_add_u32 m0, m0, %[st]\n\t"
                 "s_mov_b64 exec, %[m3]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v3], %[b]\n\t"
                 "s_add_u32 m0, m0, %[st]\n\t"
                 "s_mov_b64 exec, %[m4]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v4], %[b]\n\t"
                 "s_add_u32 m0, m0, %[st]\n\t"
                 "s_mov_b64 exec, %[m5]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v5], %[b]\n\t"
                 "s_mov_b64 exec, %[se]\n\ts_mov_b32 m0, %[k]"
                 : [k] "=&s"(keep), [se] "=&s"(sexec) : [d] "s"(dst0), [st] "s"(step), [b] "s"(sbase), [m0] "s"(mask[0]), [v0] "v"(voff[0]), [m1] "s"(mask[1]), [v1] "v"(voff[1]), [m2] "s"(mask[2]), [v2] "v"(voff[2]), [m3] "s"(mask[3]), [v3] "v"(voff[3]), [m4] "s"(mask[4]), [v4] "v"(voff[4]), [m5] "s"(mask[5]), [v5] "v"(voff[5]) : "memory", "scc");
  else if constexpr (N == 7)
    asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b64 %[se], exec\n\ts_mov_b32 m0, %[d]\n\t"
                 "s_mov_b64 exec, %[m0]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v0], %[b]\n\t"
                 "s_add_u32 m0, m0, %[st]\n\t"
                 "s_mov_b64 exec, %[m1]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v1], %[b]\n\t"
                 "s_add_u32 m0, m0, %[st]\n\t"
                 "s_mov_b64 exec, %[m2]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v2], %[b]\n\t"
                 "s_add_u32 m0, m0, %[st]\n\t"
                 "s_mov_b64 exec, %[m3]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v3], %[b]\n\t"
                 "s_add_u32 m0, m0, %[st]\n\t"
                 "s_mov_b64 exec, %[m4]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v4], %[b]\n\t"
                 "s_add_u32 m0, m0, %[st]\n\t"
                 "s_mov_b64 exec, %[m5]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v5], %[b]\n\t"
                 "s_add_u32 m0, m0, %[st]\n\t"
                 "s_mov_b64 exec, %[m6]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v6], %[b]\n\t"
                 "s_mov_b64 exec, %[se]\n\ts_mov_b32 m0, %[k]"
                 : [k] "=&s"(keep), [se] "=&s"(sexec) : [d] "s"(dst0), [st] "s"(step), [b] "s"(sbase), [m0] "s"(mask[0]), [v0] "v"(voff[0]), [m1] "s"(mask[1]), [v1] "v"(voff[1]), [m2] "s"(mask[2]), [v2] "v"(voff[2]), [m3] "s"(mask[3]), [v3] "v"(voff[3]), [m4] "s"(mask[4]), [v4] "v"(voff[4]), [m5] "s"(mask[5]), [v5] "v"(voff[5]), [m6] "s"(mask[6]), [v6] "v"(voff[6]) : "memory", "scc");
  else if constexpr (N == 8)
    asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b64 %[se], exec\n\ts_mov_b32 m0, %[d]\n\t"
                 "s_mov_b64 exec, %[m0]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v0], %[b]\n\t"
                 "s_add_u32 m0, m0, %[st]\n\t"
                 "s_mov_b64 exec, %[m1]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v1], %[b]\n\t"
                 "s_add_u32 m0, m0, %[st]\n\t"
                 "s_mov_b64 exec, %[m2]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v2], %[b]\n\t"
                 "s_add_u32 m0, m0, %[st]\n\t"
                 "s_mov_b64 exec, %[m3]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v3], %[b]\n\t"
                 "s_add_u32 m0, m0, %[st]\n\t"
                 "s_mov_b64 exec, %[m4]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v4], %[b]\n\t"
                 "s_add_u32 m0, m0, %[st]\n\t"
                 "s_mov_b64 exec, %[m5]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v5], %[b]\n\t"
                 "s_add_u32 m0, m0, %[st]\n\t"
                 "s_mov_b64 exec, %[m6]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v6], %[b]\n\t"
                 "s_add_u32 m0, m0, %[st]\n\t"
                 "s_mov_b64 exec, %[m7]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v7], %[b]\n\t"
                 "s_mov_b64 exec, %[se]\n\ts_mov_b32 m0, %[k]"
                 : [k] "=&s"(keep), [se] "=&s"(sexec) : [d] "s"(dst0), [st] "s"(step), [b] "s"(sbase), [m0] "s"(mask[0]), [v0] "v"(voff[0]), [m1] "s"(mask[1]), [v1] "v"(voff[1]), [m2] "s"(mask[2]), [v2] "v"(voff[2]), [m3] "s"(mask[3]), [v3] "v"(voff[3]), [m4] "s"(mask[4]), [v4] "v"(voff[4]), [m5] "s"(mask[5]), [v5] "v"(voff[5]), [m6] "s"(mask[6]), [v6] "v"(voff[6]), [m7] "s"(mask[7]), [v7] "v"(voff[7]) : "memory", "scc");
}

// slab / patch geometry of a (R, NI) configuration: slabs of R output rows (a multiple of 4) x the full width, NI whole
// images per block when R covers the plane; patch positions live in skewed float4 slots pos + pos/8
struct Geo { int R, NI, nbands, S, TX, PR, PW, npos, rawF4, tps; };
inline bool geo(const ConvDesc& d, const ConvCfg& cfg, int max_tiles, Geo* g) {
  if (cfg.R < 4 || (cfg.R & 3) || cfg.NI < 1) return false;
  g->TX = (d.W + 3) / 4;
  const int Hc = (d.H + 3) / 4 * 4;
  g->R = std::min(cfg.R, Hc); g->NI = cfg.NI;
  g->nbands = (d.H + g->R - 1) / g->R;
  if (g->NI > 1 && g->nbands > 1) return false;               // several slabs per block only for whole images
  g->S = d.B * g->nbands;
  g->tps = (g->R / 4) * g->TX;
  if (g->NI * g->tps > max_tiles) return false;
  g->PR = g->R + 2; g->PW = 4 * g->TX + 2;
  g->npos = g->NI * g->PR * g->PW;
  g->rawF4 = (g->npos + g->npos / 8 + 1 + 63) & ~63;          // skewed slots, whole 64-slot DMA pieces
  if ((long)d.B * d.H * d.W * std::max(std::max(d.in_cs, d.out_cs), d.res_cs) >= (1L << 31)) return false;
  return true;
}

// U = G g G^T per (co, ci), float64 on the host -> [36][Cout][Cin]
inline void u_transform(const float* w_oihw, int Cout, int Cin, std::vector<double>* u) {
  static const double G[6][3] = {{1.0 / 4, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                 {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
  u->assign((size_t)36 * Cout * Cin, 0.0);
  for (size_t oc = 0; oc < (size_t)Cout * Cin; ++oc) {
    const float* gk = w_oihw + oc * 9;
    double t[6][3];
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 3; ++j) t[i][j] = G[i][0] * gk[0 * 3 + j] + G[i][1] * gk[1 * 3 + j] + G[i][2] * gk[2 * 3 + j];
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) (*u)[(size_t)(i * 6 + j) * Cout * Cin + oc] = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
  }
}

}  // namespace w4
