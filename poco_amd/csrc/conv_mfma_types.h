// Types shared by the host logic (conv_mfma.hip) and the kernel translation units (conv_mfma_k*.hip).
#pragma once
#include "common.h"
#include <algorithm>
#include <cstdlib>

struct FastDiv {
  uint32_t magic, d;
};
__host__ inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d;
  f.magic = (uint32_t)(((1ull << 32) + d - 1) / d);
  return f;
}
// exact for n*d < 2^32 (all uses here have n, d < 65536)
__device__ __forceinline__ uint32_t fdiv(uint32_t n, FastDiv f) {
  return f.d == 1 ? n : __umulhi(n, f.magic);
}

struct ConvKParams {
  const float* in;
  const float* res;
  float* out;
  const float4* wfrag;
  const float* bias;
  int in_rs, in_ss;      // input: floats per image row (C*W) and per 16-channel slice of a row (W*16)
  int res_rs, out_rs, out_ss;   // output / residual: row stride (C*Wo), slice stride (Wo*16); slice offsets folded into the pointers
  int H, W, Ho, Wo;
  int nC16;    // Cin / 16
  int nT16;    // Cout16 / 16
  int R, NI, S;          // rows per slab, slabs per block, total slabs (= B * nbands)
  int PR, PW;            // patch rows / cols per slab
  int npos;              // NI * PR * PW
  int planeF4;           // float4 elements per LDS plane (multiple of 16; of 64 for ALG 1)
  int WM, WN;
  int NTB;               // n-tiles per block (WN * NT)
  int bufF4;             // ALG 1: float4 per LDS buffer (4 planes + KS*KS*NTB weight fragments)
  int ngroups;           // ALG 1/2: planeF4 / 64
  int nblocks_m, nb_n;   // ALG 2: tile grid walked by the persistent blocks
  int dbg;            // profiling experiments: bit0 = skip the epilogue, bit1 = skip the DMA prologue wait
  int repeat;         // K-loop repetitions (1; >1 = profiling experiment, results meaningless)
  int act;            // 0 none, 1 ReLU, 2 sigmoid, 3 ReLU for channels >= relu_from
  int relu_from;
  int res_after_act;  // add the residual after the activation (hrnet_cls.py:475-477)
  FastDiv dPW, dSlab /*PR*PW*/, dBands, dWo, dRWo;
};

constexpr int DMA_MAXG = 6;   // 64-position groups of the patch each wave may own (LDS-DMA kernels)

// One entry point per (kernel size, stride): each lives in its own translation unit so that the ~90 kernel template
// instances compile in parallel (conv_mfma_k1s1.hip, ...).
int conv_launch_k1s1(int alg, int MT, int NT, const ConvKParams& kp, dim3 grid, int nthreads, size_t lds, hipStream_t stream);
int conv_launch_k1s2(int alg, int MT, int NT, const ConvKParams& kp, dim3 grid, int nthreads, size_t lds, hipStream_t stream);
int conv_launch_k3s1(int alg, int MT, int NT, const ConvKParams& kp, dim3 grid, int nthreads, size_t lds, hipStream_t stream);
int conv_launch_k3s2(int alg, int MT, int NT, const ConvKParams& kp, dim3 grid, int nthreads, size_t lds, hipStream_t stream);
