// 3x3 stride-1 convolutions with FEW channels (Cin = Cout = 32 | 64) as a weight-stationary direct conv on the fp32 MFMA - ALG 15
// (round 5).  The BasicBlock convs of HRNet-W32's two high-resolution branches (hrnet.py:42-58: 56x56 32->32 and 28x28 64->64,
// 64 launches each per forward) are 1.85 GFLOP of direct convolution per launch at 32 crops.  Winograd F(4x4) (ALG 13) executes a
// quarter of that, but with 8 ... 16 four-channel K slices per item its launches are almost all fixed cost: 32.8 / 25.0 us at
// 0.10 of the MFMA peak - the dominant symbol of the PARE forward and the furthest below its roofline.  At these
// channel counts the whole weight tensor of ONE output n-tile fits a wave's registers (9 taps x C / 16 slices x 4 = 72 | 144
// VGPRs), so the direct conv needs no weight traffic at all inside its loop:
//
//   * a wave owns one 16-channel n-tile (weights = MFMA A operand, read once from the packed direct-conv fragments of
//     conv_pack_weights(ks = 3)) and 7 pixel sub-tiles of the block's chunk of R output rows; the C / 16 n-tile waves of a pixel
//     group read the same pixels;
//   * the chunk's input rows (+ halo, zero padding written out) are staged once in LDS in the L16 order of the activation itself
//     ([slice][pixel][16 channels]): the B operand of (tap, slice) is ONE conflict-free ds_read_b128 per sub-tile at
//     `lane base + tap offset + slice stride` (a lane's four channels = four K steps), requested one (tap, slice) ahead;
//   * 36 C / 16 MFMAs per sub-tile and nothing else in the loop; several blocks per CU (45 KB of LDS each) overlap one block's
//     staging and epilogue (shift, residual, ReLU, 16-byte stores in L16) with another's MFMAs.
#include "conv_mfma_types.h"
#include <algorithm>
#include <string>

namespace {

struct WSParams {
  const float* in;
  const float* res;
  float* out;
  const float4* wfrag;   // conv_pack_weights(ks = 3): [tap][Cin/16][Cout/16][64] float4
  const float* bias;
  int H, W, R;           // plane, output rows per chunk
  int chunks;            // chunks per image = ceil(H / R)
  int in_rs, in_ss, res_rs, out_rs, out_ss;
  int act, res_after_act;
  FastDiv dW;
};

constexpr int WS_MT = 7;           // pixel sub-tiles per wave

template <int C16>                 // Cin / 16 = Cout / 16
__global__ void __launch_bounds__(C16 == 4 ? 256 : 512)     // 64 channels: 144 weight registers, one 4-wave block per CU
conv3x3ws_kernel(const WSParams p) {
  extern __shared__ float4 patch[];                      // [C16][(R + 2) * (W + 2)][4 quads]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int idx = lane & 15, g = lane >> 4;
  const int nt = wave % C16, grp = wave / C16;
  const int b = blockIdx.x / p.chunks, y0 = (blockIdx.x - b * p.chunks) * p.R;
  const int PW = p.W + 2, PH = p.R + 2, npix = PH * PW;

  // weights of this wave's n-tile: 9 taps x C16 slices, in registers for the whole block
  float4 a[9][C16];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int s = 0; s < C16; ++s) a[t][s] = p.wfrag[((size_t)(t * C16 + s) * C16 + nt) * 64 + lane];

  // stage the patch: element i = (slice, patch pixel, quad); global and LDS runs are contiguous along a row
  const float* ib = p.in + (size_t)b * p.H * p.in_rs;
  // (four independent, unconditional loads per thread and round: clamped addresses, padding = a multiplication by 0 - a load
  //  under a branch is one dependent memory round trip per element)
  const int total = C16 * npix * 4, bd = blockDim.x;
  for (int i0 = tid; i0 < total; i0 += 4 * bd) {
    float4 v[4];
    float keep[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = min(i0 + u * bd, total - 1);
      const int q = i & 3, e = i >> 2, s = e / npix, pp = e - s * npix;
      const int py = pp / PW, px = pp - py * PW;
      const int iy = y0 + py - 1, ix = px - 1;
      keep[u] = ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) ? 1.f : 0.f;
      v[u] = *reinterpret_cast<const float4*>(ib + (size_t)min(max(iy, 0), p.H - 1) * p.in_rs + (size_t)s * p.in_ss + min(max(ix, 0), p.W - 1) * 16 + q * 4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i0 + u * bd < total) patch[i0 + u * bd] = make_float4(v[u].x * keep[u], v[u].y * keep[u], v[u].z * keep[u], v[u].w * keep[u]);
  }
  __syncthreads();

  // this lane's pixel of every sub-tile: output pixel pl (chunk-local, row-major) -> window origin (y, x) in the patch
  const int ntile = (p.R * p.W + 15) >> 4;
  int base[WS_MT], opix[WS_MT];
#pragma unroll
  for (int m = 0; m < WS_MT; ++m) {
    const int mt = grp * WS_MT + m;
    const int pl = min(mt * 16 + idx, p.R * p.W - 1);
    const int y = (int)fdiv((uint32_t)pl, p.dW), x = pl - y * p.W;
    base[m] = (y * PW + x) * 4 + g;                      // float4 index of (window origin, quad g) in slice 0
    opix[m] = (mt < ntile && mt * 16 + idx < p.R * p.W && y0 + y < p.H) ? (y << 16) | x : -1;
  }
  f32x4 acc[WS_MT];
#pragma unroll
  for (int m = 0; m < WS_MT; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 bv[2][WS_MT];
  auto fetch = [&](int k, float4 (&dst)[WS_MT]) {        // k = tap * C16 + slice
    const int t = k / C16, s = k - t * C16;
    const int off = ((t / 3) * PW + (t % 3)) * 4 + s * npix * 4;
#pragma unroll
    for (int m = 0; m < WS_MT; ++m) dst[m] = patch[base[m] + off];
  };
  fetch(0, bv[0]);
#pragma unroll
  for (int k = 0; k < 9 * C16; ++k) {
    if (k + 1 < 9 * C16) fetch(k + 1, bv[(k + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);
    const float4 w = a[k / C16][k % C16];
    // K step outermost: consecutive MFMAs never share an accumulator
#pragma unroll
    for (int m = 0; m < WS_MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, bv[k & 1][m].x, acc[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < WS_MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, bv[k & 1][m].y, acc[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < WS_MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, bv[k & 1][m].z, acc[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < WS_MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, bv[k & 1][m].w, acc[m], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  // epilogue: shift (+ residual) (ReLU) -> channels 16 nt + 4 g .. + 3 of the lane's pixel
  const float4 sh = *reinterpret_cast<const float4*>(p.bias + nt * 16 + g * 4);
#pragma unroll
  for (int m = 0; m < WS_MT; ++m) {
    if (opix[m] < 0) continue;
    const int y = y0 + (opix[m] >> 16), x = opix[m] & 0xffff;
    const size_t row = (size_t)b * p.H + y;
    f32x4 v = acc[m];
    v[0] += sh.x; v[1] += sh.y; v[2] += sh.z; v[3] += sh.w;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.res) r = *reinterpret_cast<const float4*>(p.res + row * p.res_rs + (size_t)nt * p.out_ss + x * 16 + g * 4);
    if (!p.res_after_act) { v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w; }
    if (p.act == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
    if (p.res_after_act) { v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w; }
    *reinterpret_cast<float4*>(p.out + row * p.out_rs + (size_t)nt * p.out_ss + x * 16 + g * 4) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

size_t ws_lds_bytes(const ConvDesc& d, const ConvCfg& cfg) {
  return (size_t)(d.Cin / 16) * (cfg.R + 2) * (d.W + 2) * 4 * sizeof(float4);
}

}  // namespace

// cfg: {MT = 7, NT = 1, WM = pixel groups of 7 sub-tiles per block, WN = Cin / 16, R = output rows per chunk, NI = 1, ALG = 15}
bool conv3x3ws_cfg_valid(const ConvDesc& d, const ConvCfg& cfg) {
  const int C16 = d.Cin / 16;
  return d.ks == 3 && d.stride == 1 && d.Cin == d.Cout && (d.Cin == 32 || d.Cin == 64) && cfg.MT == WS_MT && cfg.NT == 1 && cfg.WN == C16 &&
         cfg.WM >= 1 && cfg.WM * C16 <= (C16 == 4 ? 4 : 8) && cfg.R >= 1 && cfg.R <= d.H && cfg.WM * WS_MT * 16 >= cfg.R * d.W && (cfg.WM - 1) * WS_MT * 16 < cfg.R * d.W &&
         d.W <= 4096 && cfg.R < 4096 && (d.act == 0 || d.act == 1) && ws_lds_bytes(d, cfg) <= 160 * 1024 &&
         (long)d.B * d.H * d.W * std::max(std::max(d.in_cs, d.out_cs), d.res_cs) < (1L << 31);
}

size_t conv3x3ws_lds_bytes(const ConvDesc& d, const ConvCfg& cfg) { return conv3x3ws_cfg_valid(d, cfg) ? ws_lds_bytes(d, cfg) : 0; }

int conv3x3ws_launch(const ConvDesc& d, const ConvCfg& cfg, hipStream_t stream) {
  if (!conv3x3ws_cfg_valid(d, cfg)) {
    poco_set_error("conv3x3ws: ALG 15 needs ks = 3, stride 1, Cin = Cout = 32 | 64, MT = 7, NT = 1, WN = Cin / 16, WM pixel groups covering R rows "
                   "(WM * 112 >= R * W > (WM - 1) * 112), WM * WN <= 8, activation none | ReLU");
    return POCO_ERR_ARG;
  }
  if ((d.in_cs | d.in_co | d.out_cs | d.out_co | d.res_cs | d.res_co) & 15) {
    poco_set_error("conv3x3ws: channel strides / offsets must be multiples of 16");
    return POCO_ERR_ARG;
  }
  WSParams p{};
  p.in = d.in + l16_chan_off(d.in_co, d.W);
  p.res = d.res ? d.res + l16_chan_off(d.res_co, d.W) : nullptr;
  p.out = d.out + l16_chan_off(d.out_co, d.W);
  p.wfrag = reinterpret_cast<const float4*>(d.wfrag); p.bias = d.bias;
  p.H = d.H; p.W = d.W; p.R = cfg.R; p.chunks = (d.H + cfg.R - 1) / cfg.R;
  p.in_rs = d.in_cs * d.W; p.in_ss = d.W * 16;
  p.res_rs = d.res_cs * d.W; p.out_rs = d.out_cs * d.W; p.out_ss = d.W * 16;
  p.act = d.act; p.res_after_act = d.res_after_act;
  p.dW = make_fastdiv(d.W);
  const size_t lds = ws_lds_bytes(d, cfg);
  const dim3 grid(d.B * p.chunks), block(cfg.WM * cfg.WN * 64);
  static thread_local bool configured = false;
  if (!configured) {
    hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3ws_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3ws_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e1 != hipSuccess || e2 != hipSuccess) { poco_set_error("conv3x3ws: hipFuncSetAttribute failed"); return POCO_ERR_HIP; }
    configured = true;
  }
  if (d.Cin == 32) hipLaunchKernelGGL(conv3x3ws_kernel<2>, grid, block, lds, stream, p);
  else hipLaunchKernelGGL(conv3x3ws_kernel<4>, grid, block, lds, stream, p);
  POCO_HIP_CHECK(hipGetLastError());
  return POCO_OK;
}
