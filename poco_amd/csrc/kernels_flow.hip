// Conditional RealNVP coupling layers (pocolib/models/layers/real_nvp.py:25-65 with the s/t MLPs of
// pocolib/models/head/nf_head.py:13-17; called on N = B*24 rows of 9 residuals with the crop's 512-d context repeated per
// joint, nf_head.py:93-110) on the fp32 MFMA.
//
// Round 3 rewrite (the round-2 kernel - one wave per 8 rows, lane = hidden unit, every wave re-reading the 133 KB first-layer
// matrix of every net from L2 and feeding each FMA from an LDS broadcast - took 140 us for 1536 rows x 2 layers and 413 us for
// 3072 rows x 6 layers).  Two steps:
//   A. everything the context contributes to the first Linear of ALL 2*L MLPs is one GEMM, independent of the coupling
//      recursion:  P[row][(layer, net, hidden)] = W0[:, 9:] . ctx[row] + b0   (rows x 512 x L*128; 98.6 % of the flops).
//      It runs on the engine's own GEMM kernels (linear_mfma / gemm1x1: [rows][512] row-major IS their "L16 vector" layout)
//      into a scratch buffer - once per context row, i.e. once per crop when the caller passes `rep` = 24 instead of a
//      24x repeated context.
//   B. the recursion itself (this file): a block of two waves owns 16 rows; wave 0 evaluates the s-MLPs, wave 1 the t-MLPs.
//      Weights are the MFMA A operand and rows the B operand (the convention of all kernels here), so the accumulator of a
//      layer - lane (row, g) holds hidden units 16*mt + 4g .. +3 of its row - IS the B operand of the next layer's K steps
//      (K permuted in the packed fragments: step (mt, i) contracts units {16*mt + 4g + i}): 16 + 64 + 16 MFMAs per MLP with
//      the activations never leaving the registers, no LDS except the 2 x 1 KiB s/t exchange per layer (one barrier).
//      The 9-vector z lives in the same lane layout (d = 4g + i, zero padded to 16).  The 24 fragment quads of the next
//      layer are prefetched under the current layer's MFMAs (two-layer ping-pong; L is even: masks come in pairs,
//      nf_head.py:20-21).
#include "common.h"
#include "kernels.h"

namespace {

constexpr int D = 9;
constexpr int HID = 64;
constexpr int NFRAG = 24;      // float4 fragments per lane and MLP: first layer (z part) 4, second 16, third 4

__device__ __forceinline__ float leaky(float v) { return v > 0.f ? v : 0.01f * v; }
__device__ __forceinline__ float comp(const float4& a, int j) { return j == 0 ? a.x : j == 1 ? a.y : j == 2 ? a.z : a.w; }

struct LayerRegs {
  float4 w[NFRAG];    // A-operand fragments of this wave's MLP
  float4 p[4];        // context part of the first Linear (+ b0) for the wave's row: hidden 16*mt + 4g .. +3
  float4 b1[4];       // second bias, same lane layout
  float4 b2;          // third bias, d = 4g .. 4g+3 (zero for d >= 9)
  float4 m;           // mask of the layer, d = 4g .. 4g+3 (zero for d >= 9)
};

__device__ __forceinline__ void load_layer(LayerRegs& r, const FlowDev& f, int li, int net, const float* prow, int lane, int g) {
  const float4* wp = f.wpack + (size_t)(li * 2 + net) * NFRAG * 64 + lane;
#pragma unroll
  for (int k = 0; k < NFRAG; ++k) r.w[k] = wp[k * 64];
  const float* pp = prow + (li * 2 + net) * HID + 4 * g;
  const float* bp = f.b1 + (li * 2 + net) * HID + 4 * g;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    r.p[mt] = *reinterpret_cast<const float4*>(pp + 16 * mt);
    r.b1[mt] = *reinterpret_cast<const float4*>(bp + 16 * mt);
  }
  r.b2 = *reinterpret_cast<const float4*>(f.b2 + (li * 2 + net) * 16 + 4 * g);
  r.m = *reinterpret_cast<const float4*>(f.mask16 + li * 16 + 4 * g);
}

// one MLP (this wave's net) of one coupling layer on the masked state zm -> raw third-layer output (before tanh / mask)
__device__ __forceinline__ f32x4 mlp(const LayerRegs& r, const float (&zm)[4]) {
  f32x4 a1[4], a2[4], a3;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) a1[mt] = (f32x4){r.p[mt].x, r.p[mt].y, r.p[mt].z, r.p[mt].w};
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) a1[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(r.w[mt], i), zm[i], a1[mt], 0, 0, 0);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    a2[mt] = (f32x4){r.b1[mt].x, r.b1[mt].y, r.b1[mt].z, r.b1[mt].w};
#pragma unroll
    for (int i = 0; i < 4; ++i) a1[mt][i] = leaky(a1[mt][i]);
  }
  // second Linear: fragments packed [k slice = mt][n-tile = mo]; consecutive MFMAs never share an accumulator
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int mo = 0; mo < 4; ++mo)
        a2[mo] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(r.w[4 + mt * 4 + mo], i), a1[mt][i], a2[mo], 0, 0, 0);
  a3 = (f32x4){r.b2.x, r.b2.y, r.b2.z, r.b2.w};
  // third Linear (9 of 16 output rows used): a single accumulator chain, 16 dependent MFMAs
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int i = 0; i < 4; ++i) a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(r.w[20 + mt], i), leaky(a2[mt][i]), a3, 0, 0, 0);
  return a3;
}

__global__ void __launch_bounds__(128)
realnvp_mfma_kernel(const FlowDev f, const float* __restrict__ x, const float* __restrict__ P, int rep,
                    float* __restrict__ out, int N, int forward) {
  __shared__ float4 xch[2][2][64];          // [layer parity][net][lane]: s / t of the layer
  const int lane = threadIdx.x & 63;
  const int net = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int idx = lane & 15, g = lane >> 4;
  const int r = blockIdx.x * 16 + idx;
  const int rc = min(r, N - 1);             // dead lanes recompute the last row
  const float* prow = P + (size_t)(rc / rep) * (f.L * 2 * HID);
  float z[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) z[i] = (4 * g + i < D) ? x[(size_t)rc * D + 4 * g + i] : 0.f;
  float ld = 0.f;

  LayerRegs ra, rb;
  load_layer(ra, f, forward ? 0 : f.L - 1, net, prow, lane, g);

  auto couple = [&](const LayerRegs& cur, int step) {
    float zm[4];
    const float mk[4] = {cur.m.x, cur.m.y, cur.m.z, cur.m.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) zm[i] = z[i] * mk[i];
    f32x4 o = mlp(cur, zm);
    float4 ov;
    if (net == 0) ov = make_float4(tanhf(o[0]) * (1.f - mk[0]), tanhf(o[1]) * (1.f - mk[1]), tanhf(o[2]) * (1.f - mk[2]), tanhf(o[3]) * (1.f - mk[3]));
    else ov = make_float4(o[0] * (1.f - mk[0]), o[1] * (1.f - mk[1]), o[2] * (1.f - mk[2]), o[3] * (1.f - mk[3]));
    xch[step & 1][net][lane] = ov;
    __syncthreads();
    const float4 s4 = xch[step & 1][0][lane], t4 = xch[step & 1][1][lane];
    const float s[4] = {s4.x, s4.y, s4.z, s4.w}, t[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      z[i] = forward ? zm[i] + (1.f - mk[i]) * (z[i] * expf(s[i]) + t[i])          // real_nvp.py:36
                     : (1.f - mk[i]) * (z[i] - t[i]) * expf(-s[i]) + zm[i];        // real_nvp.py:51
      ld -= s[i];                                                                 // real_nvp.py:52 (partial: this lane's 4 d)
    }
  };

  for (int step = 0; step < f.L; step += 2) {
    const int l1 = forward ? step + 1 : f.L - 2 - step;
    load_layer(rb, f, l1, net, prow, lane, g);                  // in flight under layer `step`
    couple(ra, step);
    if (step + 2 < f.L) load_layer(ra, f, forward ? step + 2 : f.L - 3 - step, net, prow, lane, g);
    couple(rb, step + 1);
  }

  if (net != 0) return;
  if (forward) {
    if (r < N) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (4 * g + i < D) out[(size_t)r * D + 4 * g + i] = z[i];
    }
  } else {
    float q = z[0] * z[0] + z[1] * z[1] + z[2] * z[2] + z[3] * z[3];
    q += __shfl_xor(q, 16); q += __shfl_xor(q, 32);
    ld += __shfl_xor(ld, 16); ld += __shfl_xor(ld, 32);
    // MultivariateNormal(0, I_9).log_prob(z) + log_det   (real_nvp.py:64-65)
    if (g == 0 && r < N) out[r] = -0.5f * q - 0.5f * D * 1.8378770664093453f + ld;
  }
}

}  // namespace

size_t realnvp_scratch_floats(const FlowDev& f, int ctx_rows) { return (size_t)ctx_rows * f.L * 2 * HID; }

int launch_realnvp(const FlowDev& f, const float* x, const float* ctx, int rep, float* out, int N, int forward, float* scratch,
                   hipStream_t s) {
  const int rows = (N + rep - 1) / rep;
  // step A: P = ctx . W0[:, 9:]^T + b0 for all 2*L MLPs at once, on the engine's GEMM kernels
  ConvDesc d{};
  d.in = ctx; d.in_cs = f.ctx; d.out = scratch; d.out_cs = f.L * 2 * HID;
  d.wfrag = f.wctx_frag; d.bias = f.bctx;
  d.B = rows; d.H = 1; d.W = 1; d.Cin = f.ctx; d.Cout = f.L * 2 * HID; d.ks = 1; d.stride = 1; d.act = 0;
  const ConvCfg cfg = rows <= 256 ? ConvCfg{1, 1, 8, 1, 1, 1, 5}        // few rows: split-K linear kernel (latency-bound)
                                  : ConvCfg{2, 4, 2, 1, 2, 1, 6};       // register-direct GEMM, 32 rows x 64 columns per wave
  const int rc = conv_launch(d, cfg, s);
  if (rc != POCO_OK) return rc;
  // step B
  hipLaunchKernelGGL(realnvp_mfma_kernel, dim3((N + 15) / 16), dim3(128), 0, s, f, x, scratch, rep, out, N, forward);
  POCO_HIP_CHECK(hipGetLastError());
  return POCO_OK;
}
