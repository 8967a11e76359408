// 3x3 convolutions (stride 1 or 2, pad 1) as a register-direct gather GEMM - ALG 10 (round 3).
//
// The stride-2 3x3 convs of HRNet (transitions, the down paths of every fuse layer, the cls head: hrnet.py:196-264,
// hrnet_cls.py:306-353) and of ResNet-50's layer2-4.0 (resnet.py:101-121) cannot use Winograd and ran at 56-100 TFLOP/s
// on the LDS-staged direct kernels (ALG 0-2): their K is short (9 * Cin with Cin = 48 .. 192 on the fuse paths = 3-12
// patch slices), so a block spends much of its life filling and draining the patch pipeline, and in the forward these
// launches run alone (1.9 ms of the 15 ms W48 forward has exactly one of them resident).
//
// Here the conv is the GEMM  D[co][pix] = sum_{tap, ci} W[co][ci][tap] X[ci][pix + tap]  walked over K = 9 * Cin / 16
// steps (tap innermost: the nine taps of a 16-channel slice touch the same / neighbouring cache lines) exactly like the 1x1
// kernel of gemm1x1.hip (ALG 6): no LDS, no barrier, no prologue - every wave free-runs over K with the operands of the next
// D-1 steps in flight in registers; the 9-fold re-read of the input is served by the vector L1 / L2 (waves of a block that
// share pixels or weights meet there).  In L16 the 16-channel slice of a pixel is 64 contiguous bytes = the 16-byte quads of
// the four channel groups of the MFMA B operand.  Zero padding: a lane whose tap falls outside the image loads its centre
// pixel instead (always in range) and the value is replaced by zero (one mask bit per tap and sub-tile).
//
// Operand roles as everywhere: packed weight fragments (conv_pack_weights, ks = 3: [tap][Cin/16][Cout/16][64] float4) = A
// operand, pixels = B operand; a lane ends up with 4 consecutive output channels of one pixel.
#include "conv_mfma_types.h"

namespace {

struct G3Params {
  const float* in;       // slice offsets folded into the pointers
  const float* res;
  float* out;
  const float4* wfrag;   // [9 taps][Cin/16][Cout16/16][64] float4
  const float* bias;
  int P;                 // output pixels B*Ho*Wo
  int H, W, Ho, Wo, stride;
  int nC16, nT16, WM, WN;
  int in_rs, in_ss, res_rs, out_rs, out_ss;
  int act, res_after_act, relu_from;
  FastDiv dWo, dHo;
};

template <int MT, int NT, int D, bool HAS_RES, int SCHED>
__global__ void __launch_bounds__(512)
gemm3x3_kernel(const G3Params p) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave % p.WM, wn = wave / p.WM;
  const int idx = lane & 15, g = lane >> 4;
  const int mt0 = (blockIdx.x * p.WM + wm) * MT;        // first 16-pixel sub-tile of this wave
  const int nt0 = (blockIdx.y * p.WN + wn) * NT;        // first 16-channel tile of this wave
  if (nt0 >= p.nT16 || mt0 * 16 >= p.P) return;         // wave-uniform; there are no barriers in this kernel

  int boff[MT];      // float offset of the centre tap of this lane's pixel (slice 0, channel quad g)
  int vmask[MT];     // bit (3r + s): tap (r, s) lies inside the image
  int orow[MT];      // output image row (b*Ho + y) or -1
  int ox16[MT];      // 16 * x
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int pix = (mt0 + m) * 16 + idx;
    const uint32_t pc = (uint32_t)min(pix, p.P - 1);    // dead lanes recompute the last pixel
    const uint32_t row = fdiv(pc, p.dWo);
    const uint32_t x = pc - row * p.Wo;
    const uint32_t b = fdiv(row, p.dHo);
    const int iy = (int)(row - b * p.Ho) * p.stride, ix = (int)x * p.stride;
    boff[m] = (int)((b * p.H + (uint32_t)iy) * (uint32_t)p.in_rs) + ix * 16 + 4 * g;
    const int ym = (iy >= 1 ? 1 : 0) | 2 | (iy + 1 < p.H ? 4 : 0);
    const int xm = (ix >= 1 ? 1 : 0) | 2 | (ix + 1 < p.W ? 4 : 0);
    vmask[m] = ((ym & 1) ? xm : 0) | (xm << 3) | ((ym & 4) ? (xm << 6) : 0);
    orow[m] = pix < p.P ? (int)row : -1;
    ox16[m] = (int)x * 16;
  }
  const float4* wl = p.wfrag + (size_t)nt0 * 64 + lane;
  const int wslice = p.nT16 * 64;                        // float4 per (tap, K slice)
  int woff[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) woff[n] = (nt0 + n < p.nT16) ? n * 64 : 0;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float4 a[D][NT], b[D][MT];
  // one operand load of K step c (= 9 * slice + tap) into stage s: pieces 0..NT-1 = weight fragments, NT.. = pixel sub-tiles
  auto load_piece = [&](int s, int c, int i) {           // unconditional (c is clamped by the caller); c is wave-uniform
    const int c16 = (c * 7282) >> 16;                    // c / 9 (exact for c < 3277: Cin <= 5824)
    const int tap = c - 9 * c16;
    if (i < NT) {
      a[s][i] = wl[(size_t)(tap * p.nC16 + c16) * wslice + woff[i]];
    } else {
      const int m = i - NT;
      const int r = (tap * 11) >> 5;                     // tap / 3 for tap < 9
      const int toff = (r - 1) * p.in_rs + (tap - 3 * r - 1) * 16;
      const bool ok = (vmask[m] >> tap) & 1;
      float4 v = *reinterpret_cast<const float4*>(p.in + boff[m] + (ok ? toff : 0) + (size_t)c16 * p.in_ss);
      b[s][m] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto load = [&](int s, int c) {
#pragma unroll
    for (int i = 0; i < NT + MT; ++i) load_piece(s, c, i);
  };
  constexpr int G = SCHED & 15;                                        // MFMAs per pinned load
  constexpr int K0 = (SCHED & 16) ? 4 * MT * NT - G * (NT + MT) : 0;
  auto mma = [&](int s, bool ld, int sn, int cn) {
    int k = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const float wj = (j == 0) ? a[s][n].x : (j == 1) ? a[s][n].y : (j == 2) ? a[s][n].z : a[s][n].w;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          if constexpr (SCHED > 0) {
            const int kk = k - K0;
            if (ld && kk >= 0 && kk % G == 0 && kk / G < NT + MT) {
              load_piece(sn, cn, kk / G);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          const float bj = (j == 0) ? b[s][m].x : (j == 1) ? b[s][m].y : (j == 2) ? b[s][m].z : b[s][m].w;
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wj, bj, acc[m][n], 0, 0, 0);
          if constexpr (SCHED > 0) {
            if (ld && k >= K0 - 1 && k < K0 + G * (NT + MT)) __builtin_amdgcn_sched_barrier(0);
          }
          ++k;
        }
      }
  };
  const int nsteps = 9 * p.nC16;
  const int last = nsteps - 1;
#pragma unroll
  for (int s = 0; s < D - 1; ++s) load(s, min(s, last));
  const int nfull = nsteps / D * D;
  for (int c0 = 0; c0 < nfull; c0 += D) {
#pragma unroll
    for (int u = 0; u < D; ++u) {                        // step c0 + u lives in stage u (c0 is a multiple of D)
      if constexpr (SCHED == 0) load((u + D - 1) % D, min(c0 + u + D - 1, last));
      mma(u, true, (u + D - 1) % D, min(c0 + u + D - 1, last));
    }
  }
#pragma unroll
  for (int u = 0; u < D - 1; ++u) {                      // tail: nsteps % D steps, already (being) loaded
    if (nfull + u < nsteps) mma(u, false, 0, 0);
  }

  // ---- epilogue: shift (+ residual) (activation) -> L16 channel slice (as gemm1x1.hip) ------------------------------
  int ob[MT], rb[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    ob[m] = orow[m] >= 0 ? orow[m] * p.out_rs + ox16[m] + g * 4 : -1;
    rb[m] = max(orow[m], 0) * p.res_rs + ox16[m] + g * 4;
  }
  float4 sh[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) sh[n] = *reinterpret_cast<const float4*>(p.bias + min(nt0 + n, p.nT16 - 1) * 16 + g * 4);
  auto load_res = [&](int n, float4* r) {
    const int co = min(nt0 + n, p.nT16 - 1) * p.out_ss;
#pragma unroll
    for (int m = 0; m < MT; ++m) r[m] = *reinterpret_cast<const float4*>(p.res + rb[m] + co);
  };
  float4 rcur[MT], rnext[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) rcur[m] = rnext[m] = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (HAS_RES) load_res(0, rcur);
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    if constexpr (HAS_RES) { if (n + 1 < NT) load_res(n + 1, rnext); }
    const int co = (nt0 + n) * 16 + g * 4;
    const bool nok = nt0 + n < p.nT16;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      f32x4 v = acc[m][n];
      v[0] += sh[n].x; v[1] += sh[n].y; v[2] += sh[n].z; v[3] += sh[n].w;
      const float4 r = rcur[m];
      if (!p.res_after_act) { v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w; }
      if (p.act == 1 || (p.act == 3 && co >= p.relu_from)) {
        v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
      } else if (p.act == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = 1.f / (1.f + __expf(-v[e]));
      }
      if (p.res_after_act) { v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w; }
      if (nok && ob[m] >= 0)
        *reinterpret_cast<float4*>(p.out + ob[m] + (nt0 + n) * p.out_ss) = make_float4(v[0], v[1], v[2], v[3]);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) rcur[m] = rnext[m];
  }
}

// cfg.NI: load schedule 1 = hipcc's order; 3 = a pinned load per 4 MFMAs from the start of the step; 6 = per 4 at its end
constexpr int g3_sched(int NI) { return NI == 3 ? 4 : NI == 6 ? 16 + 4 : 0; }

template <int MT, int NT, int SCHED>
int launch_s(int D, const G3Params& p, dim3 grid, int nthreads, hipStream_t stream) {
  const bool r = p.res != nullptr;
  if (D == 2 && r) hipLaunchKernelGGL((gemm3x3_kernel<MT, NT, 2, true, SCHED>), grid, dim3(nthreads), 0, stream, p);
  else if (D == 2) hipLaunchKernelGGL((gemm3x3_kernel<MT, NT, 2, false, SCHED>), grid, dim3(nthreads), 0, stream, p);
  else if (r) hipLaunchKernelGGL((gemm3x3_kernel<MT, NT, 3, true, SCHED>), grid, dim3(nthreads), 0, stream, p);
  else hipLaunchKernelGGL((gemm3x3_kernel<MT, NT, 3, false, SCHED>), grid, dim3(nthreads), 0, stream, p);
  POCO_HIP_CHECK(hipGetLastError());
  return POCO_OK;
}

template <int MT, int NT>
int launch_d(int D, int NI, const G3Params& p, dim3 grid, int nthreads, hipStream_t stream) {
  if (NI == 3) return launch_s<MT, NT, g3_sched(3)>(D, p, grid, nthreads, stream);
  if (NI == 6) return launch_s<MT, NT, g3_sched(6)>(D, p, grid, nthreads, stream);
  return launch_s<MT, NT, 0>(D, p, grid, nthreads, stream);
}

bool tile_ok(int MT, int NT) {
  return (MT == 2 && NT == 4) || (MT == 4 && (NT == 2 || NT == 3 || NT == 4)) || (MT == 7 && (NT == 2 || NT == 3 || NT == 4)) ||
         (MT == 8 && NT == 2);
}

}  // namespace

// cfg: {MT, NT, WM, WN, R = prefetch depth D (2|3), NI = load schedule (1|3|6), ALG = 10}
bool gemm3x3_cfg_valid(const ConvDesc& d, const ConvCfg& cfg) {
  const int Ho = (d.H - 1) / d.stride + 1, Wo = (d.W - 1) / d.stride + 1;
  const long P = (long)d.B * Ho * Wo;
  return d.ks == 3 && (d.stride == 1 || d.stride == 2) && d.Cin % 16 == 0 && d.Cout % 16 == 0 && d.Cin <= 5824 && tile_ok(cfg.MT, cfg.NT) &&
         cfg.WM >= 1 && cfg.WN >= 1 && cfg.WM * cfg.WN <= 8 && (cfg.R == 2 || cfg.R == 3) && (cfg.NI == 1 || cfg.NI == 3 || cfg.NI == 6) &&
         (g3_sched(cfg.NI) & 15) * (cfg.MT + cfg.NT) <= 4 * cfg.MT * cfg.NT && P < (1L << 27) &&
         (long)d.B * d.H * d.in_cs * d.W < (1L << 31) && P * std::max(d.out_cs, d.res_cs) < (1L << 31);
}

int gemm3x3_launch(const ConvDesc& d, const ConvCfg& cfg, hipStream_t stream) {
  if (!gemm3x3_cfg_valid(d, cfg)) {
    poco_set_error("gemm3x3: ALG 10 needs ks = 3, stride 1|2, (MT,NT) in {(2,4),(4,2..4),(7,2..4),(8,2)}, WM*WN <= 8, R (depth) 2|3, NI (load schedule) 1|3|6 with room for its loads in a step");
    return POCO_ERR_ARG;
  }
  if ((d.in_cs | d.in_co | d.out_cs | d.out_co | d.res_cs | d.res_co) & 3) {
    poco_set_error("conv: channel strides/offsets must be multiples of 4");
    return POCO_ERR_ARG;
  }
  G3Params p{};
  p.H = d.H; p.W = d.W; p.stride = d.stride;
  p.Ho = (d.H - 1) / d.stride + 1; p.Wo = (d.W - 1) / d.stride + 1;
  p.in = d.in + l16_chan_off(d.in_co, d.W);
  p.res = d.res ? d.res + l16_chan_off(d.res_co, p.Wo) : nullptr;
  p.out = d.out + l16_chan_off(d.out_co, p.Wo);
  p.wfrag = reinterpret_cast<const float4*>(d.wfrag); p.bias = d.bias;
  p.P = d.B * p.Ho * p.Wo; p.nC16 = d.Cin / 16; p.nT16 = d.Cout / 16; p.WM = cfg.WM; p.WN = cfg.WN;
  p.in_rs = d.in_cs * d.W; p.in_ss = d.W * 16;
  p.res_rs = d.res_cs * p.Wo; p.out_rs = d.out_cs * p.Wo; p.out_ss = p.Wo * 16;
  p.act = d.act; p.res_after_act = d.res_after_act; p.relu_from = d.relu_from;
  p.dWo = make_fastdiv(p.Wo); p.dHo = make_fastdiv(p.Ho);
  const int mtiles = (p.P + 15) / 16;
  const dim3 grid((mtiles + cfg.MT * cfg.WM - 1) / (cfg.MT * cfg.WM), (p.nT16 + cfg.NT * cfg.WN - 1) / (cfg.NT * cfg.WN));
  const int nthreads = cfg.WM * cfg.WN * 64;
#define G3_CASE(mt, nt) if (cfg.MT == mt && cfg.NT == nt) return launch_d<mt, nt>(cfg.R, cfg.NI, p, grid, nthreads, stream)
  G3_CASE(2, 4); G3_CASE(4, 2); G3_CASE(4, 3); G3_CASE(4, 4); G3_CASE(7, 2); G3_CASE(7, 3); G3_CASE(7, 4); G3_CASE(8, 2);
#undef G3_CASE
  poco_set_error("gemm3x3: tile not instantiated");
  return POCO_ERR_ARG;
}
