// 1x1 convolutions (stride 1 or 2) as a register-direct GEMM — ALG 6.
//
// The Bottleneck 1x1 convs of ResNet-50 / HRNet layer1 / the cls head (resnet.py:101-121, hrnet.py:79-99,
// hrnet_cls.py:306-353) are plain GEMMs D[co][pix] = sum_k W[co][k] X[k][pix].  A 1x1 conv has no halo: in the
// L16 layout the 16-channel slice of 16 neighbouring pixels is ONE contiguous KiB, i.e. exactly the MFMA B operand
// of a wave (lane (idx, g) holds x[pixel idx][16c + 4g .. +3]), so it can be loaded global -> VGPR with one
// coalesced dwordx4 per sub-tile and needs no LDS staging at all.  The LDS-DMA kernels pay one block barrier per
// 16-channel slice, which for a 1x1 conv is only MT*NT*4 MFMAs of work; here there is no LDS and no barrier:
// every wave free-runs over K with its operands of the next D-1 slices in flight in registers, and the waves of a
// block that share pixels (same wm) or weights (same wn) meet in the vector L1.
//
// Operand roles as in the other kernels: packed weight fragments (conv_pack_weights, ks = 1) = A operand, pixels =
// B operand, so a lane ends up with 4 consecutive output channels of one pixel -> 16-B stores.
#include "conv_mfma_types.h"
#include <cstdio>
#include <cstdlib>

namespace {

struct G1Params {
  const float* in;       // slice offsets folded into the pointers
  const float* res;
  float* out;
  const float4* wfrag;   // [Cin/16][Cout16/16][64] float4
  const float* bias;
  int P;                 // output pixels B*Ho*Wo
  int H, W, Ho, Wo, stride;
  int nC16, nT16, WM, WN;
  int in_rs, in_ss, res_rs, out_rs, out_ss;
  int act, res_after_act, relu_from;
  FastDiv dWo, dHo;
  // second source (DUAL instances): K slices >= nC16a come from in2, read at (stride2 * y, stride2 * x) of its H2 x W2 plane
  const float* in2;
  int nC16a, in2_rs, in2_ss, H2, stride2;
};

// SCHED = 0 leaves the order of loads and MFMAs to hipcc, which sinks every load to just before its first use: the load
// latency is then hidden only by the second wave of the SIMD.  SCHED > 0 pins the order: one operand load of the next
// slice but D-2 per G = SCHED & 15 MFMAs, from the start of the slice or (SCHED & 16) ending with the slice (all of
// them in front of the MFMAs would stall the wave at the vector-memory issue).  Which is fastest depends on the tile and
// on how many waves a SIMD gets (measured: +20 % for the K = 1024 layers at 14x14 with one wave per SIMD, -15 % for
// 56x56 tiles with two), so it is a tuned parameter (cfg.NI).
template <int MT, int NT, int D, bool HAS_RES, bool DUAL = false, int SCHED = 0>
__global__ void __launch_bounds__(512)
gemm1x1_kernel(const G1Params p) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave % p.WM, wn = wave / p.WM;
  const int idx = lane & 15, g = lane >> 4;
  const int mt0 = (blockIdx.x * p.WM + wm) * MT;        // first 16-pixel sub-tile of this wave
  const int nt0 = (blockIdx.y * p.WN + wn) * NT;        // first 16-channel tile of this wave
  if (nt0 >= p.nT16 || mt0 * 16 >= p.P) return;         // wave-uniform; there are no barriers in this kernel

  int boff[MT];      // float offset of this lane's pixel (slice 0, channel quad g) in the input
  int boff2[DUAL ? MT : 1];   // ... in the second source
  int orow[MT];      // output image row (b*Ho + y) or -1
  int ox16[MT];      // 16 * x
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int pix = (mt0 + m) * 16 + idx;
    const uint32_t pc = (uint32_t)min(pix, p.P - 1);    // dead lanes re-read the last pixel
    const uint32_t row = fdiv(pc, p.dWo);
    const uint32_t x = pc - row * p.Wo;
    uint32_t irow = row, ix = x;
    if (p.stride == 2) {
      const uint32_t b = fdiv(row, p.dHo);
      irow = b * p.H + (row - b * p.Ho) * 2;
      ix = x * 2;
    }
    boff[m] = (int)(irow * (uint32_t)p.in_rs + ix * 16u) + 4 * g;
    if constexpr (DUAL) {
      const uint32_t b2 = fdiv(row, p.dHo);
      const uint32_t irow2 = b2 * (uint32_t)p.H2 + (row - b2 * (uint32_t)p.Ho) * (uint32_t)p.stride2;
      boff2[m] = (int)(irow2 * (uint32_t)p.in2_rs + x * (uint32_t)p.stride2 * 16u) + 4 * g;
    }
    orow[m] = pix < p.P ? (int)row : -1;
    ox16[m] = (int)x * 16;
  }
  const float4* wl = p.wfrag + (size_t)nt0 * 64 + lane;
  const int wslice = p.nT16 * 64;                        // float4 per K slice
  int woff[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) woff[n] = (nt0 + n < p.nT16) ? n * 64 : 0;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float4 a[D][NT], b[D][MT];
  // one operand load of slice c into stage s: pieces 0..NT-1 = weight fragments, NT..NT+MT-1 = pixel sub-tiles
  auto load_piece = [&](int s, int c, int i) {           // unconditional (c is clamped by the caller)
    if (i < NT) {
      a[s][i] = wl[(size_t)c * wslice + woff[i]];
    } else {
      const int m = i - NT;
      if constexpr (DUAL) {
        const bool first = c < p.nC16a;                      // wave-uniform
        const float* src = first ? p.in + (size_t)c * p.in_ss : p.in2 + (size_t)(c - p.nC16a) * p.in2_ss;
        b[s][m] = *reinterpret_cast<const float4*>(src + (first ? boff[m] : boff2[m]));
      } else {
        b[s][m] = *reinterpret_cast<const float4*>(p.in + boff[m] + (size_t)c * p.in_ss);
      }
    }
  };
  auto load = [&](int s, int c) {
#pragma unroll
    for (int i = 0; i < NT + MT; ++i) load_piece(s, c, i);
  };
  // the MFMAs of stage s (SCHED > 0: with the loads of slice cn into stage sn pinned between them)
  constexpr int G = SCHED & 15;                                        // MFMAs per load
  constexpr int K0 = (SCHED & 16) ? 4 * MT * NT - G * (NT + MT) : 0;   // first MFMA with a load in front: early or late in the slice
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
  const int last = p.nC16 - 1;
#pragma unroll
  for (int s = 0; s < D - 1; ++s) load(s, min(s, last));
  const int nfull = p.nC16 / D * D;
  for (int c0 = 0; c0 < nfull; c0 += D) {
#pragma unroll
    for (int u = 0; u < D; ++u) {                        // slice c0 + u lives in stage u (c0 is a multiple of D)
      if constexpr (SCHED == 0) load((u + D - 1) % D, min(c0 + u + D - 1, last));
      mma(u, true, (u + D - 1) % D, min(c0 + u + D - 1, last));
    }
  }
#pragma unroll
  for (int u = 0; u < D - 1; ++u) {                      // tail: nC16 % D slices, already (being) loaded
    if (nfull + u < p.nC16) mma(u, false, 0, 0);
  }

  // ---- epilogue: shift (+ residual) (activation) -> L16 channel slice --------------------------------------
  int ob[MT], rb[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    ob[m] = orow[m] >= 0 ? orow[m] * p.out_rs + ox16[m] + g * 4 : -1;
    rb[m] = max(orow[m], 0) * p.res_rs + ox16[m] + g * 4;
  }
  float4 sh[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) sh[n] = *reinterpret_cast<const float4*>(p.bias + min(nt0 + n, p.nT16 - 1) * 16 + g * 4);
  // residual loads are unconditional and never under a branch (hipcc then counts them: a load under a branch makes it
  // fall back to s_waitcnt vmcnt(0) before every store)
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
    if constexpr (HAS_RES) { if (n + 1 < NT) load_res(n + 1, rnext); }   // the next group's residuals are in flight during the stores
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

template <int MT, int NT, int SCHED>
int launch_s(int D, const G1Params& p, dim3 grid, int nthreads, hipStream_t stream) {
  const bool r = p.res != nullptr;
  if (D == 2 && r) hipLaunchKernelGGL((gemm1x1_kernel<MT, NT, 2, true, false, SCHED>), grid, dim3(nthreads), 0, stream, p);
  else if (D == 2) hipLaunchKernelGGL((gemm1x1_kernel<MT, NT, 2, false, false, SCHED>), grid, dim3(nthreads), 0, stream, p);
  else if (r) hipLaunchKernelGGL((gemm1x1_kernel<MT, NT, 3, true, false, SCHED>), grid, dim3(nthreads), 0, stream, p);
  else hipLaunchKernelGGL((gemm1x1_kernel<MT, NT, 3, false, false, SCHED>), grid, dim3(nthreads), 0, stream, p);
  POCO_HIP_CHECK(hipGetLastError());
  return POCO_OK;
}

// cfg.NI: 1 = hipcc's order; 2, 3, 4 = a load per 2, 4, 8 MFMAs from the start of the slice; 5, 6 = per 2, 4 at its end
constexpr int g1_sched(int NI) { return NI == 2 ? 2 : NI == 3 ? 4 : NI == 4 ? 8 : NI == 5 ? 16 + 2 : NI == 6 ? 16 + 4 : 0; }
template <int MT, int NT>
int launch_d(int D, int NI, const G1Params& p, dim3 grid, int nthreads, hipStream_t stream) {
  if (NI == 2) return launch_s<MT, NT, g1_sched(2)>(D, p, grid, nthreads, stream);
  if (NI == 3) return launch_s<MT, NT, g1_sched(3)>(D, p, grid, nthreads, stream);
  if (NI == 4) return launch_s<MT, NT, g1_sched(4)>(D, p, grid, nthreads, stream);
  if (NI == 5) return launch_s<MT, NT, g1_sched(5)>(D, p, grid, nthreads, stream);
  if (NI == 6) return launch_s<MT, NT, g1_sched(6)>(D, p, grid, nthreads, stream);
  return launch_s<MT, NT, 0>(D, p, grid, nthreads, stream);
}

bool tile_ok(int MT, int NT) {
  return (MT == 2 && NT == 4) || (MT == 4 && (NT == 2 || NT == 4)) || (MT == 7 && (NT == 2 || NT == 4)) || (MT == 8 && NT == 2);
}

}  // namespace

// cfg: {MT, NT, WM, WN, R = prefetch depth D (2|3), NI = load schedule (g1_sched), ALG = 6}
bool gemm1x1_cfg_valid(const ConvDesc& d, const ConvCfg& cfg) {
  const long P = (long)d.B * ((d.H - 1) / d.stride + 1) * ((d.W - 1) / d.stride + 1);
  return d.ks == 1 && (d.stride == 1 || d.stride == 2) && d.Cin % 16 == 0 && d.Cout % 16 == 0 && tile_ok(cfg.MT, cfg.NT) &&
         cfg.WM >= 1 && cfg.WN >= 1 && cfg.WM * cfg.WN <= 8 && (cfg.R == 2 || cfg.R == 3) && cfg.NI >= 1 && cfg.NI <= 6 && (g1_sched(cfg.NI) & 15) * (cfg.MT + cfg.NT) <= 4 * cfg.MT * cfg.NT && P < (1L << 27) &&
         (long)d.B * d.H * d.in_cs * d.W < (1L << 31) && P * std::max(d.out_cs, d.res_cs) < (1L << 31);
}

int gemm1x1_launch(const ConvDesc& d, const ConvCfg& cfg, hipStream_t stream) {
  if (!gemm1x1_cfg_valid(d, cfg)) {
    poco_set_error("gemm1x1: ALG 6 needs ks = 1, stride 1|2, (MT,NT) in {(2,4),(4,2),(4,4),(7,2),(7,4),(8,2)}, WM*WN <= 8, R (depth) 2|3, NI (load schedule) 1..6 with room for its loads in a slice");
    return POCO_ERR_ARG;
  }
  if ((d.in_cs | d.in_co | d.out_cs | d.out_co | d.res_cs | d.res_co) & 3) {
    poco_set_error("conv: channel strides/offsets must be multiples of 4");
    return POCO_ERR_ARG;
  }
  G1Params p{};
  p.H = d.H; p.W = d.W; p.stride = d.stride;
  p.Ho = (d.H - 1) / d.stride + 1; p.Wo = (d.W - 1) / d.stride + 1;
  p.in = d.in + l16_chan_off(d.in_co, d.W);
  p.res = d.res ? d.res + l16_chan_off(d.res_co, p.Wo) : nullptr;
  p.out = d.out + l16_chan_off(d.out_co, p.Wo);
  p.wfrag = reinterpret_cast<const float4*>(d.wfrag); p.bias = d.bias;
  p.P = d.B * p.Ho * p.Wo;
  p.nC16 = d.Cin / 16; p.nT16 = d.Cout / 16; p.WM = cfg.WM; p.WN = cfg.WN;
  p.in_rs = d.in_cs * d.W; p.in_ss = d.W * 16;
  p.res_rs = d.res_cs * p.Wo; p.out_rs = d.out_cs * p.Wo; p.out_ss = p.Wo * 16;
  p.act = d.act; p.res_after_act = d.res_after_act; p.relu_from = d.relu_from;
  p.dWo = make_fastdiv(p.Wo); p.dHo = make_fastdiv(p.Ho);
  const int mtiles = (p.P + 15) / 16;
  const dim3 grid((mtiles + cfg.WM * cfg.MT - 1) / (cfg.WM * cfg.MT), (p.nT16 + cfg.WN * cfg.NT - 1) / (cfg.WN * cfg.NT));
  const int nthreads = cfg.WM * cfg.WN * 64;
#define G1_CASE(mt, nt) if (cfg.MT == mt && cfg.NT == nt) return launch_d<mt, nt>(cfg.R, cfg.NI, p, grid, nthreads, stream);
  G1_CASE(2, 4) G1_CASE(4, 2) G1_CASE(4, 4) G1_CASE(7, 2) G1_CASE(7, 4) G1_CASE(8, 2)
#undef G1_CASE
  poco_set_error("gemm1x1: unsupported tile");
  return POCO_ERR_ARG;
}

// out = act( [Wa | Wb] . [a ; b(stride2)] + bias ): the K-concatenated form of  bn3(conv3(a)) + bn_d(conv_d(b))  for the
// stride-2 Bottlenecks of ResNet-50 (resnet.py:101-121, layer2-4 .0): a [B,Ho,Wo,Ca of a_cs] is conv2's output, b
// [B,H2,W2,Cb of b_cs] the block input, sampled at (2y, 2x).  wfrag: conv_pack_weights(ks = 1) over Ca + Cb input channels.
int launch_gemm1x1_dual(const float* a, int a_cs, int Ca, const float* b, int b_cs, int Cb, int H2, int W2, int stride2,
                        const float* wfrag, const float* bias, float* out, int out_cs, int Cout, int B, int Ho, int Wo, int act,
                        hipStream_t stream, int wave_layout) {
  if ((Ca | Cb | Cout) % 16 || Cout % 64 || (stride2 != 1 && stride2 != 2) || (long)B * Ho * Wo >= (1L << 27)) {
    poco_set_error("gemm1x1_dual: Ca, Cb multiples of 16, Cout a multiple of 64, stride 1|2");
    return POCO_ERR_ARG;
  }
  G1Params p{};
  p.in = a; p.in2 = b; p.res = nullptr; p.out = out;
  p.wfrag = reinterpret_cast<const float4*>(wfrag); p.bias = bias;
  p.H = Ho; p.W = Wo; p.Ho = Ho; p.Wo = Wo; p.stride = 1;
  p.P = B * Ho * Wo; p.nC16 = (Ca + Cb) / 16; p.nC16a = Ca / 16; p.nT16 = Cout / 16; p.WM = 1; p.WN = 1;
  p.in_rs = a_cs * Wo; p.in_ss = Wo * 16; p.in2_rs = b_cs * W2; p.in2_ss = W2 * 16; p.H2 = H2; p.stride2 = stride2;
  p.out_rs = out_cs * Wo; p.out_ss = Wo * 16; p.res_rs = p.out_rs;
  p.act = act; p.res_after_act = 0; p.relu_from = 0;
  p.dWo = make_fastdiv(Wo); p.dHo = make_fastdiv(Ho);
  const int mtiles = (p.P + 15) / 16;
  // tile 7x4, depth 2, one wave per block, default load schedule: the fastest of the wave layouts / schedules probed per
  // ResNet-50 stage at B = 64 (tools/dual_probe.sh builds with -DG1_DUAL_EXP to repeat that sweep)
  int WM = 1, WN = 1, NI = 1;
  if (wave_layout > 0) { WM = std::max(1, (wave_layout / 10) % 10); WN = std::max(1, wave_layout % 10); NI = wave_layout >= 100 ? wave_layout / 100 : 1; if (WM * WN > 8 || !(NI == 1 || (NI >= 3 && NI <= 6))) { WM = WN = NI = 1; } }
#ifdef G1_DUAL_EXP
  static const char* ov = getenv("POCO_G1_DUAL");         // "WM,WN,NI": timing probe, probe builds only
  if (ov) {
    int a = 1, b = 1, c = 1;
    if (sscanf(ov, "%d,%d,%d", &a, &b, &c) == 3 && a >= 1 && b >= 1 && a * b <= 8 && (c == 1 || (c >= 3 && c <= 6))) { WM = a; WN = b; NI = c; }
  }
#endif
  p.WM = WM; p.WN = WN;
  const dim3 grid((mtiles + 7 * WM - 1) / (7 * WM), (p.nT16 / 4 + WN - 1) / WN);
  const dim3 block(WM * WN * 64);
  switch (NI) {
    case 3: hipLaunchKernelGGL((gemm1x1_kernel<7, 4, 2, false, true, g1_sched(3)>), grid, block, 0, stream, p); break;
    case 4: hipLaunchKernelGGL((gemm1x1_kernel<7, 4, 2, false, true, g1_sched(4)>), grid, block, 0, stream, p); break;
    case 5: hipLaunchKernelGGL((gemm1x1_kernel<7, 4, 2, false, true, g1_sched(5)>), grid, block, 0, stream, p); break;
    case 6: hipLaunchKernelGGL((gemm1x1_kernel<7, 4, 2, false, true, g1_sched(6)>), grid, block, 0, stream, p); break;
    default: hipLaunchKernelGGL((gemm1x1_kernel<7, 4, 2, false, true>), grid, block, 0, stream, p);
  }
  POCO_HIP_CHECK(hipGetLastError());
  return POCO_OK;
}
