// EXPERIMENT (round 3; never the default, never the headline): 1x1 convolutions as a register-direct GEMM
// in SPLIT fp16 - ALG 12.  Every fp32 operand is written as hi + lo with hi = fp16(x), lo = fp16(x - hi) (22 mantissa bits
// together) and a product is three MFMAs of v_mfma_f32_16x16x32_f16 (hi*hi + hi*lo + lo*hi, fp32 accumulation; lo*lo ~ 2^-22
// is dropped).  That instruction contracts K = 32 in 16 clk where v_mfma_f32_16x16x4_f32 contracts K = 4 in 32 clk: 3 MFMAs per
// 32 channels = 48 clk against 256 clk.  Weights are split (and scaled by 2^10 so that the lo halves stay normal fp16 numbers;
// the accumulator is scaled back, exactly) on the host; activations are split in registers right after the load.
// Same operand roles, K walk and epilogue as gemm1x1.hip (ALG 6).  Lane layout of the K = 32 MFMA (tools/probes/
// mfma_f16_probe.hip): lane (r = l % 16, q = l / 16) holds the 8 consecutive k = 8 q .. 8 q + 7 of row / column r, i.e. for the
// pixel operand the channels 8 q .. 8 q + 7 of a 32-channel slice = 32 contiguous bytes of fp32 in the L16 layout.
// Enabled only by POCO_SPLIT_F16=1 (bench.py --split-f16) for the plain 1x1 convs with Cin % 32 == 0; gated by the stress
// fixtures at 1e-3 (tests/test_model_gpu.py::test_split_f16_experiment_passes_the_gate).
#include "conv_mfma_types.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

struct GhParams {
  const float* in;
  const float* res;
  float* out;
  const float4* wfrag;   // [Cin/32][Cout16/16][64 lanes][2]: 8 hi halves | 8 lo halves of 2^10 * W[co][32 c + 8 q + j]
  const float* bias;
  int P, H, W, Ho, Wo, stride;
  int nC32, nT16, WM, WN;
  int in_rs, in_ss, res_rs, out_rs, out_ss;
  int act, res_after_act, relu_from;
  int dbg;               // GH_EXP probe builds only: 1 no MFMAs, 2 no weight loads after the first slices, 4 no pixel loads / staging, 8 no stores
  FastDiv dWo, dHo;
};
#ifndef GH_EXP
#define GH_EXP 0
#endif

__device__ __forceinline__ void split8(const float4& x0, const float4& x1, h8& hi, h8& lo) {
  const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const _Float16 h = (_Float16)x[j];
    hi[j] = h;
    lo[j] = (_Float16)(x[j] - (float)h);
  }
}

template <int MT, int NT, bool HAS_RES>
__global__ void __launch_bounds__(512)
gemm1x1h_kernel(const GhParams p) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave % p.WM, wn = wave / p.WM;
  const int idx = lane & 15, g = lane >> 4;
  const int mt0 = (blockIdx.x * p.WM + wm) * MT;
  const int nt0 = (blockIdx.y * p.WN + wn) * NT;
  if (nt0 >= p.nT16 || mt0 * 16 >= p.P) return;

  int boff[MT], orow[MT], ox16[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int pix = (mt0 + m) * 16 + idx;
    const uint32_t pc = (uint32_t)min(pix, p.P - 1);
    const uint32_t row = fdiv(pc, p.dWo);
    const uint32_t x = pc - row * p.Wo;
    uint32_t irow = row, ix = x;
    if (p.stride == 2) {
      const uint32_t b = fdiv(row, p.dHo);
      irow = b * p.H + (row - b * p.Ho) * 2;
      ix = x * 2;
    }
    // channels 8 g .. 8 g + 7 of a 32-channel slice: 16-channel slice (g >> 1) of the pair, floats (g & 1) * 8 .. + 7
    boff[m] = (int)(irow * (uint32_t)p.in_rs + ix * 16u) + (g >> 1) * p.in_ss + (g & 1) * 8;
    orow[m] = pix < p.P ? (int)row : -1;
    ox16[m] = (int)x * 16;
  }
  const float4* wl = p.wfrag + ((size_t)nt0 * 64 + lane) * 2;
  const int wslice = p.nT16 * 128;                       // float4 per K = 32 slice
  int woff[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) woff[n] = (nt0 + n < p.nT16) ? n * 128 : 0;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float4 a[2][NT][2], b[2][MT][2];                      // two stages: the next slice's raw operands are in flight
  auto load = [&](int s, int c) {
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      a[s][n][0] = wl[(size_t)c * wslice + woff[n]];
      a[s][n][1] = wl[(size_t)c * wslice + woff[n] + 1];
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float* src = p.in + boff[m] + (size_t)c * 2 * p.in_ss;
      b[s][m][0] = *reinterpret_cast<const float4*>(src);
      b[s][m][1] = *reinterpret_cast<const float4*>(src + 4);
    }
  };
  auto mma = [&](int s) {
    h8 bh[MT], bl[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) split8(b[s][m][0], b[s][m][1], bh[m], bl[m]);
    // small terms first; the three partial products of a tile MT*NT MFMAs apart (no dependent chain on one accumulator)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const h8 ah = __builtin_bit_cast(h8, a[s][n][0]), al = __builtin_bit_cast(h8, a[s][n][1]);
#pragma unroll
        for (int m = 0; m < MT; ++m)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(t == 0 ? al : ah, t == 1 ? bl[m] : bh[m], acc[m][n], 0, 0, 0);
      }
  };
  const int last = p.nC32 - 1;
  load(0, 0);
  for (int c0 = 0; c0 + 1 < p.nC32; c0 += 2) {
    load(1, c0 + 1);
    mma(0);
    load(0, min(c0 + 2, last));
    mma(1);
  }
  if (p.nC32 & 1) mma(0);

  // ---- epilogue (as gemm1x1.hip): 2^-10 * acc + shift (+ residual) (activation) -> L16 channel slice ---------------------
  int ob[MT], rb[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    ob[m] = orow[m] >= 0 ? orow[m] * p.out_rs + ox16[m] + g * 4 : -1;
    rb[m] = max(orow[m], 0) * p.res_rs + ox16[m] + g * 4;
  }
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const float4 sh = *reinterpret_cast<const float4*>(p.bias + min(nt0 + n, p.nT16 - 1) * 16 + g * 4);
    const int co = (nt0 + n) * 16 + g * 4;
    const bool nok = nt0 + n < p.nT16;
    float4 r[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      r[m] = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (HAS_RES) r[m] = *reinterpret_cast<const float4*>(p.res + rb[m] + min(nt0 + n, p.nT16 - 1) * p.out_ss);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      f32x4 v = acc[m][n] * 0.0009765625f;
      v[0] += sh.x; v[1] += sh.y; v[2] += sh.z; v[3] += sh.w;
      if (!p.res_after_act) { v[0] += r[m].x; v[1] += r[m].y; v[2] += r[m].z; v[3] += r[m].w; }
      if (p.act == 1 || (p.act == 3 && co >= p.relu_from)) {
        v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
      } else if (p.act == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = 1.f / (1.f + __expf(-v[e]));
      }
      if (p.res_after_act) { v[0] += r[m].x; v[1] += r[m].y; v[2] += r[m].z; v[3] += r[m].w; }
      if (nok && ob[m] >= 0)
        *reinterpret_cast<float4*>(p.out + ob[m] + (nt0 + n) * p.out_ss) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

// ---- LDS-tiled version: a block of 4 waves (2 x 2) owns 128 pixels x 128 output channels ------------------------------------------
// The register-direct kernel above converts every pixel operand once per WAVE tile and issues 16 operand loads per 48 MFMAs: with
// the MFMA time cut to 1/5 it is bound by the vector-memory path and the conversion VALU (measured: only ~25 % faster than fp32).
// Here the 128 x 32 pixel slice is loaded ONCE per block with coalesced float4 loads (2 x 32 B per thread), split into hi / lo
// halves by the loading thread and written to LDS in the lane order of the MFMA B operand ([sub-tile][hi|lo][lane] x 16 B: a wave
// reads its operand with one conflict-free ds_read_b128); the weight fragments (already hi / lo, in operand order) go global ->
// registers one slice ahead.  One barrier per 32-channel slice = per 48 MFMAs of a wave.
constexpr int HB_PIX = 128, HB_NT = 8;                   // block tile: 8 pixel sub-tiles x 8 n-tiles; wave tile 4 x 4

template <bool HAS_RES>
__global__ void __launch_bounds__(256, 2)
gemm1x1h_tiled_kernel(const GhParams p) {
  __shared__ float4 bs[2][8 * 2 * 64];                   // [stage][sub-tile m][hi | lo][lane]: 2 x 16 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int idx = lane & 15, g = lane >> 4;
  const int pix0 = blockIdx.x * HB_PIX;
  const int nt0 = blockIdx.y * HB_NT + wn * 4;           // first n-tile of this wave

  // loader role: pairs q = tid, tid + 256: pixel q & 127 of the block, channel octet q >> 7 of the 32-channel slice
  int lboff[2], lslot[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int q = tid + 256 * r, pl = q & 127, og = q >> 7;
    const uint32_t pc = (uint32_t)min(pix0 + pl, p.P - 1);
    const uint32_t row = fdiv(pc, p.dWo);
    const uint32_t x = pc - row * p.Wo;
    uint32_t irow = row, ix = x;
    if (p.stride == 2) {
      const uint32_t b = fdiv(row, p.dHo);
      irow = b * p.H + (row - b * p.Ho) * 2;
      ix = x * 2;
    }
    lboff[r] = (int)(irow * (uint32_t)p.in_rs + ix * 16u) + (og >> 1) * p.in_ss + (og & 1) * 8;
    lslot[r] = ((pl >> 4) * 2) * 64 + og * 16 + (pl & 15);        // hi slot; lo = + 64
  }
  const float4* wl = p.wfrag + ((size_t)min(nt0, p.nT16 - 1) * 64 + lane) * 2;     // (a wave beyond Cout computes along on the last tile: no early exit, barriers)
  const int wslice = p.nT16 * 128;
  int woff[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) woff[n] = (nt0 + n < p.nT16) ? n * 128 : 0;

  f32x4 acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Register ring of 3 slices for the raw operands: with the MFMA time of a slice at ~1/5 of the fp32 kernel's, ONE slice of compute
  // no longer covers an L2 / HBM round trip (the first version, one slice ahead, ran at 1/4 of its MFMA bound: every iteration stalled
  // on its loads).  Slice c lives in ring slot c % 3; its loads are issued right after the MFMAs of slice c - 3 released the slot.
  float4 ar[3][4][2], br[3][2][2];
  auto load_b = [&](int c, float4 (&b)[2][2]) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const float* src = p.in + lboff[r] + (size_t)c * 2 * p.in_ss;
      b[r][0] = *reinterpret_cast<const float4*>(src);
      b[r][1] = *reinterpret_cast<const float4*>(src + 4);
    }
  };
  auto load_a = [&](int c, float4 (&a)[4][2]) {
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      a[n][0] = wl[(size_t)c * wslice + woff[n]];
      a[n][1] = wl[(size_t)c * wslice + woff[n] + 1];
    }
  };
  auto stage_b = [&](int st, const float4 (&b)[2][2]) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      h8 hi, lo;
      split8(b[r][0], b[r][1], hi, lo);
      bs[st][lslot[r]] = __builtin_bit_cast(float4, hi);
      bs[st][lslot[r] + 64] = __builtin_bit_cast(float4, lo);
    }
  };
  const int last = p.nC32 - 1;
#pragma unroll
  for (int u = 0; u < 3; ++u) { load_a(min(u, last), ar[u]); load_b(min(u, last), br[u]); }
  stage_b(0, br[0]);
  __syncthreads();
  for (int c0 = 0; c0 < p.nC32; c0 += 3) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int c = c0 + u;
      if (c < p.nC32) {                                   // block-uniform
        const int st = c & 1;
        // the three partial products of a tile are issued 16 MFMAs apart: back to back they would form a dependent chain on one
        // accumulator and the pipe would wait out the full latency of every one of them
        h8 bh[4], bl[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          bh[m] = __builtin_bit_cast(h8, bs[st][((wm * 4 + m) * 2) * 64 + lane]);
          bl[m] = __builtin_bit_cast(h8, bs[st][((wm * 4 + m) * 2 + 1) * 64 + lane]);
        }
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
              const h8 ah = __builtin_bit_cast(h8, ar[u][n][0]), al = __builtin_bit_cast(h8, ar[u][n][1]);
              if (!(GH_EXP && (p.dbg & 1)))
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(t == 0 ? al : ah, t == 1 ? bl[m] : bh[m], acc[m][n], 0, 0, 0);
              else acc[m][n][0] += (float)al[0] + (float)bl[m][0];
            }
        if (c + 1 < p.nC32 && !(GH_EXP && (p.dbg & 4))) stage_b(st ^ 1, br[(u + 1) % 3]);     // slice c + 1: loaded two iterations ago; the stage was last read before the previous barrier
        if (c + 3 < p.nC32) {
          if (!(GH_EXP && (p.dbg & 2))) load_a(c + 3, ar[u]);
          if (!(GH_EXP && (p.dbg & 4))) load_b(c + 3, br[u]);
        }
        __syncthreads();
      }
    }
  }

  // ---- epilogue: this wave's 4 x 4 tiles ------------------------------------------------------------------------------------------
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int pix = pix0 + (wm * 4 + m) * 16 + idx;
    const uint32_t pc = (uint32_t)min(pix, p.P - 1);
    const uint32_t row = fdiv(pc, p.dWo);
    const int ox16 = (int)(pc - row * p.Wo) * 16;
    const bool pok = pix < p.P;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int ntc = min(nt0 + n, p.nT16 - 1);
      const float4 sh = *reinterpret_cast<const float4*>(p.bias + ntc * 16 + g * 4);
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (HAS_RES) r = *reinterpret_cast<const float4*>(p.res + (size_t)row * p.res_rs + (size_t)ntc * p.out_ss + ox16 + g * 4);
      f32x4 v = acc[m][n] * 0.0009765625f;
      v[0] += sh.x; v[1] += sh.y; v[2] += sh.z; v[3] += sh.w;
      if (!p.res_after_act) { v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w; }
      const int co = (nt0 + n) * 16 + g * 4;
      if (p.act == 1 || (p.act == 3 && co >= p.relu_from)) {
        v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
      } else if (p.act == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = 1.f / (1.f + __expf(-v[e]));
      }
      if (p.res_after_act) { v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w; }
      if (pok && nt0 + n < p.nT16 && !(GH_EXP && (p.dbg & 8)))
        *reinterpret_cast<float4*>(p.out + (size_t)row * p.out_rs + (size_t)(nt0 + n) * p.out_ss + ox16 + g * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

template <int MT, int NT>
int launch_t(const GhParams& p, dim3 grid, int nthreads, hipStream_t stream) {
  if (p.res) hipLaunchKernelGGL((gemm1x1h_kernel<MT, NT, true>), grid, dim3(nthreads), 0, stream, p);
  else hipLaunchKernelGGL((gemm1x1h_kernel<MT, NT, false>), grid, dim3(nthreads), 0, stream, p);
  POCO_HIP_CHECK(hipGetLastError());
  return POCO_OK;
}

}  // namespace

size_t gemm1x1h_packed_floats(int Cin, int Cout16) { return (size_t)Cin * Cout16; }   // 2 halves (hi, lo) per weight = 4 bytes

// dst: [Cin/32][Cout16/16][64 lanes][16 halves]: lane (co_l = l % 16, q = l / 16): 8 hi | 8 lo of 2^10 * scale[co] * W[co][32 c + 8 q + j]
void gemm1x1h_pack_weights(const float* w_oi, const float* scale, int Cout, int Cin, int Cout16, float* dst) {
  _Float16* d = reinterpret_cast<_Float16*>(dst);
  const int nC32 = Cin / 32, nT16 = Cout16 / 16;
  for (int c = 0; c < nC32; ++c)
    for (int nt = 0; nt < nT16; ++nt)
      for (int lane = 0; lane < 64; ++lane) {
        const int co = nt * 16 + (lane & 15), q = lane >> 4;
        _Float16* o = d + (((size_t)c * nT16 + nt) * 64 + lane) * 16;
        for (int j = 0; j < 8; ++j) {
          float v = 0.f;
          if (co < Cout) v = (float)((double)w_oi[(size_t)co * Cin + 32 * c + 8 * q + j] * (scale ? (double)scale[co] : 1.0) * 1024.0);
          const _Float16 h = (_Float16)v;
          o[j] = h;
          o[8 + j] = (_Float16)(v - (float)h);
        }
      }
}

// cfg: {MT, NT in {(4,4),(4,2),(2,4),(2,2)}, WM, WN, R = 8 selects the LDS-tiled kernel (128 x 128 block tiles; MT..WN ignored), ALG = 12}
bool gemm1x1h_cfg_valid(const ConvDesc& d, const ConvCfg& cfg) {
  const long P = (long)d.B * ((d.H - 1) / d.stride + 1) * ((d.W - 1) / d.stride + 1);
  const bool tile = (cfg.MT == 4 || cfg.MT == 2) && (cfg.NT == 4 || cfg.NT == 2);
  return d.ks == 1 && (d.stride == 1 || d.stride == 2) && d.Cin % 32 == 0 && d.Cout % 16 == 0 && tile && cfg.WM >= 1 && cfg.WN >= 1 &&
         cfg.WM * cfg.WN <= 8 && P < (1L << 27) && (long)d.B * d.H * d.in_cs * d.W < (1L << 31) &&
         P * std::max(d.out_cs, d.res_cs) < (1L << 31);
}

int gemm1x1h_launch(const ConvDesc& d, const ConvCfg& cfg, hipStream_t stream) {
  if (!gemm1x1h_cfg_valid(d, cfg) || !d.wfrag_h) {
    poco_set_error("gemm1x1h (split-fp16 experiment): ALG 12 needs ks = 1, Cin % 32 == 0, (MT,NT) in {2,4}x{2,4}, WM*WN <= 8 and its hi/lo weight fragments");
    return POCO_ERR_ARG;
  }
  if ((d.in_cs | d.in_co | d.out_cs | d.out_co | d.res_cs | d.res_co) & 15) {
    poco_set_error("gemm1x1h: channel strides/offsets must be multiples of 16");
    return POCO_ERR_ARG;
  }
  if (d.in_co & 31) { poco_set_error("gemm1x1h: input channel offset must be a multiple of 32"); return POCO_ERR_ARG; }
  GhParams p{};
  p.H = d.H; p.W = d.W; p.stride = d.stride;
  p.Ho = (d.H - 1) / d.stride + 1; p.Wo = (d.W - 1) / d.stride + 1;
  p.in = d.in + l16_chan_off(d.in_co, d.W);
  p.res = d.res ? d.res + l16_chan_off(d.res_co, p.Wo) : nullptr;
  p.out = d.out + l16_chan_off(d.out_co, p.Wo);
  p.wfrag = reinterpret_cast<const float4*>(d.wfrag_h); p.bias = d.bias;
  p.P = d.B * p.Ho * p.Wo; p.nC32 = d.Cin / 32; p.nT16 = d.Cout / 16; p.WM = cfg.WM; p.WN = cfg.WN;
  p.in_rs = d.in_cs * d.W; p.in_ss = d.W * 16;
  p.res_rs = d.res_cs * p.Wo; p.out_rs = d.out_cs * p.Wo; p.out_ss = p.Wo * 16;
  p.act = d.act; p.res_after_act = d.res_after_act; p.relu_from = d.relu_from;
  p.dWo = make_fastdiv(p.Wo); p.dHo = make_fastdiv(p.Ho);
#if GH_EXP
  { const char* e = getenv("POCO_GH_DBG"); p.dbg = e ? atoi(e) : 0; }
#endif
  const int mtiles = (p.P + 15) / 16;
  if (cfg.R == 8) {                                      // R = 8: the LDS-tiled kernel (128 pixels x 128 channels per block of 4 waves)
    const dim3 tg((p.P + HB_PIX - 1) / HB_PIX, (p.nT16 + HB_NT - 1) / HB_NT);
    if (p.res) hipLaunchKernelGGL(gemm1x1h_tiled_kernel<true>, tg, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(gemm1x1h_tiled_kernel<false>, tg, dim3(256), 0, stream, p);
    POCO_HIP_CHECK(hipGetLastError());
    return POCO_OK;
  }
  const dim3 grid((mtiles + cfg.MT * cfg.WM - 1) / (cfg.MT * cfg.WM), (p.nT16 + cfg.NT * cfg.WN - 1) / (cfg.NT * cfg.WN));
  const int nthreads = cfg.WM * cfg.WN * 64;
  if (cfg.MT == 4 && cfg.NT == 4) return launch_t<4, 4>(p, grid, nthreads, stream);
  if (cfg.MT == 4 && cfg.NT == 2) return launch_t<4, 2>(p, grid, nthreads, stream);
  if (cfg.MT == 2 && cfg.NT == 4) return launch_t<2, 4>(p, grid, nthreads, stream);
  return launch_t<2, 2>(p, grid, nthreads, stream);
}
