// 1x1 convolutions (stride 1 or 2) as a GEMM with coalesced global traffic — ALG 9.
//
// ALG 6 (gemm1x1.hip) loads the pixel operand straight into the MFMA B-operand lane order: lane (idx, g) holds
// channels 4g..4g+3 of pixel idx, so the 16 lanes of a quarter wave touch 16 different 64-byte runs and the texture
// addresser works four times as long on such a load as on a contiguous KiB (measured: the loads of a 1x1 GEMM alone take
// 2.3x as long in that order; the 16-byte stores of the epilogue have the same shape).  With eight waves per CU streaming
// seven such loads per 16 channels the addresser, not the MFMA pipe, sets the pace.
//
// Here every global access of a wave is a contiguous KiB: lane l moves 16 bytes of pixel l/4, channel quad l%4, of a
// 16-pixel x 16-channel tile.  The tile is turned into the MFMA lane order (and the output tile back) through a
// wave-private KiB of LDS per sub-tile: ds_write_b128 in the coalesced order, ds_read_b128 in the operand order.  The
// LDS pipe is otherwise idle in this kernel, its traffic is free for the MFMA pipe (tools/probes/coissue2.hip), and a
// wave only ever reads what it wrote itself, in program order - no barriers, the waves free-run as in ALG 6.
//
// Per 16-channel slice and 16-pixel sub-tile m of a wave:  4*NT MFMAs on b[m];  ds_write g[m] (slice s+1, fetched one
// slice ago) over the LDS copy of slice s, which is dead: b[m] is in registers;  global load g[m] <- slice s+2;
// ds_read b[m] <- slice s+1.  The weights (A operand, packed fragments of conv_pack_weights, ks = 1) come straight from
// global memory / L2 as in ALG 6, double-buffered in registers.
//
// LDS position of (pixel p, quad q) inside a tile: float4 index p*4 + ((p >> 2) ^ c(q)), c = (0, 3, 1, 2): conflict-free
// for the coalesced order (8 consecutive lanes = 2 pixels x 4 quads) and for the operand order (each 16-lane group of
// a ds_read_b128 sees every (p & 3, slot) pair once).
#include "conv_mfma_types.h"

namespace {

struct G1TParams {
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
};

__device__ __forceinline__ int g1t_slot(int p, int q) {          // see the header: float4 index of (pixel p, quad q) in a tile
  const int c = (0x9C >> (2 * q)) & 3;                           // c(q) = 0, 3, 1, 2
  return p * 4 + (((p >> 2) ^ c) & 3);
}

template <int MT, int NT, bool HAS_RES>
__global__ void __launch_bounds__(512)
gemm1x1t_kernel(const G1TParams p) {
  extern __shared__ f32x4 lds4[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave % p.WM, wn = wave / p.WM;
  const int idx = lane & 15, g = lane >> 4;               // operand order: pixel idx, channel quad g
  const int pc = lane >> 2, qc = lane & 3;                // coalesced order: pixel pc, channel quad qc
  const int mt0 = (blockIdx.x * p.WM + wm) * MT;          // first 16-pixel sub-tile of this wave
  const int nt0 = (blockIdx.y * p.WN + wn) * NT;          // first 16-channel tile of this wave
  if (nt0 >= p.nT16 || mt0 * 16 >= p.P) return;           // wave-uniform; there are no barriers in this kernel
  f32x4* my = lds4 + wave * (MT * 64);                    // this wave's MT tiles
  const int wpos = g1t_slot(pc, qc), rpos = g1t_slot(idx, g);

  int boff[MT];      // float offset of this lane's 16 bytes (slice 0) in the input, coalesced order
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const uint32_t pix = (uint32_t)min((mt0 + m) * 16 + pc, p.P - 1);   // dead lanes re-read the last pixel
    const uint32_t row = fdiv(pix, p.dWo);
    const uint32_t x = pix - row * p.Wo;
    uint32_t irow = row, ix = x;
    if (p.stride == 2) {
      const uint32_t b = fdiv(row, p.dHo);
      irow = b * p.H + (row - b * p.Ho) * 2;
      ix = x * 2;
    }
    boff[m] = (int)(irow * (uint32_t)p.in_rs + ix * 16u) + 4 * qc;
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

  const int last = p.nC16 - 1;
  float4 a[2][NT];
  f32x4 b[MT], gl[MT];    // (ext-vector types: hipcc keeps arrays of the float4 struct that are copied whole in scratch)
#define G1T_LOAD_G(m, c) gl[m] = *reinterpret_cast<const f32x4*>(p.in + boff[m] + (size_t)(c) * p.in_ss)
#define G1T_LOAD_A(s, n, c) a[s][n] = wl[(size_t)(c) * wslice + woff[n]]

  // prologue: slice 0 -> LDS -> b; slice 1 in flight in gl; weights of slice 0
#pragma unroll
  for (int m = 0; m < MT; ++m) G1T_LOAD_G(m, 0);
#pragma unroll
  for (int n = 0; n < NT; ++n) G1T_LOAD_A(0, n, 0);
#pragma unroll
  for (int m = 0; m < MT; ++m) my[m * 64 + wpos] = gl[m];
#pragma unroll
  for (int m = 0; m < MT; ++m) G1T_LOAD_G(m, min(1, last));
#pragma unroll
  for (int m = 0; m < MT; ++m) b[m] = my[m * 64 + rpos];

  auto slice = [&](int st, int c) __attribute__((always_inline)) {                      // MFMAs of slice c (weights in a[st]); operands of c+1 / c+2 on their way
    const int c1 = min(c + 1, last), c2 = min(c + 2, last);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      if (m < NT) G1T_LOAD_A(st ^ 1, m, c1);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float bj = b[m][j];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const float wj = (j == 0) ? a[st][n].x : (j == 1) ? a[st][n].y : (j == 2) ? a[st][n].z : a[st][n].w;
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wj, bj, acc[m][n], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      my[m * 64 + wpos] = gl[m];                         // slice c+1 over the dead LDS copy of slice c
      G1T_LOAD_G(m, c2);
      b[m] = my[m * 64 + rpos];
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (MT < NT) {
#pragma unroll
      for (int n = MT; n < NT; ++n) G1T_LOAD_A(st ^ 1, n, c1);
    }
  };
  int c = 0;
  for (; c + 1 < p.nC16; c += 2) { slice(0, c); slice(1, c + 1); }
  if (c < p.nC16) slice(0, c);
#undef G1T_LOAD_G
#undef G1T_LOAD_A

  // ---- epilogue: one n-tile at a time through this wave's LDS tiles into the coalesced order, then shift (+ residual)
  // (activation) and contiguous KiB stores ---------------------------------------------------------------------------
  int ob[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int pix = (mt0 + m) * 16 + pc;
    const uint32_t row = fdiv((uint32_t)min(pix, p.P - 1), p.dWo);
    const int x16 = (min(pix, p.P - 1) - (int)row * p.Wo) * 16;
    ob[m] = pix < p.P ? (int)row : -1;                   // output row (b*Ho + y) or -1 ...
    boff[m] = x16 + qc * 4;                              // ... and the offset inside it (the input offsets are dead now)
  }
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const bool nok = nt0 + n < p.nT16;
    const int nn = min(nt0 + n, p.nT16 - 1);
    const float4 sh = *reinterpret_cast<const float4*>(p.bias + nn * 16 + qc * 4);
    const int co = nn * 16 + qc * 4;
    float4 r[MT];
    if constexpr (HAS_RES) {
#pragma unroll
      for (int m = 0; m < MT; ++m) r[m] = *reinterpret_cast<const float4*>(p.res + max(ob[m], 0) * p.res_rs + boff[m] + nn * p.out_ss);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) my[m * 64 + rpos] = acc[m][n];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      f32x4 v = my[m * 64 + wpos];
      v[0] += sh.x; v[1] += sh.y; v[2] += sh.z; v[3] += sh.w;
      if constexpr (HAS_RES) { if (!p.res_after_act) { v[0] += r[m].x; v[1] += r[m].y; v[2] += r[m].z; v[3] += r[m].w; } }
      if (p.act == 1 || (p.act == 3 && co >= p.relu_from)) {
        v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
      } else if (p.act == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = 1.f / (1.f + __expf(-v[e]));
      }
      if constexpr (HAS_RES) { if (p.res_after_act) { v[0] += r[m].x; v[1] += r[m].y; v[2] += r[m].z; v[3] += r[m].w; } }
      if (nok && ob[m] >= 0)
        *reinterpret_cast<float4*>(p.out + ob[m] * p.out_rs + boff[m] + nn * p.out_ss) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

template <int MT, int NT>
int launch_t(const G1TParams& p, dim3 grid, int nthreads, hipStream_t stream) {
  const size_t lds = (size_t)(nthreads / 64) * MT * 64 * sizeof(float4);
  if (p.res) hipLaunchKernelGGL((gemm1x1t_kernel<MT, NT, true>), grid, dim3(nthreads), lds, stream, p);
  else hipLaunchKernelGGL((gemm1x1t_kernel<MT, NT, false>), grid, dim3(nthreads), lds, stream, p);
  POCO_HIP_CHECK(hipGetLastError());
  return POCO_OK;
}

bool tile_ok_t(int MT, int NT) { return (MT == 7 && (NT == 2 || NT == 4)) || (MT == 4 && NT == 4) || (MT == 8 && NT == 2); }

}  // namespace

// cfg: {MT, NT, WM, WN, R = 1, NI = 1, ALG = 9}
bool gemm1x1t_cfg_valid(const ConvDesc& d, const ConvCfg& cfg) {
  const long P = (long)d.B * ((d.H - 1) / d.stride + 1) * ((d.W - 1) / d.stride + 1);
  return d.ks == 1 && (d.stride == 1 || d.stride == 2) && d.Cin % 16 == 0 && d.Cout % 16 == 0 && tile_ok_t(cfg.MT, cfg.NT) &&
         cfg.WM >= 1 && cfg.WN >= 1 && cfg.WM * cfg.WN <= 8 && cfg.R == 1 && cfg.NI == 1 && P < (1L << 27) &&
         (long)d.B * d.H * d.in_cs * d.W < (1L << 31) && P * std::max(d.out_cs, d.res_cs) < (1L << 31);
}

size_t gemm1x1t_lds_bytes(const ConvDesc& d, const ConvCfg& cfg) {
  return gemm1x1t_cfg_valid(d, cfg) ? (size_t)cfg.WM * cfg.WN * cfg.MT * 64 * sizeof(float4) : 0;
}

int gemm1x1t_launch(const ConvDesc& d, const ConvCfg& cfg, hipStream_t stream) {
  if (!gemm1x1t_cfg_valid(d, cfg)) {
    poco_set_error("gemm1x1t: ALG 9 needs ks = 1, stride 1|2, (MT,NT) in {(4,4),(7,2),(7,4),(8,2)}, WM*WN <= 8, R = NI = 1");
    return POCO_ERR_ARG;
  }
  if ((d.in_cs | d.in_co | d.out_cs | d.out_co | d.res_cs | d.res_co) & 3) {
    poco_set_error("conv: channel strides/offsets must be multiples of 4");
    return POCO_ERR_ARG;
  }
  G1TParams p{};
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
#define G1T_CASE(mt, nt) if (cfg.MT == mt && cfg.NT == nt) return launch_t<mt, nt>(p, grid, nthreads, stream);
  G1T_CASE(4, 4) G1T_CASE(7, 2) G1T_CASE(7, 4) G1T_CASE(8, 2)
#undef G1T_CASE
  poco_set_error("gemm1x1t: unsupported tile");
  return POCO_ERR_ARG;
}
