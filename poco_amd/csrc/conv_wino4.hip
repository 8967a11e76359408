// ALG 7 (experimental, stand-alone operator only): Winograd F(4x4,3x3) on the fp32 MFMA.
//
// 36 position GEMMs M_xi[co][tile] += U_xi[co][ci] V_xi[ci][tile] per 4x4 output tile = 2.25 MFMA-MACs per output
// pixel and input channel instead of 4 for F(2x2,3x3) (ALG 3/4) and 9 for the direct conv: the lever beyond the MFMA
// pipe rate for the 56x56 / 28x28 BasicBlock convs.  This first version fixes the structure and the
// numerics; it is NOT tuned (no LDS staging of the input, single-buffered U, two barriers per 16-channel slice) and is
// not selected by the tuning table.
//
//   V = B^T d B (6x6 window d),  M = sum_ci U .* V,  Y = A^T M A (4x4 outputs),  U = G g G^T (host, float64)
//
// Work split: a "tile group" is 16 tiles (the MFMA pixel dimension); three waves share a group and own two rows of
// the 6x6 position grid each (12 positions x NT n-tiles = 36*NT accumulator VGPRs... x4).  Every wave reads the whole
// 6x6 window of its lanes' tiles (column by column, straight from global memory in the L16 layout) but forms only its
// two rows of B^T d, then its 12 entries of V.  After the K loop a wave reduces its rows along the columns
// (Z = M A, 2 x 4 values), the three waves exchange Z through LDS, and wave r finishes n-tile r: Y = A^T Z, bias,
// residual, activation, 16 pixels x 4 channels per lane.
#include "conv_mfma_types.h"

namespace {

struct W4Params {
  const float* in;
  const float* res;
  float* out;
  const float4* ufrag;   // [36 positions][Cin/16][Cout16/16][64] float4 (conv_pack_weights, ks = 6)
  const float* bias;
  int B, H, W, nC16, nT16;
  int in_rs, in_ss, res_rs, out_rs, out_ss;
  int TX, TY, ntiles;    // tiles per row / column of an image, total tiles
  int G;                 // tile groups per block (block = 3 * G waves)
  int act, res_after_act;
  FastDiv dTX, dTY;
};

// rows of B^T (input transform) and of A^T (output transform) of F(4x4,3x3) [Lavin & Gray 2016]
template <int R>
__device__ __forceinline__ float4 bt_row(const float4& d0, const float4& d1, const float4& d2, const float4& d3, const float4& d4,
                                         const float4& d5) {
  auto c = [&](auto f) { return make_float4(f(d0.x, d1.x, d2.x, d3.x, d4.x, d5.x), f(d0.y, d1.y, d2.y, d3.y, d4.y, d5.y),
                                            f(d0.z, d1.z, d2.z, d3.z, d4.z, d5.z), f(d0.w, d1.w, d2.w, d3.w, d4.w, d5.w)); };
  if constexpr (R == 0) return c([](float a, float b, float cc, float d, float e, float f) { (void)b; (void)d; (void)f; return 4.f * a - 5.f * cc + e; });
  else if constexpr (R == 1) return c([](float a, float b, float cc, float d, float e, float f) { (void)a; (void)f; return -4.f * (b + cc) + d + e; });
  else if constexpr (R == 2) return c([](float a, float b, float cc, float d, float e, float f) { (void)a; (void)f; return 4.f * (b - cc) - d + e; });
  else if constexpr (R == 3) return c([](float a, float b, float cc, float d, float e, float f) { (void)a; (void)f; return 2.f * (d - b) - cc + e; });
  else if constexpr (R == 4) return c([](float a, float b, float cc, float d, float e, float f) { (void)a; (void)f; return 2.f * (b - d) - cc + e; });
  else return c([](float a, float b, float cc, float d, float e, float f) { (void)a; (void)cc; (void)e; return 4.f * b - 5.f * d + f; });
}
__device__ __forceinline__ float4 bt_row_rt(int r, const float4* t) {     // runtime row index, fully unrolled callers
  switch (r) {
    case 0: return bt_row<0>(t[0], t[1], t[2], t[3], t[4], t[5]);
    case 1: return bt_row<1>(t[0], t[1], t[2], t[3], t[4], t[5]);
    case 2: return bt_row<2>(t[0], t[1], t[2], t[3], t[4], t[5]);
    case 3: return bt_row<3>(t[0], t[1], t[2], t[3], t[4], t[5]);
    case 4: return bt_row<4>(t[0], t[1], t[2], t[3], t[4], t[5]);
    default: return bt_row<5>(t[0], t[1], t[2], t[3], t[4], t[5]);
  }
}
// A^T row i applied to six values m0..m5
__device__ __forceinline__ f32x4 at_row(int i, const f32x4& m0, const f32x4& m1, const f32x4& m2, const f32x4& m3, const f32x4& m4,
                                        const f32x4& m5) {
  switch (i) {
    case 0: return m0 + m1 + m2 + m3 + m4;
    case 1: return (m1 - m2) + 2.f * (m3 - m4);
    case 2: return (m1 + m2) + 4.f * (m3 + m4);
    default: return (m1 - m2) + 8.f * (m3 - m4) + m5;
  }
}

template <int NT, int PROW>
__device__ __forceinline__ void wino4_wave(const W4Params& p, float4* smem, int grp, int nt0, int lane, int wave, int nwaves) {
  const int idx = lane & 15, g = lane >> 4;
  // ---- this lane's tile ------------------------------------------------------------------------------------
  const int tile = (blockIdx.x * p.G + grp) * 16 + idx;
  const bool tvalid = tile < p.ntiles;
  const uint32_t tc = (uint32_t)min(tile, p.ntiles - 1);
  const uint32_t trow = fdiv(tc, p.dTX);                 // b * TY + ty
  const int tx = (int)(tc - trow * (uint32_t)p.TX);
  const uint32_t b = fdiv(trow, p.dTY);
  const int ty = (int)(trow - b * (uint32_t)p.TY);
  int rowoff[6], coloff[6];                              // float offsets of the window rows / columns, -1 = padding
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int iy = 4 * ty - 1 + k, ix = 4 * tx - 1 + k;
    rowoff[k] = (tvalid && (unsigned)iy < (unsigned)p.H) ? (int)((b * (uint32_t)p.H + (uint32_t)iy) * (uint32_t)p.in_rs) : -1;
    coloff[k] = ((unsigned)ix < (unsigned)p.W) ? ix * 16 + 4 * g : -1;
  }
  f32x4 acc[12][NT];
#pragma unroll
  for (int i = 0; i < 12; ++i)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nthreads = nwaves * 64;
  const int tid = wave * 64 + lane;
  for (int c = 0; c < p.nC16; ++c) {
    // ---- two rows of B^T d, column by column (overlaps the U copy of the other waves) --------------------------
    float4 t[2][6];
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      float4 d[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const bool ok = rowoff[k] >= 0 && coloff[s] >= 0;
        const float4 v = *reinterpret_cast<const float4*>(p.in + (ok ? rowoff[k] + coloff[s] : 0) + (size_t)c * p.in_ss);
        d[k] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      t[0][s] = bt_row<2 * PROW>(d[0], d[1], d[2], d[3], d[4], d[5]);
      t[1][s] = bt_row<2 * PROW + 1>(d[0], d[1], d[2], d[3], d[4], d[5]);
    }
    // ---- U fragments of slice c -> LDS: [36][NT][64] ----------------------------------------------------------
    __syncthreads();                                     // everybody is done with the previous slice's fragments
    for (int i = tid; i < 36 * NT * 64; i += nthreads) {
      const int pos = i / (NT * 64), rem = i - pos * (NT * 64), n = rem >> 6, l = rem & 63;
      smem[i] = p.ufrag[(((size_t)pos * p.nC16 + c) * p.nT16 + min(nt0 + n, p.nT16 - 1)) * 64 + l];
    }
    __syncthreads();
    // ---- V rows 2*PROW, 2*PROW+1 and their 12 position GEMMs ---------------------------------------------------
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
      for (int nu = 0; nu < 6; ++nu) {
        const float4 v = bt_row_rt(nu, t[r]);            // (X B)[r][nu] = sum_s X[r][s] B^T[nu][s]
        const float bv[4] = {v.x, v.y, v.z, v.w};
        const int pos = (2 * PROW + r) * 6 + nu;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const float4 a = smem[(pos * NT + n) * 64 + lane];
          const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[r * 6 + nu][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[j], acc[r * 6 + nu][n], 0, 0, 0);
        }
      }
    }
  }
  // ---- Z = M A for this wave's two rows; exchange; wave PROW finishes n-tile PROW -----------------------------------
  __syncthreads();                                       // U buffer is dead: reuse the LDS as [wave][n][2 rows][4 cols][64]
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 z = at_row(j, acc[r * 6 + 0][n], acc[r * 6 + 1][n], acc[r * 6 + 2][n], acc[r * 6 + 3][n], acc[r * 6 + 4][n],
                               acc[r * 6 + 5][n]);
        smem[(((wave * NT + n) * 2 + r) * 4 + j) * 64 + lane] = make_float4(z[0], z[1], z[2], z[3]);
      }
  __syncthreads();
  if (PROW >= NT || nt0 + PROW >= p.nT16) return;        // no barrier after this point
  constexpr int n = PROW < NT ? PROW : 0;
  f32x4 zf[6][4];
#pragma unroll
  for (int xi = 0; xi < 6; ++xi)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 z = smem[((((grp * 3 + xi / 2) * NT + n) * 2 + (xi & 1)) * 4 + j) * 64 + lane];
      zf[xi][j] = (f32x4){z.x, z.y, z.z, z.w};
    }
  const float4 sh = *reinterpret_cast<const float4*>(p.bias + (nt0 + n) * 16 + g * 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int oy = 4 * ty + i;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ox = 4 * tx + j;
      if (!tvalid || oy >= p.H || ox >= p.W) continue;
      f32x4 v = at_row(i, zf[0][j], zf[1][j], zf[2][j], zf[3][j], zf[4][j], zf[5][j]);
      v[0] += sh.x; v[1] += sh.y; v[2] += sh.z; v[3] += sh.w;
      const size_t orow = (size_t)b * p.H + oy;
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.res) r = *reinterpret_cast<const float4*>(p.res + orow * p.res_rs + (size_t)(nt0 + n) * p.out_ss + ox * 16 + g * 4);
      if (!p.res_after_act) { v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w; }
      if (p.act == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
      else if (p.act == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = 1.f / (1.f + __expf(-v[e]));
      }
      if (p.res_after_act) { v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w; }
      *reinterpret_cast<float4*>(p.out + orow * p.out_rs + (size_t)(nt0 + n) * p.out_ss + ox * 16 + g * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

template <int NT>
__global__ void __launch_bounds__(384)
conv_wino4_kernel(const W4Params p) {
  extern __shared__ float4 smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwaves = blockDim.x >> 6;
  const int grp = wave / 3, prow = wave - grp * 3;
  const int nt0 = blockIdx.y * NT;
  if (prow == 0) wino4_wave<NT, 0>(p, smem, grp, nt0, lane, wave, nwaves);
  else if (prow == 1) wino4_wave<NT, 1>(p, smem, grp, nt0, lane, wave, nwaves);
  else wino4_wave<NT, 2>(p, smem, grp, nt0, lane, wave, nwaves);
}

}  // namespace

// U = G g G^T per (co, ci), float64 on the host, as a [Cout][Cin][36] "36-tap" filter for conv_pack_weights(ks = 6)
void conv_wino4_transform_weights(const float* w_oihw, int Cout, int Cin, std::vector<float>* out) {
  static const double G[6][3] = {{1.0 / 4, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                 {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
  out->assign((size_t)Cout * Cin * 36, 0.f);
  for (size_t oc = 0; oc < (size_t)Cout * Cin; ++oc) {
    const float* gk = w_oihw + oc * 9;
    double t[6][3];
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 3; ++j) t[i][j] = G[i][0] * gk[0 * 3 + j] + G[i][1] * gk[1 * 3 + j] + G[i][2] * gk[2 * 3 + j];
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j)
        (*out)[oc * 36 + i * 6 + j] = (float)(t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2]);
  }
}

// cfg: {MT = 1, NT (1..3), WM = tile groups per block (1|2), WN = 3, R = 1, NI = 1, ALG = 7}
size_t conv_wino4_lds_bytes(const ConvDesc& d, const ConvCfg& cfg) {
  if (d.ks != 3 || d.stride != 1 || cfg.NT < 1 || cfg.NT > 3 || cfg.WM < 1 || cfg.WM > 2 || cfg.WN != 3 || d.Cin % 16 || d.Cout % 16 ||
      (long)d.B * d.H * d.W * std::max(std::max(d.in_cs, d.out_cs), d.res_cs) >= (1L << 31))
    return 0;
  return std::max<size_t>(36 * cfg.NT, (size_t)3 * cfg.WM * cfg.NT * 8) * 64 * sizeof(float4);
}

int conv_wino4_launch(const ConvDesc& d, const ConvCfg& cfg, hipStream_t stream) {
  const size_t lds = conv_wino4_lds_bytes(d, cfg);
  if (lds == 0 || !d.wfrag_wino4) {
    poco_set_error("conv(winograd 4x4): needs ks = 3, stride 1, NT 1..3, WM 1|2, WN = 3 and the 36-position weight fragments");
    return POCO_ERR_ARG;
  }
  if (d.act == 3) { poco_set_error("conv: the Winograd kernels have no per-channel ReLU split"); return POCO_ERR_ARG; }
  W4Params p{};
  p.in = d.in + l16_chan_off(d.in_co, d.W);
  p.res = d.res ? d.res + l16_chan_off(d.res_co, d.W) : nullptr;
  p.out = d.out + l16_chan_off(d.out_co, d.W);
  p.ufrag = reinterpret_cast<const float4*>(d.wfrag_wino4); p.bias = d.bias;
  p.B = d.B; p.H = d.H; p.W = d.W; p.nC16 = d.Cin / 16; p.nT16 = d.Cout / 16;
  p.in_rs = d.in_cs * d.W; p.in_ss = d.W * 16; p.res_rs = d.res_cs * d.W; p.out_rs = d.out_cs * d.W; p.out_ss = d.W * 16;
  p.TX = (d.W + 3) / 4; p.TY = (d.H + 3) / 4; p.ntiles = d.B * p.TX * p.TY;
  p.G = cfg.WM; p.act = d.act; p.res_after_act = d.res_after_act;
  p.dTX = make_fastdiv(p.TX); p.dTY = make_fastdiv(p.TY);
  const dim3 grid((p.ntiles + 16 * p.G - 1) / (16 * p.G), (p.nT16 + cfg.NT - 1) / cfg.NT);
  auto fn = cfg.NT == 3 ? conv_wino4_kernel<3> : cfg.NT == 2 ? conv_wino4_kernel<2> : conv_wino4_kernel<1>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { poco_set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return POCO_ERR_HIP; }
  }
  hipLaunchKernelGGL(fn, grid, dim3(3 * p.G * 64), lds, stream, p);
  POCO_HIP_CHECK(hipGetLastError());
  return POCO_OK;
}
