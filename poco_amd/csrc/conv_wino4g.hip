// Winograd F(4x4,3x3) as a position-batched GEMM for SMALL planes (7x7, up to 16x16) - ALG 11 (round 3).
//
// On the 7x7 planes of HRNet's fourth branch (384 -> 384 at W48: hrnet.py:42-58, 24 convs) an image has only 2 x 2 tiles of
// 4 x 4 outputs, so the kernels in which a block owns tiles x all 36 positions (ALG 7 / 8) have nothing to amortise the
// 36-position weight stream over, and the F(2x2) kernel (ALG 4) runs 16 position GEMMs of 4.8 GFLOP on half of the CUs
// (89 us per launch, and the 8-conv chain of this branch is what the stage-4 modules end with: 0.9 ms of the W48 forward
// has only this kernel resident).  Here the three steps of the algorithm are three launches with V and M staged in HBM
// (14 MB each at 64 crops - they stay in the 256 MB infinity cache):
//   1. input transform   V_xi[ci][t] = (B^T d B)_xi          one thread per (tile, channel); V is written as
//                        [36 xi][Cin/16][T/16 blocks][4 channel quads g][16 tiles][4] = per (slice, 16-tile block) exactly the
//                        lane order of the MFMA B operand (lane = 16 g + tile), so that a wave's operand load is lane * 16 B
//   2. 36 GEMMs          M_xi[co][t] = sum_ci U_xi[co][ci] V_xi[ci][t]   register-direct fp32-MFMA GEMM (the scheme of
//                        gemm1x1.hip: weights = A operand, tiles = B operand, no LDS, no barriers) with one weight matrix per
//                        position; in this layout every operand load and every store of a wave is one contiguous KiB
//   3. output transform  Y = A^T M A + shift (+ residual) (ReLU) -> L16 activation (4 x 4 pixels per tile, clipped to the plane)
// MFMA work: 2.25 MACs per pixel and channel (F(2x2): 4, direct: 9) on all 256 CUs; U = G g G^T in float64 on the host.
#include "conv_wino4_common.h"

namespace {

// float offset of (tile t, channel c of a 16-channel slice) inside a [T/16][4][16][4] operand-order plane
__device__ __forceinline__ int opnd_off(int t, int c) { return (t >> 4) * 256 + (c >> 2) * 64 + (t & 15) * 4 + (c & 3); }

struct WgParams {
  const float* in;  const float* res;  float* out;
  float* V;  float* M;
  float* Vnext;          // fused tail (wg_mid_kernel): the NEXT conv's V, produced from this conv's output tile by tile; else nullptr
  int store_y;           // fused tail: the output tensor itself is still needed (a later residual / another consumer)
  const float4* ufrag;   // [36][Cin/16][Cout/16][64] float4 (conv_pack_weights(ks = 1) per position)
  const float* bias;
  int B, H, W, TY, TX, T, Tp;       // tiles per image = TY*TX, T = B*TY*TX, Tp = T padded to a multiple of 128
  int nC16, nT16;
  int in_rs, in_ss, res_rs, out_rs, out_ss;
  int act, res_after_act;
  int WM, WN;
  FastDiv dTpi, dTX;
};

// ---- 1. input transform ----------------------------------------------------------------------------------------------------
// thread = (channel c of the slice, tile t); 16 consecutive threads = the 16 channels of one tile (64-byte runs both ways)
__global__ void __launch_bounds__(256)
wg_in_kernel(const WgParams p) {
  const int c = threadIdx.x & 15;
  const int t = blockIdx.x * 16 + (threadIdx.x >> 4);
  const int slice = blockIdx.y;
  if (t >= p.Tp) return;
  float d[6][6];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int s = 0; s < 6; ++s) d[r][s] = 0.f;
  if (t < p.T) {
    const uint32_t b = fdiv((uint32_t)t, p.dTpi);
    const uint32_t rem = (uint32_t)t - b * (uint32_t)(p.TY * p.TX);
    const uint32_t ty = fdiv(rem, p.dTX);
    const int tx = (int)(rem - ty * (uint32_t)p.TX);
    const int iy0 = 4 * (int)ty - 1, ix0 = 4 * tx - 1;
    const float* base = p.in + (size_t)slice * p.in_ss + c;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const int iy = iy0 + r;
      const bool yok = (unsigned)iy < (unsigned)p.H;
      const float* row = base + ((size_t)b * p.H + (yok ? iy : 0)) * p.in_rs;
#pragma unroll
      for (int s = 0; s < 6; ++s) {
        const int ix = ix0 + s;
        const bool ok = yok && (unsigned)ix < (unsigned)p.W;
        const float v = row[(ok ? ix : 0) * 16];
        d[r][s] = ok ? v : 0.f;
      }
    }
  }
  // B^T d B: columns first (rows of B^T over r), then rows
  float u[6][6];
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    u[0][s] = w4::bt_row<0>(d[0][s], d[1][s], d[2][s], d[3][s], d[4][s], d[5][s]);
    u[1][s] = w4::bt_row<1>(d[0][s], d[1][s], d[2][s], d[3][s], d[4][s], d[5][s]);
    u[2][s] = w4::bt_row<2>(d[0][s], d[1][s], d[2][s], d[3][s], d[4][s], d[5][s]);
    u[3][s] = w4::bt_row<3>(d[0][s], d[1][s], d[2][s], d[3][s], d[4][s], d[5][s]);
    u[4][s] = w4::bt_row<4>(d[0][s], d[1][s], d[2][s], d[3][s], d[4][s], d[5][s]);
    u[5][s] = w4::bt_row<5>(d[0][s], d[1][s], d[2][s], d[3][s], d[4][s], d[5][s]);
  }
  float* vo = p.V + (size_t)slice * p.Tp * 16 + opnd_off(t, c);
  const size_t pstride = (size_t)p.nC16 * p.Tp * 16;
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    vo[(size_t)(r * 6 + 0) * pstride] = w4::bt_row<0>(u[r][0], u[r][1], u[r][2], u[r][3], u[r][4], u[r][5]);
    vo[(size_t)(r * 6 + 1) * pstride] = w4::bt_row<1>(u[r][0], u[r][1], u[r][2], u[r][3], u[r][4], u[r][5]);
    vo[(size_t)(r * 6 + 2) * pstride] = w4::bt_row<2>(u[r][0], u[r][1], u[r][2], u[r][3], u[r][4], u[r][5]);
    vo[(size_t)(r * 6 + 3) * pstride] = w4::bt_row<3>(u[r][0], u[r][1], u[r][2], u[r][3], u[r][4], u[r][5]);
    vo[(size_t)(r * 6 + 4) * pstride] = w4::bt_row<4>(u[r][0], u[r][1], u[r][2], u[r][3], u[r][4], u[r][5]);
    vo[(size_t)(r * 6 + 5) * pstride] = w4::bt_row<5>(u[r][0], u[r][1], u[r][2], u[r][3], u[r][4], u[r][5]);
  }
}

// ---- 2. the 36 position GEMMs ------------------------------------------------------------------------------------------------
// grid (tile groups, n groups, 36 positions); a wave owns MT 16-tile sub-tiles x NT 16-channel tiles of one position and free-runs
// over K with the operands of the next D-1 slices in flight (gemm1x1.hip's scheme; every load / store is a contiguous KiB here)
template <int MT, int NT, int D>
__global__ void __launch_bounds__(512)
wg_gemm_kernel(const WgParams p) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave % p.WM, wn = wave / p.WM;
  const int xi = blockIdx.z;
  const int mt0 = (blockIdx.x * p.WM + wm) * MT;
  const int nt0 = (blockIdx.y * p.WN + wn) * NT;
  if (nt0 >= p.nT16 || mt0 * 16 >= p.Tp) return;
  const float* vb = p.V + (size_t)xi * p.nC16 * p.Tp * 16 + (size_t)mt0 * 256 + lane * 4;          // + slice*Tp*16 + m*256
  const float4* wl = p.ufrag + (size_t)xi * p.nC16 * p.nT16 * 64 + (size_t)nt0 * 64 + lane;       // + slice*nT16*64 + n*64
  const int wslice = p.nT16 * 64, vslice = p.Tp * 16;
  int woff[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) woff[n] = (nt0 + n < p.nT16) ? n * 64 : 0;
  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 a[D][NT], b[D][MT];
  auto load = [&](int s, int c) {
#pragma unroll
    for (int n = 0; n < NT; ++n) a[s][n] = wl[(size_t)c * wslice + woff[n]];
#pragma unroll
    for (int m = 0; m < MT; ++m) b[s][m] = *reinterpret_cast<const float4*>(vb + (size_t)c * vslice + m * 256);
  };
  auto mma = [&](int s) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const float wj = (j == 0) ? a[s][n].x : (j == 1) ? a[s][n].y : (j == 2) ? a[s][n].z : a[s][n].w;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const float bj = (j == 0) ? b[s][m].x : (j == 1) ? b[s][m].y : (j == 2) ? b[s][m].z : b[s][m].w;
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wj, bj, acc[m][n], 0, 0, 0);
        }
      }
  };
  const int last = p.nC16 - 1;
#pragma unroll
  for (int s = 0; s < D - 1; ++s) load(s, min(s, last));
  const int nfull = p.nC16 / D * D;
  for (int c0 = 0; c0 < nfull; c0 += D) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
      load((u + D - 1) % D, min(c0 + u + D - 1, last));
      mma(u);
    }
  }
#pragma unroll
  for (int u = 0; u < D - 1; ++u)
    if (nfull + u < p.nC16) mma(u);
  float* mo = p.M + (size_t)xi * p.nT16 * p.Tp * 16 + (size_t)mt0 * 256 + lane * 4;
#pragma unroll
  for (int n = 0; n < NT; ++n)
    if (nt0 + n < p.nT16) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
        *reinterpret_cast<float4*>(mo + (size_t)(nt0 + n) * vslice + m * 256) = make_float4(acc[m][n][0], acc[m][n][1], acc[m][n][2], acc[m][n][3]);
    }
}

// ---- 3. output transform + epilogue ----------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
wg_out_kernel(const WgParams p) {
  const int c = threadIdx.x & 15;
  const int t = blockIdx.x * 16 + (threadIdx.x >> 4);
  const int nt = blockIdx.y;
  if (t >= p.T) return;
  const float* mi = p.M + (size_t)nt * p.Tp * 16 + opnd_off(t, c);
  const size_t pstride = (size_t)p.nT16 * p.Tp * 16;
  float m[6][6];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int s = 0; s < 6; ++s) m[r][s] = mi[(size_t)(r * 6 + s) * pstride];
  // Z = A^T M (4 x 6), Y = Z A (4 x 4)
  float z[4][6];
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    const float p12 = m[1][s] + m[2][s], m12 = m[1][s] - m[2][s], p34 = m[3][s] + m[4][s], m34 = m[3][s] - m[4][s];
    z[0][s] = m[0][s] + p12 + p34;
    z[1][s] = m12 + 2.f * m34;
    z[2][s] = p12 + 4.f * p34;
    z[3][s] = m12 + 8.f * m34 + m[5][s];
  }
  const uint32_t b = fdiv((uint32_t)t, p.dTpi);
  const uint32_t rem = (uint32_t)t - b * (uint32_t)(p.TY * p.TX);
  const uint32_t ty = fdiv(rem, p.dTX);
  const int tx = (int)(rem - ty * (uint32_t)p.TX);
  const float sh = p.bias[nt * 16 + c];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float p12 = z[i][1] + z[i][2], m12 = z[i][1] - z[i][2], p34 = z[i][3] + z[i][4], m34 = z[i][3] - z[i][4];
    const float y[4] = {z[i][0] + p12 + p34, m12 + 2.f * m34, p12 + 4.f * p34, m12 + 8.f * m34 + z[i][5]};
    const int oy = 4 * (int)ty + i;
    if (oy >= p.H) continue;
    const size_t row = (size_t)b * p.H + oy;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ox = 4 * tx + j;
      if (ox >= p.W) continue;
      float v = y[j] + sh;
      float r = 0.f;
      if (p.res) r = p.res[row * p.res_rs + (size_t)nt * p.out_ss + ox * 16 + c];
      if (!p.res_after_act) v += r;
      if (p.act == 1) v = fmaxf(v, 0.f);
      if (p.res_after_act) v += r;
      p.out[row * p.out_rs + (size_t)nt * p.out_ss + ox * 16 + c] = v;
    }
  }
}

// ---- 3'. output transform of conv k fused with the input transform of conv k+1 (consecutive ALG 11 convs of a branch chain) -----
// A block = 16 tiles x the 16 channels of one slice; with TY*TX dividing 16 these are whole images, so the 6 x 6 windows of the
// next conv's tiles (which overlap the neighbouring tiles of the same image) can be read back from an LDS copy of the block's
// output pixels: y = A^T M A + shift (+ residual) (ReLU) -> LDS (zero border) -> d -> B^T d B -> V of the next conv.
// One launch and one 7.5 us kernel less per conv, and the intermediate of a BasicBlock (conv1's output) never reaches memory.
__global__ void __launch_bounds__(256)
wg_mid_kernel(const WgParams p) {
  extern __shared__ float ytile[];                       // [16 / tpi images][4 TY + 2][4 TX + 2][16 channels]
  const int c = threadIdx.x & 15;
  const int tl = threadIdx.x >> 4;                        // tile inside the block
  const int t = blockIdx.x * 16 + tl;
  const int nt = blockIdx.y;
  const int tpi = p.TY * p.TX;
  const int PH = 4 * p.TY + 2, PW = 4 * p.TX + 2;
  for (int i = threadIdx.x; i < (16 / tpi) * PH * PW * 16; i += 256) ytile[i] = 0.f;
  __syncthreads();
  const int il = tl / tpi, rem = tl - il * tpi;
  const int ty = rem / p.TX, tx = rem - ty * p.TX;
  float* yl = ytile + ((size_t)il * PH * PW) * 16 + c;     // + (py * PW + px) * 16
  if (t < p.T) {
    const float* mi = p.M + (size_t)nt * p.Tp * 16 + opnd_off(t, c);
    const size_t pstride = (size_t)p.nT16 * p.Tp * 16;
    float m[6][6];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int s = 0; s < 6; ++s) m[r][s] = mi[(size_t)(r * 6 + s) * pstride];
    float z[4][6];
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      const float p12 = m[1][s] + m[2][s], m12 = m[1][s] - m[2][s], p34 = m[3][s] + m[4][s], m34 = m[3][s] - m[4][s];
      z[0][s] = m[0][s] + p12 + p34;
      z[1][s] = m12 + 2.f * m34;
      z[2][s] = p12 + 4.f * p34;
      z[3][s] = m12 + 8.f * m34 + m[5][s];
    }
    const int b = t / tpi;
    const float sh = p.bias[nt * 16 + c];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float p12 = z[i][1] + z[i][2], m12 = z[i][1] - z[i][2], p34 = z[i][3] + z[i][4], m34 = z[i][3] - z[i][4];
      const float y[4] = {z[i][0] + p12 + p34, m12 + 2.f * m34, p12 + 4.f * p34, m12 + 8.f * m34 + z[i][5]};
      const int oy = 4 * ty + i;
      if (oy >= p.H) continue;
      const size_t row = (size_t)b * p.H + oy;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ox = 4 * tx + j;
        if (ox >= p.W) continue;
        float v = y[j] + sh;
        float r = 0.f;
        if (p.res) r = p.res[row * p.res_rs + (size_t)nt * p.out_ss + ox * 16 + c];
        if (!p.res_after_act) v += r;
        if (p.act == 1) v = fmaxf(v, 0.f);
        if (p.res_after_act) v += r;
        if (p.store_y) p.out[row * p.out_rs + (size_t)nt * p.out_ss + ox * 16 + c] = v;
        yl[((oy + 1) * PW + ox + 1) * 16] = v;
      }
    }
  }
  __syncthreads();
  float d[6][6];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int s = 0; s < 6; ++s) d[r][s] = (t < p.T) ? yl[((4 * ty + r) * PW + 4 * tx + s) * 16] : 0.f;
  float u[6][6];
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    u[0][s] = w4::bt_row<0>(d[0][s], d[1][s], d[2][s], d[3][s], d[4][s], d[5][s]);
    u[1][s] = w4::bt_row<1>(d[0][s], d[1][s], d[2][s], d[3][s], d[4][s], d[5][s]);
    u[2][s] = w4::bt_row<2>(d[0][s], d[1][s], d[2][s], d[3][s], d[4][s], d[5][s]);
    u[3][s] = w4::bt_row<3>(d[0][s], d[1][s], d[2][s], d[3][s], d[4][s], d[5][s]);
    u[4][s] = w4::bt_row<4>(d[0][s], d[1][s], d[2][s], d[3][s], d[4][s], d[5][s]);
    u[5][s] = w4::bt_row<5>(d[0][s], d[1][s], d[2][s], d[3][s], d[4][s], d[5][s]);
  }
  // this conv's n-tile nt is slice nt of the next conv's input (Cout == next Cin)
  float* vo = p.Vnext + (size_t)nt * p.Tp * 16 + opnd_off(t, c);
  const size_t vstride = (size_t)p.nT16 * p.Tp * 16;
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    vo[(size_t)(r * 6 + 0) * vstride] = w4::bt_row<0>(u[r][0], u[r][1], u[r][2], u[r][3], u[r][4], u[r][5]);
    vo[(size_t)(r * 6 + 1) * vstride] = w4::bt_row<1>(u[r][0], u[r][1], u[r][2], u[r][3], u[r][4], u[r][5]);
    vo[(size_t)(r * 6 + 2) * vstride] = w4::bt_row<2>(u[r][0], u[r][1], u[r][2], u[r][3], u[r][4], u[r][5]);
    vo[(size_t)(r * 6 + 3) * vstride] = w4::bt_row<3>(u[r][0], u[r][1], u[r][2], u[r][3], u[r][4], u[r][5]);
    vo[(size_t)(r * 6 + 4) * vstride] = w4::bt_row<4>(u[r][0], u[r][1], u[r][2], u[r][3], u[r][4], u[r][5]);
    vo[(size_t)(r * 6 + 5) * vstride] = w4::bt_row<5>(u[r][0], u[r][1], u[r][2], u[r][3], u[r][4], u[r][5]);
  }
}

inline int tiles_padded(int T) { return (T + 127) / 128 * 128; }

}  // namespace

// U per position in conv_pack_weights(ks = 1) order: [36][Cin/16][Cout16/16][64][4]
size_t conv_wino4g_packed_floats(int Cin, int Cout16) { return (size_t)36 * Cin * Cout16; }

void conv_wino4g_pack_weights(const float* w_oihw, const float* scale, int Cout, int Cin, int Cout16, float* dst) {
  std::vector<double> u;
  w4::u_transform(w_oihw, Cout, Cin, &u);
  std::vector<float> ux((size_t)Cout * Cin);
  for (int xi = 0; xi < 36; ++xi) {
    for (size_t i = 0; i < (size_t)Cout * Cin; ++i) ux[i] = (float)(u[(size_t)xi * Cout * Cin + i] * (scale ? (double)scale[i / Cin] : 1.0));
    conv_pack_weights(ux.data(), nullptr, Cout, Cin, 1, Cout16, dst + (size_t)xi * Cin * Cout16);
  }
}

// scratch floats (V + M) for a conv of this shape
size_t conv_wino4g_scratch_floats(int B, int H, int W, int Cin, int Cout) {
  const int T = B * ((H + 3) / 4) * ((W + 3) / 4);
  return (size_t)36 * tiles_padded(T) * (2 * std::max(Cin, Cout) + Cout);      // V (two of them: chained convs ping-pong) + M
}

// can the output transform of this conv produce the next conv's V directly (wg_mid_kernel)?
bool conv_wino4g_can_chain(int H, int W, int Cout, int next_Cin) {
  const int tpi = ((H + 3) / 4) * ((W + 3) / 4);
  return Cout == next_Cin && 16 % tpi == 0;
}

// cfg: {MT, NT in (1,2,4) or (8,2): 16-tile x 16-channel sub-tiles per wave, WM, WN, R = prefetch depth D (2|3|4|6; the weights of a conv are cold: read once per forward), NI = 1, ALG = 11}
bool conv_wino4g_cfg_valid(const ConvDesc& d, const ConvCfg& cfg) {
  const bool tile = ((cfg.MT == 1 || cfg.MT == 2 || cfg.MT == 4) && (cfg.NT == 1 || cfg.NT == 2 || cfg.NT == 4)) || (cfg.MT == 8 && cfg.NT == 2);
  return d.ks == 3 && d.stride == 1 && d.H <= 16 && d.W <= 16 && d.Cin % 16 == 0 && d.Cout % 16 == 0 && tile && cfg.WM >= 1 && cfg.WN >= 1 &&
         cfg.WM * cfg.WN <= 8 && (cfg.R == 2 || cfg.R == 3 || cfg.R == 4 || cfg.R == 6) && (d.act == 0 || d.act == 1) &&
         (cfg.R * (cfg.MT + cfg.NT) + cfg.MT * cfg.NT <= 56) &&
         conv_wino4g_scratch_floats(d.B, d.H, d.W, d.Cin, d.Cout) < (1ull << 31) &&
         (long)d.B * d.H * d.W * std::max(std::max(d.in_cs, d.out_cs), d.res_cs) < (1L << 31);
}

int conv_wino4g_launch(const ConvDesc& d, const ConvCfg& cfg, hipStream_t stream) {
  if (!conv_wino4g_cfg_valid(d, cfg)) {
    poco_set_error("conv(winograd 4x4 as GEMM): ALG 11 needs ks = 3, stride 1, planes <= 16x16, (MT,NT) in {(2,4),(4,2),(4,4),(8,2)}, WM*WN <= 8, R (depth) 2|3|4|6 within the register budget, activation none|ReLU");
    return POCO_ERR_ARG;
  }
  const size_t need = conv_wino4g_scratch_floats(d.B, d.H, d.W, d.Cin, d.Cout);
  if (!d.wfrag_wino4g || !d.scratch || d.scratch_floats < need) {
    poco_set_error("conv(winograd 4x4 as GEMM): ALG 11 needs its per-position weight fragments and a scratch buffer for V and M");
    return POCO_ERR_ARG;
  }
  if ((d.in_cs | d.in_co | d.out_cs | d.out_co | d.res_cs | d.res_co) & 15) {
    poco_set_error("conv(winograd 4x4 as GEMM): channel strides/offsets must be multiples of 16");
    return POCO_ERR_ARG;
  }
  WgParams p{};
  p.in = d.in + l16_chan_off(d.in_co, d.W);
  p.res = d.res ? d.res + l16_chan_off(d.res_co, d.W) : nullptr;
  p.out = d.out + l16_chan_off(d.out_co, d.W);
  p.B = d.B; p.H = d.H; p.W = d.W; p.TY = (d.H + 3) / 4; p.TX = (d.W + 3) / 4;
  p.T = d.B * p.TY * p.TX; p.Tp = tiles_padded(p.T);
  p.nC16 = d.Cin / 16; p.nT16 = d.Cout / 16;
  const size_t vsz = (size_t)36 * p.Tp * std::max(d.Cin, d.Cout);
  p.V = d.scratch + (d.wg_vsel ? vsz : 0);
  p.M = d.scratch + 2 * vsz;
  p.Vnext = d.wg_emit_next ? d.scratch + (d.wg_vsel ? 0 : vsz) : nullptr;
  p.store_y = d.wg_store_y;
  if (d.wg_emit_next && !conv_wino4g_can_chain(d.H, d.W, d.Cout, d.Cout)) {
    poco_set_error("conv(winograd 4x4 as GEMM): the fused tail needs tiles-per-image dividing 16");
    return POCO_ERR_ARG;
  }
  p.ufrag = reinterpret_cast<const float4*>(d.wfrag_wino4g); p.bias = d.bias;
  p.in_rs = d.in_cs * d.W; p.in_ss = d.W * 16; p.res_rs = d.res_cs * d.W; p.out_rs = d.out_cs * d.W; p.out_ss = d.W * 16;
  p.act = d.act; p.res_after_act = d.res_after_act;
  p.WM = cfg.WM; p.WN = cfg.WN;
  p.dTpi = make_fastdiv(p.TY * p.TX); p.dTX = make_fastdiv(p.TX);
  if (!d.wg_skip_in) hipLaunchKernelGGL(wg_in_kernel, dim3(p.Tp / 16, p.nC16), dim3(256), 0, stream, p);
  const dim3 grid((p.Tp / 16 + cfg.MT * cfg.WM - 1) / (cfg.MT * cfg.WM), (p.nT16 + cfg.NT * cfg.WN - 1) / (cfg.NT * cfg.WN), 36);
  const dim3 block(cfg.WM * cfg.WN * 64);
#define WG_CASE(mt, nt)                                                                                                \
  if (cfg.MT == mt && cfg.NT == nt) {                                                                                  \
    if (cfg.R == 2) hipLaunchKernelGGL((wg_gemm_kernel<mt, nt, 2>), grid, block, 0, stream, p);                        \
    else if (cfg.R == 3) hipLaunchKernelGGL((wg_gemm_kernel<mt, nt, 3>), grid, block, 0, stream, p);                   \
    else if (cfg.R == 4) hipLaunchKernelGGL((wg_gemm_kernel<mt, nt, 4>), grid, block, 0, stream, p);                   \
    else hipLaunchKernelGGL((wg_gemm_kernel<mt, nt, 6>), grid, block, 0, stream, p);                                   \
  }
  WG_CASE(1, 1) WG_CASE(1, 2) WG_CASE(1, 4) WG_CASE(2, 1) WG_CASE(2, 2) WG_CASE(2, 4) WG_CASE(4, 1) WG_CASE(4, 2) WG_CASE(4, 4) WG_CASE(8, 2)
#undef WG_CASE
  if (p.Vnext) {
    const size_t lds = (size_t)(16 / (p.TY * p.TX)) * (4 * p.TY + 2) * (4 * p.TX + 2) * 16 * sizeof(float);
    hipLaunchKernelGGL(wg_mid_kernel, dim3(p.Tp / 16, p.nT16), dim3(256), lds, stream, p);
  } else {
    hipLaunchKernelGGL(wg_out_kernel, dim3((p.T + 15) / 16, p.nT16), dim3(256), 0, stream, p);
  }
  POCO_HIP_CHECK(hipGetLastError());
  return POCO_OK;
}
