// Winograd F(2x2, 3x3) convolution (stride 1, pad 1) + folded-BN shift + residual + activation on
// v_mfma_f32_16x16x4_f32  --  ALG 3 of the conv operator (same call sites as conv_mfma.hip).
//
// Why: on gfx950 exact-fp32 MFMA runs at the fp32 vector rate (157 TF), so the 3x3 convs of HRNet are
// bound by MFMA issue, not by HBM.  F(2x2,3x3) needs 16 multiplies per 2x2 output tile and input
// channel instead of 36 (2.25x fewer MFMAs) at the price of cheap VALU transforms; everything stays
// fp32 (error a few ulp larger than the direct form, far inside the 1e-3 gate).
//
//   Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A          (Lavin & Gray 2016)
//
// Mapping (MI355X-first):
//   * the 16 transform positions xi are 16 independent GEMMs  M_xi[co][tile] += U_xi[co][ci] V_xi[ci][tile];
//     one wave owns ONE 16-tile sub-tile x NT 16-channel n-tiles x ALL 16 positions, so the inverse
//     transform needs no cross-lane / LDS exchange: lane (tile = l&15, g = l>>4) ends up with the 16
//     M_xi values of its tile for 4 consecutive output channels, transforms them in registers and
//     stores the 2x2 output pixels as 16-byte NHWC pieces.
//   * the raw input halo patch of the block (R output rows x full width, 16-channel slice) is streamed
//     into LDS by LDS-DMA exactly as in ALG 1; each lane reads its tile's 4x4 window (16 ds_read_b128
//     of 4 channels each) and runs B^T d B on float4s.  The patch buffer is single: after every wave
//     has its window in registers (barrier) the DMA of the next slice overwrites it while the MFMAs of
//     this slice run.
//   * the transformed weights U (16 positions, host-side in float64, BN scale folded) are packed in MFMA
//     fragment order and double-buffered in LDS by the same DMA.
#include "common.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace {

struct FastDiv {
  uint32_t magic, d;
};
inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d;
  f.magic = (uint32_t)(((1ull << 32) + d - 1) / d);
  return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, FastDiv f) { return f.d == 1 ? n : __umulhi(n, f.magic); }

__device__ float4 g_zero_page_w[4];   // 64 B of zeros: source of the padding lanes of a 16-channel slice

__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

// Same, wave-uniform base (SGPR pair) + per-lane 32-bit byte offset: no 64-bit VALU address arithmetic.
__device__ __forceinline__ void lds_dma16_sv(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}

struct WinoParams {
  const float* in;
  const float* res;
  float* out;
  const float4* ufrag;   // [16][Cin/16][Cout16/16][64] float4
  const float* bias;
  int in_rs, in_ss, res_rs, out_rs, out_ss;   // L16 strides (floats): image row (C*W), 16-channel slice of a row (W*16)
  int H, W;              // == Ho, Wo
  int nC16, nT16;
  int R, NI, S;          // output rows per slab (even), slabs per block, total slabs
  int TX;                // tiles per row = ceil(W/2)
  int PR, PW, npos, planeF4, ngroups;
  int WM, WN, NTB;
  int ubufF4;            // float4 per U buffer = 16 * NTB * 64
  int nblocks_m, nb_n;   // ALG 4: tile grid walked by the persistent blocks
  int act, res_after_act;
  int dbg;               // profiling experiments: 4 = skip MFMAs, 8 = skip window reads, 16 = skip DMA of slices > 0
  FastDiv dPW, dSlab, dBands, dTX, dTslab /* (R/2)*TX */;
};

constexpr int WINO_MAXG = 6;

__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// Shared output stage of both Winograd kernels: v[n][k] (k = r'*2 + q', a 2x2 pixel block per lane, 4
// channels each) + shift (+ residual) (activation) -> NHWC.  Residual loads are batched and unconditional
// (dead pixels read pixel 0) and nothing is loaded between stores - see conv_store_tile in conv_mfma.hip.
// ROWSEL: -1 = all four pixels of the lane's 2x2 block, 0 / 1 = only its upper / lower pixel row (k = 2*ROWSEL + {0,1})
template <int NT, bool HAS_RES, int ROWSEL = -1>
__device__ __forceinline__ void wino_store_impl(const WinoParams& p, f32x4 (&v)[NT][4], int nt0, int g, int ob, int oy,
                                                int ox) {
  constexpr int K0 = ROWSEL < 0 ? 0 : 2 * ROWSEL, K1 = ROWSEL < 0 ? 4 : 2 * ROWSEL + 2;
  int orow[4], ocol[4];      // image row (b*H + y) and 16*x of the lane's 2x2 output pixels
  bool ok[4];
#pragma unroll
  for (int k = K0; k < K1; ++k) {
    const int yy = oy + (k >> 1), xx = ox + (k & 1);
    ok[k] = (ob >= 0) && yy < p.H && xx < p.W;
    orow[k] = ok[k] ? ob * p.H + yy : 0;
    ocol[k] = ok[k] ? xx * 16 : 0;
  }
  const float lo = p.act == 1 ? 0.f : -INFINITY;      // act is none | ReLU here (conv_wino_launch rejects the others)
  float4 sh[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) sh[n] = *reinterpret_cast<const float4*>(p.bias + min(nt0 + n, p.nT16 - 1) * 16 + g * 4);
  float4 rr[NT][4];
  if constexpr (HAS_RES) {
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int k = K0; k < K1; ++k)
        rr[n][k] = *reinterpret_cast<const float4*>(p.res + (size_t)orow[k] * p.res_rs + ocol[k] + min(nt0 + n, p.nT16 - 1) * p.out_ss + g * 4);
  }
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const bool nok = nt0 + n < p.nT16;
    const int co = (nt0 + n) * 16 + g * 4;
#pragma unroll
    for (int k = K0; k < K1; ++k) {
      f32x4 v4 = v[n][k];
      v4[0] += sh[n].x; v4[1] += sh[n].y; v4[2] += sh[n].z; v4[3] += sh[n].w;
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (HAS_RES) r = rr[n][k];
      if (!p.res_after_act) { v4[0] += r.x; v4[1] += r.y; v4[2] += r.z; v4[3] += r.w; }
      v4[0] = fmaxf(v4[0], lo); v4[1] = fmaxf(v4[1], lo); v4[2] = fmaxf(v4[2], lo); v4[3] = fmaxf(v4[3], lo);   // ReLU as a clamp
      if (p.res_after_act) { v4[0] += r.x; v4[1] += r.y; v4[2] += r.z; v4[3] += r.w; }
      if (nok && ok[k])
        *reinterpret_cast<float4*>(p.out + (size_t)orow[k] * p.out_rs + ocol[k] + (nt0 + n) * p.out_ss + g * 4) = make_float4(v4[0], v4[1], v4[2], v4[3]);
    }
  }
}
template <int NT, int ROWSEL = -1>
__device__ __forceinline__ void wino_store(const WinoParams& p, f32x4 (&v)[NT][4], int nt0, int g, int ob, int oy, int ox) {
  if (p.res != nullptr) wino_store_impl<NT, true, ROWSEL>(p, v, nt0, g, ob, oy, ox);
  else wino_store_impl<NT, false, ROWSEL>(p, v, nt0, g, ob, oy, ox);
}

// NT = 1: up to 12 waves per block (VGPR cap 170), NT = 2: up to 8 waves (cap 256)
template <int NT>
__global__ void __launch_bounds__(NT == 1 ? 768 : 512)
conv_wino_kernel(const WinoParams p) {
  extern __shared__ float4 smem[];   // [raw patch: 4 planes][U buffer 0][U buffer 1]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwaves = blockDim.x >> 6;
  const int wm = wave % p.WM;
  const int wn = wave / p.WM;
  const int idx = lane & 15;
  const int g = lane >> 4;
  const int s0 = blockIdx.x * p.NI;
  const int ntb0 = blockIdx.y * p.NTB;
  const int nt0 = ntb0 + wn * NT;

  // ---- this lane's tile -----------------------------------------------------------------------
  int base;        // patch position of the window's top-left corner
  int oy, ox, ob;  // output coordinates of the tile's (0,0) pixel; ob < 0 = no tile
  {
    const uint32_t tidx = (uint32_t)(wm * 16 + idx);
    const uint32_t sl = fdiv(tidx, p.dTslab);
    const uint32_t rem = tidx - sl * p.dTslab.d;
    const uint32_t tyl = fdiv(rem, p.dTX);
    const uint32_t tx = rem - tyl * p.dTX.d;
    const uint32_t s = s0 + sl;
    const uint32_t b = fdiv(s, p.dBands);
    const uint32_t band = s - b * p.dBands.d;
    const bool valid = (sl < (uint32_t)p.NI) && (s < (uint32_t)p.S);
    base = valid ? (int)((sl * p.PR + 2 * tyl) * p.PW + 2 * tx) : 0;
    ob = valid ? (int)b : -1;
    oy = (int)(band * p.R + 2 * tyl);
    ox = (int)(2 * tx);
  }

  // ---- raw-patch DMA bookkeeping --------------------------------------------------------------
  int goff[WINO_MAXG];
#pragma unroll
  for (int k = 0; k < WINO_MAXG; ++k) {
    goff[k] = -1;
    const int grp = wave + k * nwaves;
    const uint32_t pos = (uint32_t)(grp * 64 + lane);
    if (grp < p.ngroups && pos < (uint32_t)p.npos) {
      const uint32_t sl = fdiv(pos, p.dSlab);
      const uint32_t rem = pos - sl * p.dSlab.d;
      const uint32_t prow = fdiv(rem, p.dPW);
      const uint32_t pcol = rem - prow * p.dPW.d;
      const uint32_t s = s0 + sl;
      const uint32_t b = fdiv(s, p.dBands);
      const uint32_t band = s - b * p.dBands.d;
      const int iy = (int)(band * p.R) - 1 + (int)prow;
      const int ix = (int)pcol - 1;
      if (s < (uint32_t)p.S && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
        goff[k] = (int)((size_t)(b * p.H + iy) * p.in_rs + ix * 16);
    }
  }
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float4*)smem;
  const int nuitems = 16 * p.NTB;

  auto issue_raw = [&](int c) {
#pragma unroll
    for (int k = 0; k < WINO_MAXG; ++k) {
      const int grp = wave + k * nwaves;
      if (grp < p.ngroups) {
        const float* src0 = (goff[k] >= 0) ? p.in + goff[k] + c * p.in_ss : (const float*)g_zero_page_w;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          lds_dma16(src0 + q * 4, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + (unsigned)(q * p.planeF4 + grp * 64) * 16u)));
      }
    }
  };
  auto issue_u = [&](int c, int buf) {
    const unsigned ub = lds_base + (unsigned)(4 * p.planeF4 + buf * p.ubufF4) * 16u;
    for (int i = wave; i < nuitems; i += nwaves) {
      const int xi = i / p.NTB, j = i - xi * p.NTB;
      const int nt = min(ntb0 + j, p.nT16 - 1);
      const float4* src = p.ufrag + (((size_t)xi * p.nC16 + c) * p.nT16 + nt) * 64 + lane;
      lds_dma16(src, (unsigned)__builtin_amdgcn_readfirstlane((int)(ub + (unsigned)(i * 64) * 16u)));
    }
  };

  f32x4 acc[16][NT];
#pragma unroll
  for (int xi = 0; xi < 16; ++xi)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[xi][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  issue_raw(0);
  issue_u(0, 0);

  for (int c = 0; c < p.nC16; ++c) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                   // raw(c) and U(c) have landed
    // ---- window -> registers ---------------------------------------------------------------
    const float4* pl = smem + g * p.planeF4 + base;
    float4 v[16];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) v[r * 4 + q] = (p.dbg & 8) ? make_float4(1.f, 2.f, 3.f, 4.f) : pl[r * p.PW + q];
    __syncthreads();                                   // every wave holds its window: patch is free
    if (c + 1 < p.nC16 && !(p.dbg & 16)) {
      issue_raw(c + 1);
      issue_u(c + 1, (c + 1) & 1);
    }
    // ---- V = B^T d B  (rows, then columns; in place) ------------------------------------------
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 d0 = v[q], d1 = v[4 + q], d2 = v[8 + q], d3 = v[12 + q];
      v[q] = f4sub(d0, d2); v[4 + q] = f4add(d1, d2); v[8 + q] = f4sub(d2, d1); v[12 + q] = f4sub(d1, d3);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float4 t0 = v[r * 4], t1 = v[r * 4 + 1], t2 = v[r * 4 + 2], t3 = v[r * 4 + 3];
      v[r * 4] = f4sub(t0, t2); v[r * 4 + 1] = f4add(t1, t2); v[r * 4 + 2] = f4sub(t2, t1); v[r * 4 + 3] = f4sub(t1, t3);
    }
    // ---- 16 position GEMMs -----------------------------------------------------------------------
    const float4* ul = smem + 4 * p.planeF4 + (c & 1) * p.ubufF4 + (wn * NT) * 64 + lane;
    if (p.dbg & 4) { asm volatile("" :: "v"(v[0].x), "v"(v[5].y), "v"(v[10].z), "v"(v[15].w)); continue; }
    // the transform must be complete before the MFMA stream starts, and the U fragments of the next
    // position pair are fetched while the current pair's MFMAs issue (hipcc would otherwise sink
    // both next to their consumers: VALU->MFMA wait states + exposed ds_read latency per MFMA)
    __builtin_amdgcn_sched_barrier(0);
    float4 ub[2][2][NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) { ub[0][0][n] = ul[n * 64]; ub[0][1][n] = ul[(p.NTB + n) * 64]; }
#pragma unroll
    for (int xp = 0; xp < 16; xp += 2) {
      const int cur = (xp >> 1) & 1, nxt = cur ^ 1;
      if (xp + 2 < 16) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          ub[nxt][0][n] = ul[((xp + 2) * p.NTB + n) * 64];
          ub[nxt][1][n] = ul[((xp + 3) * p.NTB + n) * 64];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      const float a0[4] = {v[xp].x, v[xp].y, v[xp].z, v[xp].w};
      const float a1[4] = {v[xp + 1].x, v[xp + 1].y, v[xp + 1].z, v[xp + 1].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const float4 u0 = ub[cur][0][n], u1 = ub[cur][1][n];
          const float w0 = (j == 0) ? u0.x : (j == 1) ? u0.y : (j == 2) ? u0.z : u0.w;
          const float w1 = (j == 0) ? u1.x : (j == 1) ? u1.y : (j == 2) ? u1.z : u1.w;
          acc[xp][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0, a0[j], acc[xp][n], 0, 0, 0);
          acc[xp + 1][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1, a1[j], acc[xp + 1][n], 0, 0, 0);
        }
      }
    }
  }

  // ---- Y = A^T M A, epilogue ----------------------------------------------------------------------
  if (ob < 0 || nt0 >= p.nT16) return;
  f32x4 yv[NT][4];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    f32x4 s[2][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      s[0][q] = acc[q][n] + acc[4 + q][n] + acc[8 + q][n];
      s[1][q] = acc[4 + q][n] - acc[8 + q][n] - acc[12 + q][n];
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      yv[n][r * 2] = s[r][0] + s[r][1] + s[r][2];
      yv[n][r * 2 + 1] = s[r][1] - s[r][2] - s[r][3];
    }
  }
  wino_store<NT>(p, yv, nt0, g, ob, oy, ox);
}

// ------------------------------------------------------------------------------------------------
// ALG 4: Winograd with "half-position" waves and a software-pipelined input transform.
//   * a wave owns one 16-tile sub-tile, NT n-tiles and HALF of the 16 positions (the two transform
//     columns c in {0,1} or {2,3}): 8*NT accumulators instead of 16*NT, so one wave covers up to 48
//     output channels with a single window read + transform (V is reused by 3 n-tiles);
//   * the raw patch is double-buffered: while the MFMAs of slice c issue, the same wave reads the
//     window of slice c+1 from the other buffer and transforms it (VALU beside MFMA); one barrier per
//     slice, no barrier-separated read/transform/MFMA phases;
//   * the two halves of a tile meet once, at the end: each hands the partial sums of one pixel row of the 2x2
//     output blocks to its partner through LDS (the U buffers are dead by then) and finishes the other row.
// ------------------------------------------------------------------------------------------------
#ifdef WINO_TRACE
// timing probe (tools/wino_trace.py; build: hipcc ... -DWINO_TRACE=1, see the tool's docstring): every wave sums the
// shader clocks it spends in the phases of the persistent loop; [block][wave][8] written once at kernel end.
__device__ unsigned long long g_wino_trace[512 * 8 * 8];
#define TR_NOW() ((unsigned long long)__builtin_readcyclecounter())
#define TR_ADD(k, t0, t1) tr[k] += (t1) - (t0)
#else
#define TR_NOW() 0ull
#define TR_ADD(k, t0, t1) (void)0
#endif
#ifndef WINO_EXP
#define WINO_EXP 0     // timing experiments only (tools/build_exp.sh); non-zero values compute garbage
#endif
// DEPTH (round 4, cfg.MT = 3): depth of the raw and U rings.  With DEPTH 2 the fragments of slice c+1 and the patch of slice c+2 are
// requested at the top of slice c and awaited (s_waitcnt vmcnt(0)) at the top of slice c+1: they have ONE slice time to arrive.  At
// small batch sizes (few tiles per launch, 8 NT MFMAs x 4 k-steps = 0.3-0.5 us of MFMAs per slice) the LDS-DMA round trip of ~1.5 us
// IS the slice time.  DEPTH 3 requests U(c+2) and raw(c+3) at slice c and waits with a COUNTED vmcnt that leaves exactly the batch
// of the previous slice in flight: two slice times per fetch.  Costs one more raw + U buffer of LDS (NT <= 2 in practice).
__device__ __forceinline__ void wino_wait_vm(int n) {     // s_waitcnt vmcnt(n), n wave-uniform; n > 24 -> vmcnt(0) (conservative)
  switch (n) {
#define W_(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    W_(1) W_(2) W_(3) W_(4) W_(5) W_(6) W_(7) W_(8) W_(9) W_(10) W_(11) W_(12) W_(13) W_(14) W_(15) W_(16) W_(17) W_(18) W_(19) W_(20)
    W_(21) W_(22) W_(23) W_(24)
#undef W_
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}
template <int NT, int DEPTH = 2>
__global__ void __launch_bounds__(512)
conv_wino2_kernel(const WinoParams p) {
  extern __shared__ float4 smem[];   // [raw buf 0 .. DEPTH-1][U buf 0 .. DEPTH-1]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwaves = blockDim.x >> 6;
  const int wm = wave % p.WM;
  const int half = wave / p.WM;          // 0: positions (r, 0..1); 1: positions (r, 2..3)
  // every wave issues 1/8 of a slice's LDS-DMA pieces (WINO_EXP & 64: only the upper position half issues)
  const int dw = (WINO_EXP & 64) ? wm : wave;                 // this wave's index / count among the issuers
  const int dn = (WINO_EXP & 64) ? p.WM : (int)(blockDim.x >> 6);
  const int idx = lane & 15;
  const int g = lane >> 4;
  const int ntiles = p.nblocks_m * p.nb_n;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float4*)smem;
  const int rawF4 = 4 * p.planeF4;
  const int nuitems = 16 * NT;

  // Persistent: the block walks tiles t = blockIdx.x, +gridDim.x, ...; slices of consecutive tiles form one
  // pipeline (running slice counter `it` selects the LDS buffers), so the first fetches of tile k+1 are
  // in flight during the exchange/epilogue of tile k and its stores drain under tile k+1's MFMAs.
  auto decode_goff = [&](int tile, int* go) {
    const int s0 = (tile % p.nblocks_m) * p.NI;
#pragma unroll
    for (int k = 0; k < WINO_MAXG; ++k) {
      go[k] = -1;
      const int grp = dw + k * dn;
      uint32_t pos = (uint32_t)(grp * 64 + lane);
      asm volatile("" : "+v"(pos));     // opaque: keep the tile-invariant part out of long-lived VGPRs
      if (grp < p.ngroups && pos < (uint32_t)p.npos) {
        const uint32_t sl = fdiv(pos, p.dSlab);
        const uint32_t rem = pos - sl * p.dSlab.d;
        const uint32_t prow = fdiv(rem, p.dPW);
        const uint32_t pcol = rem - prow * p.dPW.d;
        const uint32_t s = s0 + sl;
        const uint32_t b = fdiv(s, p.dBands);
        const uint32_t band = s - b * p.dBands.d;
        const int iy = (int)(band * p.R) - 1 + (int)prow;
        const int ix = (int)pcol - 1;
        if (s < (uint32_t)p.S && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
          go[k] = (int)((size_t)(b * p.H + iy) * p.in_rs + ix * 16);
      }
    }
  };
  auto issue_raw = [&](int c, int it, const int* go) {
    const unsigned rb = lds_base + (unsigned)((it % DEPTH) * rawF4) * 16u;
#pragma unroll
    for (int k = 0; k < WINO_MAXG; ++k) {
      const int grp = dw + k * dn;
      if (grp < p.ngroups) {
        const float* src0 = (go[k] >= 0) ? p.in + go[k] + c * p.in_ss : (const float*)g_zero_page_w;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          lds_dma16(src0 + q * 4, (unsigned)__builtin_amdgcn_readfirstlane((int)(rb + (unsigned)(q * p.planeF4 + grp * 64) * 16u)));
      }
    }
  };
  auto issue_u = [&](int c, int it, int nt0) {
    // item i = (position xi, n-tile j) lives at ((xi*nC16 + c)*nT16 + nt) KiB of the packed buffer: a wave-uniform
    // base for the slice + a 32-bit per-item offset stepped incrementally (scalar ALU only, one v_add per DMA)
    const unsigned ub0 = lds_base + (unsigned)(DEPTH * rawF4 + (it % DEPTH) * p.ubufF4) * 16u;
    const char* sb = reinterpret_cast<const char*>(p.ufrag) + (size_t)((unsigned)c * (unsigned)p.nT16) * 1024u;
    const unsigned xstride = (unsigned)(p.nC16 * p.nT16) * 1024u;
    // every block reads the SAME fragments of slice c from L2: rotate the issue order by the block index so that
    // the CUs of an XCD do not all queue on one L2 channel at the same moment (dbg & 128 = plain order)
    const int rot = (p.dbg & 128) ? 0 : (int)((blockIdx.x >> 3) * 5u) % nuitems;
    for (int i = dw; i < nuitems; i += dn) {
      int ii = i + rot;
      if (ii >= nuitems) ii -= nuitems;
      const int xi = ii / NT, j = ii - xi * NT;
      const unsigned off = (unsigned)xi * xstride + (unsigned)min(nt0 + j, p.nT16 - 1) * 1024u;
      lds_dma16_sv(sb, off + (unsigned)lane * 16u, (unsigned)__builtin_amdgcn_readfirstlane((int)(ub0 + (unsigned)ii * 1024u)));
    }
  };

  // Every wave issues its share of the LDS-DMA and waits for it itself (s_waitcnt vmcnt(0) before the barrier that
  // publishes the data).  On gfx9 stores and loads share the one VM counter, so the waits are placed where no
  // store of the wave can still be in flight for long: before its output stores, and at the top of slices c > 0.
  const bool dma_wave = (WINO_EXP & 64) ? (half == 1) : true;
  // XCD-aware walk: workgroup b runs on XCD b % 8 (round-robin dispatch), so XCD x owns the contiguous tile
  // range [x*per, (x+1)*per) and its gridDim/8 blocks stride through it: neighbouring row bands (shared halo
  // rows) and the same tiles of consecutive layers meet in one L2 instead of eight.  dbg & 64 = plain walk.
  int t = blockIdx.x, tstep = gridDim.x, tend = ntiles;
  if (!(p.dbg & 64) && (gridDim.x & 7) == 0 && ntiles >= (int)gridDim.x) {
    const int per = (ntiles + 7) >> 3;
    t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    tstep = gridDim.x >> 3;
    tend = min(ntiles, ((int)(blockIdx.x & 7) + 1) * per);
  }
  if (t >= tend) return;
  int goff[WINO_MAXG], goffN[WINO_MAXG];
  int it0 = 0;
  bool first_tile = true;
#ifdef WINO_TRACE
  unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long tr_start = TR_NOW();
  const unsigned long long tr_rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  if (dma_wave) {
    decode_goff(t, goff);
    const int nt0 = (t / p.nblocks_m) * NT;
    issue_raw(0, it0, goff);
    issue_u(0, it0, nt0);
    if (p.nC16 > 1) issue_raw(1, it0 + 1, goff);
    if constexpr (DEPTH == 3) {
      if (p.nC16 > 1) issue_u(1, it0 + 1, nt0);
      if (p.nC16 > 2) issue_raw(2, it0 + 2, goff);
    }
  }
  // DMA instructions this wave issues per slice (wave-uniform): its share of the U items and 4 planes per raw group
  int nU = 0, nR = 0;
  for (int i = dw; i < nuitems; i += dn) ++nU;
#pragma unroll
  for (int k = 0; k < WINO_MAXG; ++k) if (dw + k * dn < p.ngroups) nR += 4;

  for (; t < tend; t += tstep) {
    const int tn = t + tstep;
    const bool has_next = tn < tend;
    const int nt0 = (t / p.nblocks_m) * NT;
    // ---- this lane's tile ---------------------------------------------------------------------
    int base, oy, ox, ob;
    {
      const int s0 = (t % p.nblocks_m) * p.NI;
      uint32_t tidx = (uint32_t)(wm * 16 + idx);
      asm volatile("" : "+v"(tidx));
      const uint32_t sl = fdiv(tidx, p.dTslab);
      const uint32_t rem = tidx - sl * p.dTslab.d;
      const uint32_t tyl = fdiv(rem, p.dTX);
      const uint32_t tx = rem - tyl * p.dTX.d;
      const uint32_t s = s0 + sl;
      const uint32_t b = fdiv(s, p.dBands);
      const uint32_t band = s - b * p.dBands.d;
      const bool valid = (sl < (uint32_t)p.NI) && (s < (uint32_t)p.S);
      base = valid ? (int)((sl * p.PR + 2 * tyl) * p.PW + 2 * tx) : 0;
      ob = valid ? (int)b : -1;
      oy = (int)(band * p.R + 2 * tyl);
      ox = (int)(2 * tx);
    }
    f32x4 acc[8][NT];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // the first fetches of a tile are waited for by their issuer: here for the block's first tile, inside the
    // previous tile's output stage otherwise (before that wave's stores, so that no wait ever covers a store)
    [[maybe_unused]] const unsigned long long tA = TR_NOW();
    if (dma_wave && first_tile) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    first_tile = false;
    __syncthreads();                   // raw(0), U(0) (and raw(1)) of this tile have landed
    [[maybe_unused]] unsigned long long tB = TR_NOW();
    TR_ADD(0, tA, tB);                 // 0: first-slice wait + tile-start barrier

    // The K loop is instantiated once per half (wave-uniform) so that its body is ONE basic block: hipcc
    // can then interleave the window reads + transform of slice c+1 with the MFMAs of slice c.
    auto kloop = [&](auto HC) {
      constexpr int HALF = decltype(HC)::value;
      auto load_transform = [&](int it, float4* vout) {
        const float4* pl = smem + (it % DEPTH) * rawF4 + g * p.planeF4 + base + HALF;
        float4 d[4][3];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int a = 0; a < 3; ++a) d[r][a] = pl[r * p.PW + a];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const float4 d0 = d[0][a], d1 = d[1][a], d2 = d[2][a], d3 = d[3][a];
          d[0][a] = f4sub(d0, d2); d[1][a] = f4add(d1, d2); d[2][a] = f4sub(d2, d1); d[3][a] = f4sub(d1, d3);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if constexpr (HALF == 0) { vout[r * 2] = f4sub(d[r][0], d[r][2]); vout[r * 2 + 1] = f4add(d[r][1], d[r][2]); }
          else { vout[r * 2] = f4sub(d[r][1], d[r][0]); vout[r * 2 + 1] = f4sub(d[r][0], d[r][2]); }   // local cols = window cols 1,2,3
        }
      };
      float4 vcur[8];
      load_transform(it0, vcur);
      for (int c = 0; c < p.nC16; ++c) {
        const int it = it0 + c;
        // raw(c+1) and U(c) landed; everybody is done with slice c-1 (for c == 0: with the window read of
        // slice 0 above, whose buffer the raw(2) DMA below overwrites)
        if constexpr (HALF == 1 || !(WINO_EXP & 64)) {
          // slice 0 has nothing to wait for (its data landed before the tile's first barrier); the wait of slice 1
          // also retires this wave's output stores of the previous tile, long since written
          [[maybe_unused]] const unsigned long long t0 = TR_NOW();
          TR_ADD(1, tB, t0);           // 1: slice body (window reads, transform, MFMAs, DMA issue)
          if constexpr (DEPTH == 3) {
            // everything but the batch requested at slice c-1 (U(c+1), raw(c+2)) has landed: U(c) and raw(c+1) are there
            if (c > 0) wino_wait_vm((c + 1 < p.nC16 ? nU : 0) + (c + 2 < p.nC16 ? nR : 0));
          } else {
            if (c > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
          [[maybe_unused]] const unsigned long long t1 = TR_NOW();
          TR_ADD(2, t0, t1);           // 2: waiting for this wave's own DMA
          tB = t1;
        }
#if !(WINO_EXP & 8)
        __syncthreads();
#endif
        {
          [[maybe_unused]] const unsigned long long t2 = TR_NOW();
          TR_ADD(3, tB, t2);           // 3: slice barrier
          tB = t2;
        }
        // the two waves of a SIMD (lower / upper half of one tile group) issue their DMA share at different points of
        // the slice - the lower half here, the upper half after its first two transform rows - so that one of them
        // always feeds the MFMA pipe while the other sits in the (slow) DMA issue
        auto issue_slice = [&]() {
#if !(WINO_EXP & 4)
#if !(WINO_EXP & 16)
          if (c + DEPTH - 1 < p.nC16) issue_u(c + DEPTH - 1, it + DEPTH - 1, nt0);      // (into the slot of U(c-1): consumed before this slice's barrier)
#endif
#if !(WINO_EXP & 32)
          if (c + DEPTH < p.nC16) issue_raw(c + DEPTH, it + DEPTH, goff);               // (into the slot of raw(c): its window was read during slice c-1)
#endif
#endif
        };
#ifndef WINO_ISSUE_LO
#define WINO_ISSUE_LO (-1)     // transform row before which the lower / upper half issues its DMA share (-1 = slice top)
#define WINO_ISSUE_HI 2
#endif
        constexpr int ISSUE_AT = (WINO_EXP & 8192) ? -1 : (HALF == 0 ? WINO_ISSUE_LO : WINO_ISSUE_HI);
        constexpr bool LATE_ISSUE = ISSUE_AT >= 0 && !(WINO_EXP & 64);
        if constexpr ((HALF == 1 || !(WINO_EXP & 64)) && !LATE_ISSUE) issue_slice();
        float4 vnext[8];
#if WINO_EXP & 1
#pragma unroll
        for (int i = 0; i < 8; ++i) { vnext[i] = vcur[i]; asm volatile("" : "+v"(vnext[i].x), "+v"(vnext[i].y), "+v"(vnext[i].z), "+v"(vnext[i].w)); }
#else
        load_transform(it + 1, vnext);   // past the last slice this reads stale LDS and is never used
#endif
        const float4* ul = smem + DEPTH * rawF4 + (it % DEPTH) * p.ubufF4 + (2 * HALF * NT) * 64 + lane;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if constexpr (LATE_ISSUE) { if (r == ISSUE_AT) issue_slice(); }
          float4 u0[NT], u1[NT];
#pragma unroll
          for (int n = 0; n < NT; ++n) {
#if WINO_EXP & 2
            u0[n] = vcur[(r + n) & 7]; u1[n] = vcur[(r + n + 3) & 7];
#else
            u0[n] = ul[((4 * r) * NT + n) * 64]; u1[n] = ul[((4 * r + 1) * NT + n) * 64];
#endif
          }
          const float a0[4] = {vcur[2 * r].x, vcur[2 * r].y, vcur[2 * r].z, vcur[2 * r].w};
          const float a1[4] = {vcur[2 * r + 1].x, vcur[2 * r + 1].y, vcur[2 * r + 1].z, vcur[2 * r + 1].w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
              const float w0 = (j == 0) ? u0[n].x : (j == 1) ? u0[n].y : (j == 2) ? u0[n].z : u0[n].w;
              const float w1 = (j == 0) ? u1[n].x : (j == 1) ? u1[n].y : (j == 2) ? u1[n].z : u1[n].w;
              acc[2 * r][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0, a0[j], acc[2 * r][n], 0, 0, 0);
              acc[2 * r + 1][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1, a1[j], acc[2 * r + 1][n], 0, 0, 0);
            }
          }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) vcur[i] = vnext[i];
      }
    };
    if (half == 0) kloop(std::integral_constant<int, 0>{});
    else kloop(std::integral_constant<int, 1>{});

    // ---- partial inverse transform of this half --------------------------------------------------
    // s[r'][q] = A^T over r ;  y[.][0] = s0+s1+s2, y[.][1] = s1-s2-s3 ; this half holds q = 2*half + {0,1}
    f32x4 y[NT][4];   // [n][r'*2 + q']
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      f32x4 sA[2], sB[2];
      sA[0] = acc[0][n] + acc[2][n] + acc[4][n];  sA[1] = acc[2][n] - acc[4][n] - acc[6][n];
      sB[0] = acc[1][n] + acc[3][n] + acc[5][n];  sB[1] = acc[3][n] - acc[5][n] - acc[7][n];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        if (half == 0) { y[n][r * 2] = sA[r] + sB[r]; y[n][r * 2 + 1] = sB[r]; }
        else           { y[n][r * 2] = sA[r];         y[n][r * 2 + 1] = -sA[r] - sB[r]; }
      }
    }
    [[maybe_unused]] const unsigned long long tC = TR_NOW();
    TR_ADD(1, tB, tC);                 // the last slice's body + partial inverse transform
    const int itl = it0 + p.nC16 - 1;  // last slice of this tile
    const int itn = itl + 1;           // first slice of the next tile
    if (has_next && dma_wave) decode_goff(tn, goffN);   // address math of the next tile's patch: off the critical path
    __syncthreads();                   // every wave is done with the raw and U buffers of this tile
    if (has_next && dma_wave) {
      issue_raw(0, itn, goffN);
      issue_u(0, itn, (tn / p.nblocks_m) * NT);
      if (p.nC16 > 1) issue_raw(1, itn + 1, goffN);
      if constexpr (DEPTH == 3) {
        if (p.nC16 > 1) issue_u(1, itn + 1, (tn / p.nblocks_m) * NT);
        if (p.nC16 > 2) issue_raw(2, itn + 2, goffN);
      }
    }
    // Output stage, split between the halves: the lower half finishes the upper pixel row of every 2x2 block
    // (k = 0,1), the upper half the lower row (k = 2,3); each hands the partial sums of the OTHER row to its
    // partner through LDS (the last slice's U buffer is dead): [wave][n][2][lane].
    float4* xch = smem + DEPTH * rawF4 + (itl % DEPTH) * p.ubufF4;
    {
      const int ks = half == 0 ? 2 : 0;                 // the row this half gives away
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const f32x4 v = (half == 0) ? y[n][2 + k] : y[n][k];
          xch[((wave * NT + n) * 2 + k) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
        }
      (void)ks;
    }
    __syncthreads();
    if (!(p.dbg & 1)) {
      const int pw = half == 0 ? wave + p.WM : wave - p.WM;    // partner: same tile group, other half
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const float4 o = xch[((pw * NT + n) * 2 + k) * 64 + lane];
          f32x4& d = (half == 0) ? y[n][k] : y[n][2 + k];
          d[0] += o.x; d[1] += o.y; d[2] += o.z; d[3] += o.w;
        }
      // a DMA issuer retires the next tile's first fetches BEFORE storing (no later wait then covers a store)
      if (dma_wave) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (half == 0) wino_store<NT, 0>(p, y, nt0, g, ob, oy, ox);
      else wino_store<NT, 1>(p, y, nt0, g, ob, oy, ox);
    } else if (dma_wave) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (has_next) {
#pragma unroll
      for (int k = 0; k < WINO_MAXG; ++k) goff[k] = goffN[k];
    }
    it0 = itn % DEPTH;
    [[maybe_unused]] const unsigned long long tD = TR_NOW();
    TR_ADD(4, tC, tD);                 // 4: end-of-tile stage (barriers, exchange, next tile's first issue, stores)
  }
#ifdef WINO_TRACE
  tr[5] = TR_NOW() - tr_start;         // 5: whole kernel
  tr[6] = tr_start;
  tr[7] = __builtin_amdgcn_s_memrealtime() - tr_rt0;   // 100 MHz reference clock over the same interval
  if (lane == 0 && blockIdx.x < 512) {
#pragma unroll
    for (int k = 0; k < 8; ++k) g_wino_trace[((size_t)blockIdx.x * 8 + wave) * 8 + k] = tr[k];
  }
#endif
}
#ifdef WINO_TRACE
extern "C" int poco_debug_wino_trace(unsigned long long* host, size_t n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wino_trace), n * sizeof(unsigned long long));
}
#endif

struct WGeo {
  int TX, nbands, S, PR, PW, npos, planeF4, nblocks_m;
};
bool wgeo(const ConvDesc& d, const ConvCfg& c, WGeo* g) {
  if (c.R < 2 || (c.R & 1) || c.NI < 1 || c.WM < 1 || c.WN < 1 || c.NT < 1 || c.NT > (c.ALG == 4 ? 3 : 2)) return false;
  g->TX = (d.W + 1) / 2;
  const int Hc = (d.H + 1) / 2 * 2;
  if (c.R > Hc) return false;
  g->nbands = (d.H + c.R - 1) / c.R;
  g->S = d.B * g->nbands;
  g->PR = c.R + 2;
  g->PW = 2 * g->TX + 2;
  g->npos = c.NI * g->PR * g->PW;
  g->planeF4 = (g->npos + 63) / 64 * 64;   // dead lanes use base 0: every window stays inside its slab (PR >= 4)
  g->nblocks_m = (g->S + c.NI - 1) / c.NI;
  return true;
}

}  // namespace

size_t conv_wino_lds_bytes(const ConvDesc& d, const ConvCfg& cfg) {
  WGeo g;
  if (!wgeo(d, cfg, &g)) return 0;
  if (cfg.ALG == 4) { // DEPTH raw buffers + DEPTH U buffers (the exchange area at the end reuses a U buffer); DEPTH = 3 with cfg.MT = 3
    if (cfg.MT != 1 && !(POCO_EXPERIMENTS && cfg.MT == 3)) return 0;      // (the 3-deep rings: 6-8 % slower, experiment builds only)
    const size_t depth = cfg.MT == 3 ? 3 : 2;
    return (depth * 4 * g.planeF4 + depth * 16 * cfg.NT * 64) * sizeof(float4);
  }
  return ((size_t)4 * g.planeF4 + (size_t)2 * 16 * cfg.WN * cfg.NT * 64) * sizeof(float4);
}

// U = G g G^T per (co, ci), float64 on the host, written as a [Cout][Cin][16] "16-tap" filter that
// conv_pack_weights(ks = 4) lays out in fragment order.
void conv_wino_transform_weights(const float* w_oihw, int Cout, int Cin, std::vector<float>* out) {
  static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
  out->assign((size_t)Cout * Cin * 16, 0.f);
  for (size_t oc = 0; oc < (size_t)Cout * Cin; ++oc) {
    const float* gk = w_oihw + oc * 9;
    double t[4][3];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 3; ++j) t[i][j] = G[i][0] * gk[0 * 3 + j] + G[i][1] * gk[1 * 3 + j] + G[i][2] * gk[2 * 3 + j];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j)
        (*out)[oc * 16 + i * 4 + j] = (float)(t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2]);
  }
}

int conv_wino_launch(const ConvDesc& d, const ConvCfg& cfg, hipStream_t stream) {
  if (d.ks != 3 || d.stride != 1 || !d.wfrag_wino) {
    poco_set_error("conv(winograd): needs a 3x3 stride-1 conv with transformed weights");
    return POCO_ERR_ARG;
  }
  WGeo g;
  if (!wgeo(d, cfg, &g)) { poco_set_error("conv(winograd): invalid tile configuration"); return POCO_ERR_ARG; }
  if (cfg.ALG == 4) {
    if (cfg.WN != 2 || cfg.WM > 4) { poco_set_error("conv(winograd/half): WN must be 2 (the two position halves), WM <= 4"); return POCO_ERR_ARG; }
    if (cfg.NI * (cfg.R / 2) * g.TX > cfg.WM * 16) { poco_set_error("conv(winograd/half): tiles per block exceed WM*16"); return POCO_ERR_ARG; }
    const int nw = 2 * cfg.WM;
    if ((g.planeF4 / 64 + cfg.WM - 1) / cfg.WM > WINO_MAXG) { poco_set_error("conv(winograd/half): patch too large"); return POCO_ERR_ARG; }
    const size_t lds4 = conv_wino_lds_bytes(d, cfg);
    if (lds4 > 160 * 1024 || (size_t)cfg.WM * cfg.NT * 4 * 64 > (size_t)2 * 16 * cfg.NT * 64) { poco_set_error("conv(winograd/half): LDS budget exceeded"); return POCO_ERR_ARG; }
    WinoParams p;
    p.in = d.in + l16_chan_off(d.in_co, d.W); p.res = d.res ? d.res + l16_chan_off(d.res_co, d.W) : nullptr; p.out = d.out + l16_chan_off(d.out_co, d.W);
    p.ufrag = reinterpret_cast<const float4*>(d.wfrag_wino); p.bias = d.bias;
    p.in_rs = d.in_cs * d.W; p.in_ss = d.W * 16; p.res_rs = d.res_cs * d.W; p.out_rs = d.out_cs * d.W; p.out_ss = d.W * 16;
    p.H = d.H; p.W = d.W; p.nC16 = d.Cin / 16; p.nT16 = d.Cout / 16;
    p.R = cfg.R; p.NI = cfg.NI; p.S = g.S; p.TX = g.TX; p.PR = g.PR; p.PW = g.PW; p.npos = g.npos; p.planeF4 = g.planeF4;
    p.ngroups = g.planeF4 / 64;
    p.WM = cfg.WM; p.WN = 2; p.NTB = cfg.NT; p.ubufF4 = 16 * cfg.NT * 64;
    p.act = d.act; p.res_after_act = d.res_after_act;
    p.dbg = 0;
#if POCO_PROBES       // timing-probe builds only (tools/build_exp.sh conv_wino.hip POCO_PROBES 1)
    {
      static const int dbg4 = [] { const char* e = getenv("POCO_CONV_DBG"); return e ? atoi(e) : 0; }();
      p.dbg = dbg4;
    }
#endif
    p.dPW = make_fastdiv(g.PW); p.dSlab = make_fastdiv(g.PR * g.PW); p.dBands = make_fastdiv(g.nbands);
    p.dTX = make_fastdiv(g.TX); p.dTslab = make_fastdiv((cfg.R / 2) * g.TX);
    p.nblocks_m = g.nblocks_m; p.nb_n = (p.nT16 + cfg.NT - 1) / cfg.NT;
    const long tiles4 = (long)p.nblocks_m * p.nb_n;
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(2, (160 * 1024) / std::max<size_t>(lds4, 1)));
    // balanced persistent grid: every block walks the same number of tiles (896 tiles -> 224 blocks x 4 instead of
    // 256 blocks doing 4 or 3), which takes the same time and leaves the spare CUs to the other lanes' kernels
    const long cap4 = 256L * per_cu;
    const long rounds4 = (tiles4 + cap4 - 1) / cap4;
    long g4 = (tiles4 + rounds4 - 1) / rounds4;
    if (g4 > 8) g4 = std::min(cap4, (g4 + 7) / 8 * 8);      // multiple of 8 for the XCD-aware walk
    dim3 grid4((unsigned)g4, 1);
    const bool deep = cfg.MT == 3;
    if (lds4 == 0) { poco_set_error("conv(winograd/half): MT must be 1 (2-deep rings; 3 = 3-deep rings exists in experiment builds only)"); return POCO_ERR_ARG; }
    void (*fn4)(const WinoParams) =
#if POCO_EXPERIMENTS      // the 3-deep-ring instances exist only in experiment builds (python -m poco_amd.build --experiments)
        deep ? (cfg.NT == 3 ? conv_wino2_kernel<3, 3> : cfg.NT == 2 ? conv_wino2_kernel<2, 3> : conv_wino2_kernel<1, 3>) :
#endif
        (cfg.NT == 3 ? conv_wino2_kernel<3, 2> : cfg.NT == 2 ? conv_wino2_kernel<2, 2> : conv_wino2_kernel<1, 2>);
    if (lds4 > 64 * 1024) {
      static thread_local bool configured4[8] = {};
      if (!configured4[cfg.NT + (deep ? 4 : 0)]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn4), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { poco_set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return POCO_ERR_HIP; }
        configured4[cfg.NT + (deep ? 4 : 0)] = true;
      }
    }
    hipLaunchKernelGGL(fn4, grid4, dim3(nw * 64), lds4, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { poco_set_error(std::string("conv(winograd/half) launch: ") + hipGetErrorString(e)); return POCO_ERR_HIP; }
    return POCO_OK;
  }
  const int nwaves = cfg.WM * cfg.WN;
  if (nwaves > (cfg.NT == 1 ? 12 : 8)) { poco_set_error("conv(winograd): at most 12 (NT=1) / 8 (NT=2) waves per block"); return POCO_ERR_ARG; }
  if (cfg.NI * (cfg.R / 2) * g.TX > cfg.WM * 16) { poco_set_error("conv(winograd): tiles per block exceed WM*16"); return POCO_ERR_ARG; }
  if ((g.planeF4 / 64 + nwaves - 1) / nwaves > WINO_MAXG) { poco_set_error("conv(winograd): patch too large"); return POCO_ERR_ARG; }
  const size_t lds = conv_wino_lds_bytes(d, cfg);
  if (lds > 160 * 1024) { poco_set_error("conv(winograd): LDS budget exceeded"); return POCO_ERR_ARG; }
  WinoParams p;
  p.in = d.in + l16_chan_off(d.in_co, d.W); p.res = d.res ? d.res + l16_chan_off(d.res_co, d.W) : nullptr; p.out = d.out + l16_chan_off(d.out_co, d.W);
    p.ufrag = reinterpret_cast<const float4*>(d.wfrag_wino); p.bias = d.bias;
  p.in_rs = d.in_cs * d.W; p.in_ss = d.W * 16; p.res_rs = d.res_cs * d.W; p.out_rs = d.out_cs * d.W; p.out_ss = d.W * 16;
  p.H = d.H; p.W = d.W; p.nC16 = d.Cin / 16; p.nT16 = d.Cout / 16;
  p.R = cfg.R; p.NI = cfg.NI; p.S = g.S; p.TX = g.TX; p.PR = g.PR; p.PW = g.PW; p.npos = g.npos; p.planeF4 = g.planeF4;
  p.ngroups = g.planeF4 / 64;
  p.WM = cfg.WM; p.WN = cfg.WN; p.NTB = cfg.WN * cfg.NT; p.ubufF4 = 16 * p.NTB * 64;
  p.act = d.act; p.res_after_act = d.res_after_act;
  p.dbg = 0;
#if POCO_PROBES
  {
    static const int dbg = [] { const char* e = getenv("POCO_CONV_DBG"); return e ? atoi(e) : 0; }();
    p.dbg = dbg;
  }
#endif
  p.dPW = make_fastdiv(g.PW); p.dSlab = make_fastdiv(g.PR * g.PW); p.dBands = make_fastdiv(g.nbands);
  p.dTX = make_fastdiv(g.TX); p.dTslab = make_fastdiv((cfg.R / 2) * g.TX);
  const int nb_n = (p.nT16 + p.NTB - 1) / p.NTB;
  dim3 grid(g.nblocks_m, nb_n);
  auto fn = cfg.NT == 2 ? conv_wino_kernel<2> : conv_wino_kernel<1>;
  if (lds > 64 * 1024) {
    static thread_local bool configured[3] = {false, false, false};
    if (!configured[cfg.NT]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) { poco_set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return POCO_ERR_HIP; }
      configured[cfg.NT] = true;
    }
  }
  hipLaunchKernelGGL(fn, grid, dim3(nwaves * 64), lds, stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { poco_set_error(std::string("conv(winograd) launch: ") + hipGetErrorString(e)); return POCO_ERR_HIP; }
  return POCO_OK;
}
