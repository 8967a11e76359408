// Fused conv (1x1 / 3x3, stride 1 / 2) + folded-BN shift + residual + ReLU for gfx950 (CDNA4).
//
// Replaces the torch call sites  conv -> bn -> relu (+ "out += residual")  of the reference:
//   pocolib/models/backbone/hrnet.py:45-47,82-88,96-97,467-472 ; hrnet_cls.py:439-444 ;
//   resnet.py:104-118 ; head/pare_head.py:468-491.
//
// Design (MI355X-first, not a translation of a cuDNN call):
//   * activations NHWC fp32; the conv is an implicit GEMM  D[co][pix] = sum_k W[co][k] X[k][pix]
//     executed on v_mfma_f32_16x16x4_f32 (exact fp32, 157 TF peak).  The *weights* are the MFMA
//     A operand and the *pixels* the B operand, so every lane ends up with 4 consecutive output
//     channels of one pixel -> 16-byte NHWC stores and 16-byte bias/residual loads.
//   * a block owns NI slabs of R output rows.  For each 16-channel slice of Cin the input halo
//     patch of those slabs is staged once in LDS (zero padded) and re-used by all ks*ks taps and
//     all output-channel tiles: HBM/L2 sees each input element ~(R+2)/R times instead of 9.
//   * LDS layout: 4 planes (one per channel quad) of [patch position][4 floats].  Lane l of an
//     MFMA needs pixel (l&15) / channel group (l>>4); ds_read_b128 services lanes in groups that
//     mix channel groups but never repeat a pixel, so with the plane stride a multiple of 256 B
//     the bank slot depends only on the pixel position -> conflict free for consecutive pixels.
//   * one ds_read_b128 yields the operands of 4 MFMAs (K is permuted: MFMA j contracts channels
//     {4g+j}); the weight fragments are pre-packed on the host in exactly that order so a wave
//     reads them from global/L2 with one fully coalesced 1 KiB load per (tap, n-tile).
//   * epilogue: + shift[co] (+ residual) (ReLU), written into a channel slice of the destination
//     (makes torch.cat free, hrnet.py:519).
#include "common.h"
#include <algorithm>
#include <cstdlib>

namespace {

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

// Epilogue shared by the direct kernels: shift (+ residual) (activation) -> NHWC channel slice.
// All loads of an n-tile group (bias, residuals) are issued BEFORE the stores of the previous group and
// nothing is loaded between stores: on gfx9 loads and stores share the one VM counter, so a load -> wait
// -> store chain per output (what a naive loop compiles to) serialises every store behind a full memory
// round trip.  out_pix(m) returns the output pixel index of sub-tile m for this lane or -1.
template <int MT, int NT, bool HAS_RES>
__device__ __forceinline__ void conv_store_tile_impl(const ConvKParams& p, f32x4 (&acc)[MT][NT], int nt0, int g,
                                                     const int (&oo)[MT]) {
  // residual loads are unconditional (dead lanes read pixel 0) so that hipcc can count them: a load under
  // a branch makes it fall back to s_waitcnt vmcnt(0) before every store
  // L16 offsets of this lane's pixels: row*rs + x*16 = pix*16 + row*(rs - 16*Wo)
  int ob[MT], rb[HAS_RES ? MT : 1];   // ob < 0: dead lane
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const uint32_t pix = (uint32_t)max(oo[m], 0);
    const uint32_t row = fdiv(pix, p.dWo);
    ob[m] = oo[m] >= 0 ? (int)(pix * 16u + row * (uint32_t)(p.out_rs - 16 * p.Wo)) : -1;
    if constexpr (HAS_RES) rb[m] = (int)(pix * 16u + row * (uint32_t)(p.res_rs - 16 * p.Wo));
  }
  auto load_res = [&](int n, float4* r) {
    const int co = (min(nt0 + n, p.nT16 - 1)) * p.out_ss + g * 4;
#pragma unroll
    for (int m = 0; m < MT; ++m)
      r[m] = *reinterpret_cast<const float4*>(p.res + rb[m] + co);
  };
  float4 sh[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
    sh[n] = *reinterpret_cast<const float4*>(p.bias + (min(nt0 + n, p.nT16 - 1)) * 16 + g * 4);
  constexpr bool PIPE = MT <= 7;          // register budget: double-buffer the residual group only for small MT
  float4 rcur[MT], rnext[PIPE ? MT : 1];
  if constexpr (HAS_RES && PIPE) load_res(0, rcur);
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    if constexpr (HAS_RES) {
      if constexpr (PIPE) { if (n + 1 < NT) load_res(n + 1, rnext); }
      else load_res(n, rcur);
    }
    const int co = (nt0 + n) * 16 + g * 4;
    const bool nok = nt0 + n < p.nT16;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      f32x4 v = acc[m][n];
      v[0] += sh[n].x; v[1] += sh[n].y; v[2] += sh[n].z; v[3] += sh[n].w;
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (HAS_RES) r = rcur[m];
      if (!p.res_after_act) { v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w; }
      if (p.act == 1 || (p.act == 3 && co >= p.relu_from)) {
        v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
      } else if (p.act == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = 1.f / (1.f + __expf(-v[e]));
      }
      if (p.res_after_act) { v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w; }
      if (nok && ob[m] >= 0)
        *reinterpret_cast<float4*>(p.out + ob[m] + (nt0 + n) * p.out_ss + g * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
    if constexpr (HAS_RES && PIPE) {
#pragma unroll
      for (int m = 0; m < MT; ++m) rcur[m] = rnext[m];
    }
  }
}

template <int MT, int NT, typename OutPix>
__device__ __forceinline__ void conv_store_tile(const ConvKParams& p, f32x4 (&acc)[MT][NT], int nt0, int g,
                                                OutPix out_pix) {
  int oo[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) oo[m] = out_pix(m);
  if (p.res != nullptr) conv_store_tile_impl<MT, NT, true>(p, acc, nt0, g, oo);
  else conv_store_tile_impl<MT, NT, false>(p, acc, nt0, g, oo);
}

template <int KS, int STRIDE, int MT, int NT>
__global__ void __launch_bounds__(512)
conv_mfma_kernel(const ConvKParams p) {
  extern __shared__ float4 patch[];
  constexpr int PAD = (KS - 1) / 2;
  const int tid = threadIdx.x;
  const int nthreads = blockDim.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave % p.WM;
  const int wn = wave / p.WM;
  const int idx = lane & 15;
  const int g = lane >> 4;
  const int s0 = blockIdx.x * p.NI;
  const int nt0 = (blockIdx.y * p.WN + wn) * NT;   // first 16-channel tile of this wave

  // ---- per-lane pixel decode for the MT sub-tiles of this wave -------------------------------
  int base[MT];   // patch position of tap (0,0) for this lane's pixel
  int ooff[MT];   // output pixel index (b*Ho + y)*Wo + x, or -1
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const uint32_t pix = (uint32_t)((wm * MT + m) * 16 + idx);
    const uint32_t sl = fdiv(pix, p.dRWo);
    const uint32_t rem = pix - sl * p.dRWo.d;
    const uint32_t yl = fdiv(rem, p.dWo);
    const uint32_t x = rem - yl * p.dWo.d;
    const uint32_t s = s0 + sl;
    const uint32_t b = fdiv(s, p.dBands);
    const uint32_t band = s - b * p.dBands.d;
    const uint32_t y = band * p.R + yl;
    const bool valid = (sl < (uint32_t)p.NI) && (s < (uint32_t)p.S) && (y < (uint32_t)p.Ho);
    base[m] = valid ? (int)((sl * p.PR + yl * STRIDE) * p.PW + x * STRIDE) : 0;
    ooff[m] = valid ? (int)((b * p.Ho + y) * p.Wo + x) : -1;
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const bool nvalid = nt0 < p.nT16;   // wave-uniform (grid.y may overshoot when WN does not divide)
  const int total_units = ((p.npos + 7) >> 3) << 5;   // (pos rounded to 8) * 4 quads

  // Register-staged pipeline: a thread's patch units (position, channel quad) are decoded ONCE (their global
  // offsets stay in registers), the loads of slice c+1 are issued right after the barrier and ride under the
  // MFMAs of slice c, and land in LDS after the next barrier.  Falls back to decode-per-slice for big patches.
  constexpr int MAXU = (MT <= 7) ? 12 : 0;
  const bool reg_stage = MAXU > 0 && total_units <= MAXU * nthreads;     // block-uniform
  auto unit_pos = [&](int u, int* q) {
    const int w = u & 31;
    *q = w >> 3;
    return (uint32_t)(((u >> 5) << 3) + (w & 7));
  };
  auto unit_goff = [&](uint32_t pos, int q) -> int {       // float offset of (pos, q) in slice 0, -1 = zero padding
    if (pos >= (uint32_t)p.npos) return -1;
    const uint32_t sl = fdiv(pos, p.dSlab);
    const uint32_t rem = pos - sl * p.dSlab.d;
    const uint32_t prow = fdiv(rem, p.dPW);
    const uint32_t pcol = rem - prow * p.dPW.d;
    const uint32_t s = s0 + sl;
    const uint32_t b = fdiv(s, p.dBands);
    const uint32_t band = s - b * p.dBands.d;
    const int iy = (int)(band * p.R) * STRIDE - PAD + (int)prow;
    const int ix = (int)pcol - PAD;
    if (s < (uint32_t)p.S && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
      return (int)((size_t)(b * p.H + iy) * p.in_rs + ix * 16 + q * 4);
    return -1;
  };
  int uoff[MAXU > 0 ? MAXU : 1];
  float4 st[MAXU > 0 ? MAXU : 1];
  if (reg_stage) {
#pragma unroll
    for (int k = 0; k < MAXU; ++k) {
      const int u = tid + k * nthreads;
      int q;
      const uint32_t pos = unit_pos(u, &q);
      uoff[k] = (u < total_units) ? unit_goff(pos, q) : -1;
    }
  }
  auto gload = [&](int c) {                     // unconditional loads (dead units read offset 0) + select
#pragma unroll
    for (int k = 0; k < MAXU; ++k)
      st[k] = *reinterpret_cast<const float4*>(p.in + (size_t)max(uoff[k], 0) + (size_t)c * p.in_ss);
  };
  auto lwrite = [&]() {
#pragma unroll
    for (int k = 0; k < MAXU; ++k) {
      const int u = tid + k * nthreads;
      if (u < total_units) {
        int q;
        const uint32_t pos = unit_pos(u, &q);
        if (pos < (uint32_t)p.npos) patch[q * p.planeF4 + pos] = (uoff[k] >= 0) ? st[k] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  if (reg_stage) gload(0);

  const int nIter = p.nC16 * p.repeat;   // repeat > 1 only for profiling experiments
  for (int it = 0; it < nIter; ++it) {
    const int c = it % p.nC16;
    if (it > 0) __syncthreads();
    // ---- stage the 16-channel slice c of the halo patch ------------------------------------
    if (reg_stage) {
      lwrite();
    } else {
#pragma unroll 4
      for (int u = tid; u < total_units; u += nthreads) {
        int q;
        const uint32_t pos = unit_pos(u, &q);
        if (pos < (uint32_t)p.npos) {
          const int off = unit_goff(pos, q);
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (off >= 0) v = *reinterpret_cast<const float4*>(p.in + (size_t)off + (size_t)c * p.in_ss);
          patch[q * p.planeF4 + pos] = v;
        }
      }
    }
    __syncthreads();
    if (reg_stage && it + 1 < nIter) gload((it + 1) % p.nC16);   // in flight during this slice's MFMAs

    if (nvalid) {
      const float4* wc = p.wfrag + ((size_t)c * p.nT16 + nt0) * 64 + lane;
      const size_t wtap = (size_t)p.nC16 * p.nT16 * 64;   // stride between taps
      const float4* pl = patch + g * p.planeF4;
      float4 wv[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) wv[n] = wc[(nt0 + n < p.nT16) ? n * 64 : 0];
      int tr = 0, ts = 0;
#pragma unroll 1
      for (int tap = 0; tap < KS * KS; ++tap) {
        const int toff = tr * p.PW + ts;
        if (++ts == KS) { ts = 0; ++tr; }
        // prefetch the next tap's weight fragments (L2-resident, 1 KiB per n-tile per wave)
        float4 wnx[NT];
        const int tnext = (tap + 1 < KS * KS) ? tap + 1 : tap;
#pragma unroll
        for (int n = 0; n < NT; ++n) wnx[n] = wc[tnext * wtap + ((nt0 + n < p.nT16) ? n * 64 : 0)];
#pragma unroll
        for (int m0 = 0; m0 < MT; m0 += 2) {
          const float4 a0 = pl[base[m0] + toff];
          const float4 a1 = pl[base[(m0 + 1 < MT) ? m0 + 1 : m0] + toff];
          const float a0v[4] = {a0.x, a0.y, a0.z, a0.w};
          const float a1v[4] = {a1.x, a1.y, a1.z, a1.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
              const float wj = (j == 0) ? wv[n].x : (j == 1) ? wv[n].y : (j == 2) ? wv[n].z : wv[n].w;
              acc[m0][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wj, a0v[j], acc[m0][n], 0, 0, 0);
              if (m0 + 1 < MT)
                acc[m0 + 1][n] =
                    __builtin_amdgcn_mfma_f32_16x16x4f32(wj, a1v[j], acc[m0 + 1][n], 0, 0, 0);
            }
          }
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) wv[n] = wnx[n];
      }
    }
  }

  // ---- epilogue: shift (+ residual) (ReLU) -> NHWC channel slice -----------------------------
  if (!nvalid) return;
  conv_store_tile<MT, NT>(p, acc, nt0, g, [&](int m) { return ooff[m]; });
}

// ------------------------------------------------------------------------------------------------
// ALG 1: same math, but the halo patch AND the weight fragments of the next 16-channel slice are
// streamed into the other half of a double-buffered LDS by LDS-DMA (global_load_lds_dwordx4) while
// the MFMAs of the current slice run: no staging VGPRs, no ds_write pass, one barrier per slice.
// The LDS images are lane-linear per wave instruction by construction (64 consecutive patch
// positions of one channel-quad plane; one 1 KiB weight fragment), which is exactly what the DMA
// writes (wave-uniform base + lane*16).  Zero padding comes from a 16-byte zero page in HBM.
// The DMA is issued through inline asm so that hipcc does not drain it at every ds_read/barrier
// (cdna_hip_programming.md 5.7); completion = our own s_waitcnt vmcnt(0) + the block barrier.
// ------------------------------------------------------------------------------------------------
__device__ float4 g_zero_page;

__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

constexpr int DMA_MAXG = 6;   // 64-position groups of the patch each wave may own

template <int KS, int STRIDE, int MT, int NT>
__global__ void __launch_bounds__(512)
conv_dma_kernel(const ConvKParams p) {
  extern __shared__ float4 smem[];
  constexpr int PAD = (KS - 1) / 2;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwaves = blockDim.x >> 6;
  const int wm = wave % p.WM;
  const int wn = wave / p.WM;
  const int idx = lane & 15;
  const int g = lane >> 4;
  const int s0 = blockIdx.x * p.NI;
  const int ntb0 = blockIdx.y * p.NTB;             // first n-tile of the block
  const int nt0 = ntb0 + wn * NT;                  // first n-tile of this wave

  int base[MT];
  int ooff[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const uint32_t pix = (uint32_t)((wm * MT + m) * 16 + idx);
    const uint32_t sl = fdiv(pix, p.dRWo);
    const uint32_t rem = pix - sl * p.dRWo.d;
    const uint32_t yl = fdiv(rem, p.dWo);
    const uint32_t x = rem - yl * p.dWo.d;
    const uint32_t s = s0 + sl;
    const uint32_t b = fdiv(s, p.dBands);
    const uint32_t band = s - b * p.dBands.d;
    const uint32_t y = band * p.R + yl;
    const bool valid = (sl < (uint32_t)p.NI) && (s < (uint32_t)p.S) && (y < (uint32_t)p.Ho);
    base[m] = valid ? (int)((sl * p.PR + yl * STRIDE) * p.PW + x * STRIDE) : 0;
    ooff[m] = valid ? (int)((b * p.Ho + y) * p.Wo + x) : -1;
  }

  // source offsets (floats) of this lane's patch positions, one per owned 64-position group
  int goff[DMA_MAXG];
#pragma unroll
  for (int k = 0; k < DMA_MAXG; ++k) {
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
      const int iy = (int)(band * p.R) * STRIDE - PAD + (int)prow;
      const int ix = (int)pcol - PAD;
      if (s < (uint32_t)p.S && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
        goff[k] = (int)((size_t)(b * p.H + iy) * p.in_rs + ix * 16);
    }
  }

  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float4*)smem;
  const int nwitems = KS * KS * p.NTB;

  auto issue = [&](int c, int buf) {
    const unsigned bb = lds_base + (unsigned)buf * (unsigned)p.bufF4 * 16u;
#pragma unroll
    for (int k = 0; k < DMA_MAXG; ++k) {
      const int grp = wave + k * nwaves;
      if (grp < p.ngroups) {
        const float* src0 = p.in + goff[k] + c * p.in_ss;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const void* src = (goff[k] >= 0) ? (const void*)(src0 + q * 4) : (const void*)&g_zero_page;
          lds_dma16(src, (unsigned)__builtin_amdgcn_readfirstlane((int)(bb + (unsigned)(q * p.planeF4 + grp * 64) * 16u)));
        }
      }
    }
    for (int i = wave; i < nwitems; i += nwaves) {
      const int tap = i / p.NTB, j = i - tap * p.NTB;
      const int nt = min(ntb0 + j, p.nT16 - 1);
      const float4* src = p.wfrag + (((size_t)tap * p.nC16 + c) * p.nT16 + nt) * 64 + lane;
      lds_dma16(src, (unsigned)__builtin_amdgcn_readfirstlane((int)(bb + (unsigned)(4 * p.planeF4 + i * 64) * 16u)));
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int nIter = p.nC16 * p.repeat;   // repeat > 1 only for profiling experiments
  for (int it = 0; it < nIter; ++it) {
    if (it + 1 < nIter) issue((it + 1) % p.nC16, (it + 1) & 1);
    const float4* bufp = smem + (size_t)(it & 1) * p.bufF4;
    const float4* pl = bufp + g * p.planeF4;
    const float4* wl = bufp + 4 * p.planeF4 + (wn * NT) * 64 + lane;
    if constexpr (MT <= 7 && KS == 3) {
      // register double-buffered taps: the LDS reads of tap t+1 are in flight while tap t's MFMAs
      // issue (one wave per SIMD has nobody else to hide the ds_read latency behind)
      float4 wv[2][NT], av[2][MT];
#pragma unroll
      for (int n = 0; n < NT; ++n) wv[0][n] = wl[n * 64];
#pragma unroll
      for (int m = 0; m < MT; ++m) av[0][m] = pl[base[m]];
#pragma unroll
      for (int tap = 0; tap < KS * KS; ++tap) {
        const int cur = tap & 1, nxt = cur ^ 1;
        if (tap + 1 < KS * KS) {
          const int toff = ((tap + 1) / KS) * p.PW + ((tap + 1) % KS);
#pragma unroll
          for (int n = 0; n < NT; ++n) wv[nxt][n] = wl[((tap + 1) * p.NTB + n) * 64];
#pragma unroll
          for (int m = 0; m < MT; ++m) av[nxt][m] = pl[base[m] + toff];
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetch reads ahead of this tap's MFMAs
#pragma unroll
        for (int m0 = 0; m0 < MT; m0 += 2) {
          const float a0v[4] = {av[cur][m0].x, av[cur][m0].y, av[cur][m0].z, av[cur][m0].w};
          const int m1 = (m0 + 1 < MT) ? m0 + 1 : m0;
          const float a1v[4] = {av[cur][m1].x, av[cur][m1].y, av[cur][m1].z, av[cur][m1].w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
              const float4 w4 = wv[cur][n];
              const float wj = (j == 0) ? w4.x : (j == 1) ? w4.y : (j == 2) ? w4.z : w4.w;
              acc[m0][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wj, a0v[j], acc[m0][n], 0, 0, 0);
              if (m0 + 1 < MT)
                acc[m0 + 1][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wj, a1v[j], acc[m0 + 1][n], 0, 0, 0);
            }
          }
        }
      }
    } else {
    int tr = 0, ts = 0;
#pragma unroll 1
    for (int tap = 0; tap < KS * KS; ++tap) {
      const int toff = tr * p.PW + ts;
      if (++ts == KS) { ts = 0; ++tr; }
      float4 wv[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) wv[n] = wl[(tap * p.NTB + n) * 64];
#pragma unroll
      for (int m0 = 0; m0 < MT; m0 += 2) {
        const float4 a0 = pl[base[m0] + toff];
        const float4 a1 = pl[base[(m0 + 1 < MT) ? m0 + 1 : m0] + toff];
        const float a0v[4] = {a0.x, a0.y, a0.z, a0.w};
        const float a1v[4] = {a1.x, a1.y, a1.z, a1.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const float wj = (j == 0) ? wv[n].x : (j == 1) ? wv[n].y : (j == 2) ? wv[n].z : wv[n].w;
            acc[m0][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wj, a0v[j], acc[m0][n], 0, 0, 0);
            if (m0 + 1 < MT)
              acc[m0 + 1][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wj, a1v[j], acc[m0 + 1][n], 0, 0, 0);
          }
        }
      }
    }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  if (nt0 >= p.nT16 || (p.dbg & 1)) return;
  conv_store_tile<MT, NT>(p, acc, nt0, g, [&](int m) { return ooff[m]; });
}

// ------------------------------------------------------------------------------------------------
// ALG 2: ALG 1 made PERSISTENT.  The grid is sized to the machine (blocks = CUs x resident blocks),
// each block walks tiles  t = blockIdx.x, +gridDim.x, ...  and the (tile, 16-channel slice) pairs
// form one flat software pipeline: the LDS-DMA of the next slice - or of the NEXT TILE's first
// slice - is in flight during the current slice's MFMAs, and a tile's epilogue stores are issued
// right after its last barrier and drain in the background while the next tile computes.  This
// hides the per-tile prologue (first patch fetch) and epilogue (an HBM-write burst of the whole
// output tile that every CU used to do at the same time) that cost 10-25 us per launch in ALG 1.
// ------------------------------------------------------------------------------------------------
template <int KS, int STRIDE, int MT, int NT>
__global__ void __launch_bounds__(512)
conv_dma_persist_kernel(const ConvKParams p) {
  extern __shared__ float4 smem[];
  constexpr int PAD = (KS - 1) / 2;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwaves = blockDim.x >> 6;
  const int wm = wave % p.WM;
  const int wn = wave / p.WM;
  const int idx = lane & 15;
  const int g = lane >> 4;
  const int ntiles = p.nblocks_m * p.nb_n;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float4*)smem;
  const int nwitems = KS * KS * p.NTB;

  int base[MT];
  int goff[DMA_MAXG], goffN[DMA_MAXG];

  auto decode_pixels = [&](int tile) {
    const int s0 = (tile % p.nblocks_m) * p.NI;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      uint32_t pix = (uint32_t)((wm * MT + m) * 16 + idx);
      asm volatile("" : "+v"(pix));   // opaque: keep hipcc from hoisting the tile-invariant part into VGPRs
      const uint32_t sl = fdiv(pix, p.dRWo);
      const uint32_t rem = pix - sl * p.dRWo.d;
      const uint32_t yl = fdiv(rem, p.dWo);
      const uint32_t x = rem - yl * p.dWo.d;
      const uint32_t s = s0 + sl;
      const uint32_t b = fdiv(s, p.dBands);
      const uint32_t band = s - b * p.dBands.d;
      const uint32_t y = band * p.R + yl;
      const bool valid = (sl < (uint32_t)p.NI) && (s < (uint32_t)p.S) && (y < (uint32_t)p.Ho);
      base[m] = valid ? (int)((sl * p.PR + yl * STRIDE) * p.PW + x * STRIDE) : 0;
    }
  };
  // output pixel index of sub-tile m (recomputed in the epilogue instead of living in VGPRs)
  auto out_pixel = [&](int tile, int m) -> int {
    const int s0 = (tile % p.nblocks_m) * p.NI;
    uint32_t pix = (uint32_t)((wm * MT + m) * 16 + idx);
    asm volatile("" : "+v"(pix));
    const uint32_t sl = fdiv(pix, p.dRWo);
    const uint32_t rem = pix - sl * p.dRWo.d;
    const uint32_t yl = fdiv(rem, p.dWo);
    const uint32_t x = rem - yl * p.dWo.d;
    const uint32_t s = s0 + sl;
    const uint32_t b = fdiv(s, p.dBands);
    const uint32_t band = s - b * p.dBands.d;
    const uint32_t y = band * p.R + yl;
    const bool valid = (sl < (uint32_t)p.NI) && (s < (uint32_t)p.S) && (y < (uint32_t)p.Ho);
    return valid ? (int)((b * p.Ho + y) * p.Wo + x) : -1;
  };
  auto decode_goff = [&](int tile, int* go) {
    const int s0 = (tile % p.nblocks_m) * p.NI;
#pragma unroll
    for (int k = 0; k < DMA_MAXG; ++k) {
      go[k] = -1;
      const int grp = wave + k * nwaves;
      uint32_t pos = (uint32_t)(grp * 64 + lane);
      asm volatile("" : "+v"(pos));
      if (grp < p.ngroups && pos < (uint32_t)p.npos) {
        const uint32_t sl = fdiv(pos, p.dSlab);
        const uint32_t rem = pos - sl * p.dSlab.d;
        const uint32_t prow = fdiv(rem, p.dPW);
        const uint32_t pcol = rem - prow * p.dPW.d;
        const uint32_t s = s0 + sl;
        const uint32_t b = fdiv(s, p.dBands);
        const uint32_t band = s - b * p.dBands.d;
        const int iy = (int)(band * p.R) * STRIDE - PAD + (int)prow;
        const int ix = (int)pcol - PAD;
        if (s < (uint32_t)p.S && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
          go[k] = (int)((size_t)(b * p.H + iy) * p.in_rs + ix * 16);
      }
    }
  };
  auto issue = [&](int c, int buf, const int* go, int ntb0) {
    const unsigned bb = lds_base + (unsigned)buf * (unsigned)p.bufF4 * 16u;
#pragma unroll
    for (int k = 0; k < DMA_MAXG; ++k) {
      const int grp = wave + k * nwaves;
      if (grp < p.ngroups) {
        const float* src0 = p.in + go[k] + c * p.in_ss;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const void* src = (go[k] >= 0) ? (const void*)(src0 + q * 4) : (const void*)&g_zero_page;
          lds_dma16(src, (unsigned)__builtin_amdgcn_readfirstlane((int)(bb + (unsigned)(q * p.planeF4 + grp * 64) * 16u)));
        }
      }
    }
    for (int i = wave; i < nwitems; i += nwaves) {
      const int tap = i / p.NTB, j = i - tap * p.NTB;
      const int nt = min(ntb0 + j, p.nT16 - 1);
      const float4* src = p.wfrag + (((size_t)tap * p.nC16 + c) * p.nT16 + nt) * 64 + lane;
      lds_dma16(src, (unsigned)__builtin_amdgcn_readfirstlane((int)(bb + (unsigned)(4 * p.planeF4 + i * 64) * 16u)));
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int t = blockIdx.x;
  if (t >= ntiles) return;
  decode_goff(t, goff);
  issue(0, 0, goff, (t / p.nblocks_m) * p.NTB);
  decode_pixels(t);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int it = 0;
  for (; t < ntiles; t += gridDim.x) {
    const int tn = t + gridDim.x;
    const bool has_next = tn < ntiles;
    const int ntb0 = (t / p.nblocks_m) * p.NTB;
    const int nt0 = ntb0 + wn * NT;
    for (int c = 0; c < p.nC16; ++c, ++it) {
      if (c + 1 < p.nC16) issue(c + 1, (it + 1) & 1, goff, ntb0);
      else if (has_next) {
        decode_goff(tn, goffN);
        issue(0, (it + 1) & 1, goffN, (tn / p.nblocks_m) * p.NTB);
      }
      const float4* bufp = smem + (size_t)(it & 1) * p.bufF4;
      const float4* pl = bufp + g * p.planeF4;
      const float4* wl = bufp + 4 * p.planeF4 + (wn * NT) * 64 + lane;
      if constexpr (MT <= 7 && KS == 3) {
        float4 wv[2][NT], av[2][MT];
#pragma unroll
        for (int n = 0; n < NT; ++n) wv[0][n] = wl[n * 64];
#pragma unroll
        for (int m = 0; m < MT; ++m) av[0][m] = pl[base[m]];
#pragma unroll
        for (int tap = 0; tap < KS * KS; ++tap) {
          const int cur = tap & 1, nxt = cur ^ 1;
          if (tap + 1 < KS * KS) {
            const int toff = ((tap + 1) / KS) * p.PW + ((tap + 1) % KS);
#pragma unroll
            for (int n = 0; n < NT; ++n) wv[nxt][n] = wl[((tap + 1) * p.NTB + n) * 64];
#pragma unroll
            for (int m = 0; m < MT; ++m) av[nxt][m] = pl[base[m] + toff];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int m0 = 0; m0 < MT; m0 += 2) {
            const int m1 = (m0 + 1 < MT) ? m0 + 1 : m0;
            const float a0v[4] = {av[cur][m0].x, av[cur][m0].y, av[cur][m0].z, av[cur][m0].w};
            const float a1v[4] = {av[cur][m1].x, av[cur][m1].y, av[cur][m1].z, av[cur][m1].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
              for (int n = 0; n < NT; ++n) {
                const float4 w4 = wv[cur][n];
                const float wj = (j == 0) ? w4.x : (j == 1) ? w4.y : (j == 2) ? w4.z : w4.w;
                acc[m0][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wj, a0v[j], acc[m0][n], 0, 0, 0);
                if (m0 + 1 < MT)
                  acc[m0 + 1][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wj, a1v[j], acc[m0 + 1][n], 0, 0, 0);
              }
            }
          }
        }
      } else {
        int tr = 0, ts = 0;
#pragma unroll 1
        for (int tap = 0; tap < KS * KS; ++tap) {
          const int toff = tr * p.PW + ts;
          if (++ts == KS) { ts = 0; ++tr; }
          float4 wv[NT];
#pragma unroll
          for (int n = 0; n < NT; ++n) wv[n] = wl[(tap * p.NTB + n) * 64];
#pragma unroll
          for (int m0 = 0; m0 < MT; m0 += 2) {
            const float4 a0 = pl[base[m0] + toff];
            const float4 a1 = pl[base[(m0 + 1 < MT) ? m0 + 1 : m0] + toff];
            const float a0v[4] = {a0.x, a0.y, a0.z, a0.w};
            const float a1v[4] = {a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
              for (int n = 0; n < NT; ++n) {
                const float wj = (j == 0) ? wv[n].x : (j == 1) ? wv[n].y : (j == 2) ? wv[n].z : wv[n].w;
                acc[m0][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wj, a0v[j], acc[m0][n], 0, 0, 0);
                if (m0 + 1 < MT)
                  acc[m0 + 1][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wj, a1v[j], acc[m0 + 1][n], 0, 0, 0);
              }
            }
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    // ---- epilogue of tile t: stores drain while the next tile's MFMAs run ------------------------
    if (nt0 < p.nT16 && !(p.dbg & 1)) conv_store_tile<MT, NT>(p, acc, nt0, g, [&](int m) { return out_pixel(t, m); });
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (has_next) {
#pragma unroll
      for (int k = 0; k < DMA_MAXG; ++k) goff[k] = goffN[k];
      decode_pixels(tn);
    }
  }
}

template <int KS, int STRIDE, int MT, int NT>
int launch_inst(int alg, const ConvKParams& kp, dim3 grid, int nthreads, size_t lds, hipStream_t stream) {
  auto fn = alg == 2 ? conv_dma_persist_kernel<KS, STRIDE, MT, NT>
            : alg == 1 ? conv_dma_kernel<KS, STRIDE, MT, NT> : conv_mfma_kernel<KS, STRIDE, MT, NT>;
  if (lds > 64 * 1024) {
    static thread_local size_t configured_alg[3] = {0, 0, 0};
    size_t& configured = configured_alg[alg];
    if (lds > configured) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
      if (e != hipSuccess) {
        poco_set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
        return POCO_ERR_HIP;
      }
      configured = 160 * 1024;
    }
  }
  hipLaunchKernelGGL(fn, grid, dim3(nthreads), lds, stream, kp);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    poco_set_error(std::string("conv launch: ") + hipGetErrorString(e));
    return POCO_ERR_HIP;
  }
  return POCO_OK;
}

template <int KS, int STRIDE>
int launch_mtnt(int alg, int MT, int NT, const ConvKParams& kp, dim3 grid, int nthreads, size_t lds,
                hipStream_t stream) {
#define POCO_CASE(mt, nt) \
  if (MT == mt && NT == nt) return launch_inst<KS, STRIDE, mt, nt>(alg, kp, grid, nthreads, lds, stream);
  POCO_CASE(4, 1) POCO_CASE(4, 2) POCO_CASE(4, 3) POCO_CASE(4, 4)
  POCO_CASE(7, 1) POCO_CASE(7, 2) POCO_CASE(7, 3) POCO_CASE(7, 4)
  POCO_CASE(13, 1) POCO_CASE(13, 2) POCO_CASE(13, 3)
#undef POCO_CASE
  poco_set_error("conv: unsupported (MT,NT) = (" + std::to_string(MT) + "," + std::to_string(NT) + ")");
  return POCO_ERR_ARG;
}

struct Geometry {
  int Ho, Wo, nbands, S, PR, PW, npos, planeF4, nblocks_m;
};

bool geometry(const ConvDesc& d, const ConvCfg& c, Geometry* g) {
  const int pad = (d.ks - 1) / 2;
  g->Ho = (d.H + 2 * pad - d.ks) / d.stride + 1;
  g->Wo = (d.W + 2 * pad - d.ks) / d.stride + 1;
  if (c.R < 1 || c.NI < 1 || c.WM < 1 || c.WN < 1) return false;
  if (c.R > g->Ho) return false;
  g->nbands = (g->Ho + c.R - 1) / c.R;
  g->S = d.B * g->nbands;
  g->PR = (c.R - 1) * d.stride + d.ks;
  g->PW = (g->Wo - 1) * d.stride + d.ks;
  g->npos = c.NI * g->PR * g->PW;
  g->planeF4 = c.ALG >= 1 ? ((g->npos + 63) / 64) * 64 : ((g->npos + 15) / 16) * 16;
  g->nblocks_m = (g->S + c.NI - 1) / c.NI;
  return true;
}

}  // namespace

size_t conv_packed_weight_floats(int Cin, int Cout16, int ks) {
  return (size_t)ks * ks * Cin * Cout16;
}

// dst layout: [tap][cin/16][cout16/16][lane 0..63][j 0..3]
//   lane = g*16 + co_l ;  value = W[co = nt*16 + co_l][cin = c16*16 + 4*g + j][tap] * scale[co]
void conv_pack_weights(const float* w, const float* scale, int Cout, int Cin, int ks, int Cout16,
                       float* dst) {
  const int nC16 = Cin / 16, nT16 = Cout16 / 16, taps = ks * ks;
  for (int tap = 0; tap < taps; ++tap)
    for (int c16 = 0; c16 < nC16; ++c16)
      for (int nt = 0; nt < nT16; ++nt)
        for (int lane = 0; lane < 64; ++lane) {
          const int g = lane >> 4, col = lane & 15;
          const int co = nt * 16 + col;
          float* o = dst + ((((size_t)tap * nC16 + c16) * nT16 + nt) * 64 + lane) * 4;
          for (int j = 0; j < 4; ++j) {
            const int ci = c16 * 16 + 4 * g + j;
            float v = 0.f;
            if (co < Cout) {
              v = w[((size_t)co * Cin + ci) * taps + tap];
              if (scale) v *= scale[co];
            }
            o[j] = v;
          }
        }
}

static size_t lds_bytes_for(const ConvDesc& d, const ConvCfg& cfg, const Geometry& g) {
  if (cfg.ALG >= 1)
    return (size_t)2 * ((size_t)4 * g.planeF4 + (size_t)d.ks * d.ks * cfg.WN * cfg.NT * 64) * sizeof(float4);
  return (size_t)4 * g.planeF4 * sizeof(float4);
}

size_t conv_lds_bytes(const ConvDesc& d, const ConvCfg& cfg) {
  if (cfg.ALG == 5) return linear_cfg_valid(d, cfg) ? (size_t)cfg.WM * 4 * 64 * sizeof(float4) : 0;
  if (cfg.ALG == 3 || cfg.ALG == 4) return conv_wino_lds_bytes(d, cfg);
  Geometry g;
  if (!geometry(d, cfg, &g)) return 0;
  return lds_bytes_for(d, cfg, g);
}

ConvCfg conv_default_cfg(const ConvDesc& d) {
  const int pad = (d.ks - 1) / 2;
  const int Ho = (d.H + 2 * pad - d.ks) / d.stride + 1;
  const int Wo = (d.W + 2 * pad - d.ks) / d.stride + 1;
  const int nT16 = d.Cout / 16;
  if (d.H == 1 && d.W == 1 && d.ks == 1 && d.Cin % 16 == 0) {   // Linear layer on B rows: latency-bound, split K
    const int nC16 = d.Cin / 16;
    return ConvCfg{1, 1, nC16 >= 32 ? 8 : nC16 >= 8 ? 4 : nC16 >= 2 ? 2 : 1, 1, 1, 1, 5};
  }
  ConvCfg best{};
  double best_cost = 1e300;
  const int mts[3] = {4, 7, 13};
  for (int mi = 0; mi < 3; ++mi) {
    const int MT = mts[mi];
    for (int NT = 1; NT <= (MT == 13 ? (d.ks == 3 ? 2 : 3) : 4); ++NT) {
      if (nT16 % NT) continue;
      for (int WN = 1; WN <= 8; WN *= 2) {
        for (int WM = 1; WM * WN <= 8; WM *= 2) {
          if (WM * WN < 2 && d.B * Ho * Wo > 4096) continue;   // keep >= 2 waves for staging
          const int mcap = WM * MT * 16;
          // candidate slab shapes: bands of R rows of one image, or NI whole images
          for (int mode = 0; mode < 2; ++mode) {
            int R, NI;
            if (mode == 0) {
              R = mcap / Wo;
              if (R < 1) continue;
              if (R > Ho) R = Ho;
              NI = 1;
              // allow several bands per block when one band underfills the wave rows
              if (R < Ho) NI = mcap / (R * Wo);
              if (NI < 1) NI = 1;
            } else {
              R = Ho;
              NI = mcap / (Ho * Wo);
              if (NI < 1) continue;
              if (NI > d.B) NI = d.B;
            }
            ConvCfg c{MT, NT, WM, WN, R, NI, 0};
            Geometry g;
            if (!geometry(d, c, &g)) continue;
            const size_t lds = (size_t)4 * g.planeF4 * 16;
            if (lds > 72 * 1024) continue;
            const int nb_n = (nT16 + WN * NT - 1) / (WN * NT);
            const long blocks = (long)g.nblocks_m * nb_n;
            // MFMA slots issued vs useful
            const double issued = (double)blocks * WM * WN * MT * NT;
            const double useful = (double)d.B * Ho * Wo / 16.0 * nT16;
            const double eff = useful / issued;
            // waves available to fill 1024 SIMDs
            const double waves = (double)blocks * WM * WN;
            const double fill = waves >= 2048 ? 1.0 : (waves >= 1024 ? 0.9 : waves / 1024.0 * 0.85);
            // staging redundancy: halo rows + re-staging per n-block
            const double halo = (double)g.PR / (c.R * d.stride) * nb_n;
            const double stage_pen = 1.0 + 0.03 * halo * (d.ks == 1 ? 3.0 : 1.0);
            const double reg_pen = (MT * NT * 4 > 128) ? 1.08 : 1.0;
            const double cost = stage_pen * reg_pen / (eff * fill);
            if (cost < best_cost) {
              best_cost = cost;
              best = c;
            }
          }
        }
      }
    }
  }
  return best;
}

int conv_launch(const ConvDesc& d, const ConvCfg& cfg, hipStream_t stream) {
  if (cfg.ALG == 5) return linear_launch(d, cfg, stream);
  if (cfg.ALG == 3 || cfg.ALG == 4) {
    if (d.Cin % 16 || d.Cout % 16 || ((d.in_cs | d.in_co | d.out_cs | d.out_co | d.res_cs | d.res_co) & 3)) {
      poco_set_error("conv: channel counts/strides must be multiples of 16/4");
      return POCO_ERR_ARG;
    }
    if (d.act == 3) { poco_set_error("conv: the Winograd kernels have no per-channel ReLU split"); return POCO_ERR_ARG; }
    return conv_wino_launch(d, cfg, stream);
  }
  if (!(d.ks == 1 || d.ks == 3) || !(d.stride == 1 || d.stride == 2)) {
    poco_set_error("conv: ks must be 1|3 and stride 1|2");
    return POCO_ERR_ARG;
  }
  if (d.Cin % 16 || d.Cout % 16) {
    poco_set_error("conv: Cin and (padded) Cout must be multiples of 16");
    return POCO_ERR_ARG;
  }
  if ((d.in_cs | d.in_co | d.out_cs | d.out_co | d.res_cs | d.res_co) & 3) {
    poco_set_error("conv: channel strides/offsets must be multiples of 4");
    return POCO_ERR_ARG;
  }
  Geometry g;
  if (!geometry(d, cfg, &g)) {
    poco_set_error("conv: invalid tile configuration");
    return POCO_ERR_ARG;
  }
  const int nwaves = cfg.WM * cfg.WN;
  if (nwaves < 1 || nwaves > 8) {
    poco_set_error("conv: WM*WN must be in 1..8");
    return POCO_ERR_ARG;
  }
  if (cfg.NI * cfg.R * g.Wo > cfg.WM * cfg.MT * 16) {
    poco_set_error("conv: block pixel count exceeds WM*MT*16");
    return POCO_ERR_ARG;
  }
  const size_t lds = lds_bytes_for(d, cfg, g);
  if (cfg.ALG >= 1 && (g.planeF4 / 64 + nwaves - 1) / nwaves > DMA_MAXG) {
    poco_set_error("conv: halo patch too large for the LDS-DMA variant");
    return POCO_ERR_ARG;
  }
  if (cfg.ALG < 0 || cfg.ALG > 3) {
    poco_set_error("conv: unknown ALG");
    return POCO_ERR_ARG;
  }
  if (lds > 160 * 1024) {
    poco_set_error("conv: halo patch does not fit in LDS");
    return POCO_ERR_ARG;
  }
  ConvKParams kp;
  kp.in = d.in + l16_chan_off(d.in_co, d.W);
  kp.res = d.res ? d.res + l16_chan_off(d.res_co, g.Wo) : nullptr;
  kp.out = d.out + l16_chan_off(d.out_co, g.Wo);
  kp.wfrag = reinterpret_cast<const float4*>(d.wfrag);
  kp.bias = d.bias;
  kp.in_rs = d.in_cs * d.W; kp.in_ss = d.W * 16;
  kp.res_rs = d.res_cs * g.Wo; kp.out_rs = d.out_cs * g.Wo; kp.out_ss = g.Wo * 16;
  kp.H = d.H; kp.W = d.W; kp.Ho = g.Ho; kp.Wo = g.Wo;
  kp.nC16 = d.Cin / 16; kp.nT16 = d.Cout / 16;
  kp.R = cfg.R; kp.NI = cfg.NI; kp.S = g.S;
  kp.PR = g.PR; kp.PW = g.PW; kp.npos = g.npos; kp.planeF4 = g.planeF4;
  kp.WM = cfg.WM; kp.WN = cfg.WN;
  kp.NTB = cfg.WN * cfg.NT;
  kp.bufF4 = 4 * g.planeF4 + d.ks * d.ks * kp.NTB * 64;
  kp.ngroups = g.planeF4 / 64;
  kp.act = d.act; kp.res_after_act = d.res_after_act; kp.relu_from = d.relu_from;
  {
    static const int rep = [] { const char* e = getenv("POCO_CONV_REPEAT"); return e ? atoi(e) : 1; }();
    kp.repeat = rep >= 0 ? rep : 1;
    static const int dbg = [] { const char* e = getenv("POCO_CONV_DBG"); return e ? atoi(e) : 0; }();
    kp.dbg = dbg;
  }
  kp.dPW = make_fastdiv(g.PW);
  kp.dSlab = make_fastdiv(g.PR * g.PW);
  kp.dBands = make_fastdiv(g.nbands);
  kp.dWo = make_fastdiv(g.Wo);
  kp.dRWo = make_fastdiv(cfg.R * g.Wo);
  const int nb_n = (kp.nT16 + cfg.WN * cfg.NT - 1) / (cfg.WN * cfg.NT);
  dim3 grid(g.nblocks_m, nb_n);
  kp.nblocks_m = g.nblocks_m; kp.nb_n = nb_n;
  if (cfg.ALG == 2) {
    // persistent: as many blocks as can be resident (LDS-limited), never more than tiles
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(2, (160 * 1024) / std::max<size_t>(lds, 1)));
    const long tiles = (long)g.nblocks_m * nb_n;
    grid = dim3((unsigned)std::min<long>(tiles, 256L * per_cu), 1);
  }
  const int nthreads = nwaves * 64;
  if (d.ks == 1 && d.stride == 1) return launch_mtnt<1, 1>(cfg.ALG, cfg.MT, cfg.NT, kp, grid, nthreads, lds, stream);
  if (d.ks == 1 && d.stride == 2) return launch_mtnt<1, 2>(cfg.ALG, cfg.MT, cfg.NT, kp, grid, nthreads, lds, stream);
  if (d.ks == 3 && d.stride == 1) return launch_mtnt<3, 1>(cfg.ALG, cfg.MT, cfg.NT, kp, grid, nthreads, lds, stream);
  return launch_mtnt<3, 2>(cfg.ALG, cfg.MT, cfg.NT, kp, grid, nthreads, lds, stream);
}
