// Device code of the direct conv kernels (ALG 0/1/2) + their launch dispatch.  Included by conv_mfma_k*.hip only.
#pragma once
#include "conv_mfma_types.h"

namespace {

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

  // weight fragments of the tap that runs next, fetched one tap ahead (L2-resident, 1 KiB per n-tile per wave);
  // the last tap of a slice fetches tap 0 of the next slice, so no slice starts on an exposed L2 round trip
  const size_t wtap = (size_t)p.nC16 * p.nT16 * 64;   // stride between taps
  const size_t wslice = (size_t)p.nT16 * 64;          // stride between 16-channel slices
  const float4* wbase = p.wfrag + (size_t)nt0 * 64 + lane;
  int wnoff[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) wnoff[n] = (nt0 + n < p.nT16) ? n * 64 : 0;
  float4 wv[NT];
  if (nvalid) {
#pragma unroll
    for (int n = 0; n < NT; ++n) wv[n] = wbase[wnoff[n]];
  }

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
      const float4* wc = wbase + (size_t)c * wslice;
      const float4* wcn = wbase + (size_t)((it + 1 < nIter) ? (it + 1) % p.nC16 : c) * wslice;
      const float4* pl = patch + g * p.planeF4;
      int tr = 0, ts = 0;
#pragma unroll 1
      for (int tap = 0; tap < KS * KS; ++tap) {
        const int toff = tr * p.PW + ts;
        if (++ts == KS) { ts = 0; ++tr; }
        float4 wnx[NT];
        const float4* wnp = (tap + 1 < KS * KS) ? wc + (size_t)(tap + 1) * wtap : wcn;
#pragma unroll
        for (int n = 0; n < NT; ++n) wnx[n] = wnp[wnoff[n]];
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

}  // namespace
