// ALG 7: Winograd F(4x4,3x3) on the fp32 MFMA (3x3 stride-1 convs on planes >= 28x28; hrnet.py:42-58).
//
// 36 position GEMMs M_xi[co][tile] += U_xi[co][ci] V_xi[ci][tile] per 4x4 output tile = 2.25 MFMA-MACs per output
// pixel and input channel instead of 4 for F(2x2,3x3) (ALG 3/4) and 9 for the direct conv: the lever beyond the MFMA
// pipe rate for the 56x56 / 28x28 BasicBlock convs.
//
//   V = B^T d B (6x6 window d),  M = sum_ci U .* V,  Y = A^T M A (4x4 outputs),  U = G g G^T (host, float64)
//
// Shape of this kernel
//   * K is walked in slices of FOUR input channels = one v_mfma_f32_16x16x4_f32 per (position, n-tile): the U fragments
//     of a slice are 36 x NT x 256 B (27 KiB at NT = 3) and the raw patch 16 B per pixel, so several slices fit in LDS
//     at once (a 16-channel slice of F(4x4) would need 108 KiB of U alone).  Lanes carry ONE channel: the window,
//     B^T d B and V are plain floats (12 + 9 + 6 VGPRs instead of their float4 versions).
//   * a tile group = 16 tiles (the MFMA pixel dimension); FOUR waves share a group and own 9 of the 36 positions each
//     (position = 6 xi + nu, wave q owns 9q .. 9q+8: 1.5 rows of the position grid), 9 x NT accumulators per wave;
//     block = 2 groups = 8 waves.  A wave reads the 6x6 window of its lanes' tiles column by column from the LDS
//     patch, forms the two rows of B^T d it needs and its 9 entries of V.
//   * weights: packed as [Cin/4][9 position quads][Cout/16][64 lanes][4] - lane (co, g) holds U[4 quad + j][co][4 c4 + g]
//     for j = 0..3, so one ds_read_b128 feeds the A operands of four positions.
//   * after the K loop a wave reduces its (partial) rows along the columns (Z = M A, 2 x 4 values per n-tile); the four
//     waves of a group exchange Z through LDS one n-tile at a time and wave q finishes n-tile q: Y = A^T Z, bias,
//     residual, activation, 16 pixels x 4 channels per lane.
//   * staging: two 4-channel slices (a "pair") per barrier; patch and fragments of pair m+1 are streamed by LDS-DMA
//     (global_load_lds_dwordx4: 64 patch positions x 16 B, or one 1 KiB fragment, per wave instruction) into the other
//     half of a double buffer while pair m computes; every wave issues 1/8 of the pieces and waits for its own
//     (s_waitcnt vmcnt(0)) before the barrier that publishes the pair.
//   * blocks are persistent over (slab group, n-tile group) items; the first pair of the next item streams in during the
//     Z exchange / epilogue of the current one (no measurable gain yet: the epilogue's 16 stores per lane dominate the
//     fixed cost).
// State: correct and pipelined; no XCD-aware walk, epilogue not overlapped with the next item's compute.
#include "conv_wino4_common.h"

namespace {

struct W4Params {
  const float* in;
  const float* res;
  float* out;
  const float4* ufrag;   // [Cin/4][9][Cout16/16][64] float4
  const float* bias;
  int B, H, W, nC4, nT16;
  int in_rs, in_ss, res_rs, out_rs, out_ss;
  int R, NI, S, nbands, TX, PR, PW, npos, rawF4, tiles_per_slab;
  int nblocks_m, nb_n;   // work items walked by the persistent blocks: slab groups x n-tile groups
  int act, res_after_act;
  FastDiv dPW, dSlab, dBands, dTX, dTslab;
};

using w4::at_c;
using w4::bt_row;

#ifndef W4_EXP
#define W4_EXP 0      // timing probes (tools/build_exp.sh conv_wino4.hip W4_EXP n): 1 no window reads/transform, 2 no U reads,
#endif                // 4 no DMA after the first pair, 8 no MFMAs; non-zero values compute garbage
constexpr int W4_MAXP = 2;      // 64-position patch pieces per wave (npos <= 2 * 8 * 64)

__device__ float4 g_zero_page_w4[1];   // 16 B of zeros: source of the padding lanes

__device__ __forceinline__ void w4_dma16(const void* gsrc, unsigned lds_dst) { w4::dma16(gsrc, lds_dst); }

template <int NT, int Q>
__device__ __forceinline__ void wino4_wave(const W4Params& p, float4* smem, int grp, int lane, int wave) {
  constexpr int P0 = 9 * Q;                 // first position of this wave
  constexpr int RA = P0 / 6;                // its two position rows: RA (from column P0 % 6 on) and RA + 1
  const int idx = lane & 15, g = lane >> 4;
  // LDS: [2 buffers][2 slices]{ raw: rawF4 float4 (4 channels of one patch position each) | U: [9][NT][64] float4 }
  // patch position pos lives in float4 slot pos + pos/8 (one unused slot after every 8): tiles are 4 pixels = 64 B
  // apart, so without the skew the 16 tile lanes of a window read hit 2 of the 8 four-bank groups (8-way conflict)
  const int rawF4 = p.rawF4;
  const int sliceF4 = rawF4 + 9 * NT * 64;
  const int bufF4 = max(2 * sliceF4, 8 * 2 * 4 * 64);      // a stage buffer also hosts the 64 KiB exchange area
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float4*)smem;

  // Persistent: the block walks work items w = blockIdx.x, + gridDim.x, ... (item = slab group x n-tile group); the
  // slice pairs of consecutive items form ONE pipeline (running pair counter `it` selects the LDS buffer), so the first
  // pair of item k+1 streams in during the exchange / epilogue of item k.
  const int nitems = p.nblocks_m * p.nb_n;
  int it = 0;
  auto decode_goff = [&](int item, int* go) {
    const int s0g = (item % p.nblocks_m) * p.NI;
#pragma unroll
    for (int k = 0; k < W4_MAXP; ++k) {
      go[k] = -1;
      const uint32_t slot = (uint32_t)((wave + 8 * k) * 64 + lane);    // piece wave + 8k, lane = slot inside the piece
      const uint32_t k9 = __umulhi(slot, 477218589u);                  // slot / 9 (exact for slot < 2^16)
      const uint32_t r9 = slot - 9 * k9;
      const uint32_t pos = 8 * k9 + r9;
      if (r9 < 8 && pos < (uint32_t)p.npos) {
        const uint32_t psl = fdiv(pos, p.dSlab);
        const uint32_t prem = pos - psl * (uint32_t)(p.PR * p.PW);
        const uint32_t prow = fdiv(prem, p.dPW);
        const int pcol = (int)(prem - prow * (uint32_t)p.PW);
        const uint32_t ps = (uint32_t)s0g + psl;
        const uint32_t pb = fdiv(ps, p.dBands);
        const int pband = (int)(ps - pb * (uint32_t)p.nbands);
        const int iy = pband * p.R - 1 + (int)prow, ix = pcol - 1;
        if (ps < (uint32_t)p.S && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
          go[k] = (int)((pb * (uint32_t)p.H + (uint32_t)iy) * (uint32_t)p.in_rs) + ix * 16;
      }
    }
  };
  const int npieces_raw = rawF4 >> 6;
  auto issue_pair = [&](int m, int buf, const int* go, int nt0i) {      // slices 2m, 2m+1 of an item -> buffer buf
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c4 = 2 * m + h;
      const unsigned sb = lds_base + (unsigned)(buf * bufF4 + h * sliceF4) * 16u;
      const size_t coff = (size_t)(c4 >> 2) * p.in_ss + (c4 & 3) * 4;
#pragma unroll
      for (int k = 0; k < W4_MAXP; ++k) {
        const int piece = wave + 8 * k;
        if (piece < npieces_raw) {
          const void* src = go[k] >= 0 ? (const void*)(p.in + go[k] + coff) : (const void*)g_zero_page_w4;
          w4_dma16(src, (unsigned)__builtin_amdgcn_readfirstlane((int)(sb + (unsigned)piece * 1024u)));
        }
      }
      for (int i = wave; i < 9 * NT; i += 8) {
        const int quad = i / NT, n = i - quad * NT;
        const float4* src = p.ufrag + (((size_t)c4 * 9 + quad) * p.nT16 + min(nt0i + n, p.nT16 - 1)) * 64 + lane;
        w4_dma16(src, (unsigned)__builtin_amdgcn_readfirstlane((int)(sb + (unsigned)(rawF4 + i * 64) * 16u)));
      }
    }
  };
  int goff[W4_MAXP], goffN[W4_MAXP];
  // XCD-aware walk: workgroup b runs on XCD b % 8 (round-robin dispatch), so XCD x owns the contiguous item range
  // [x*per, (x+1)*per) and its gridDim/8 blocks stride through it: neighbouring row bands (shared halo rows) meet in
  // one L2 instead of eight (W4_EXP & 512: plain walk)
  int first = blockIdx.x, istep = gridDim.x, iend = nitems;
  if (!(W4_EXP & 512) && (gridDim.x & 7) == 0 && nitems >= (int)gridDim.x) {
    const int per = (nitems + 7) >> 3;
    first = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    istep = gridDim.x >> 3;
    iend = min(nitems, ((int)(blockIdx.x & 7) + 1) * per);
  }
  if (first >= iend) return;
  decode_goff(first, goff);
  issue_pair(0, 0, goff, (first / p.nblocks_m) * NT);
  for (int item = first; item < iend; item += istep) {
  const int nt0 = (item / p.nblocks_m) * NT;
  const int inext = item + istep;
  const bool has_next = inext < iend;
  if (has_next) decode_goff(inext, goffN);
  // ---- this lane's tile -------------------------------------------------------------------------------------------
  const int s0 = (item % p.nblocks_m) * p.NI;
  const uint32_t tidx = (uint32_t)(grp * 16 + idx);
  const uint32_t sl = fdiv(tidx, p.dTslab);
  const uint32_t rem = tidx - sl * (uint32_t)p.tiles_per_slab;
  const uint32_t tyl = fdiv(rem, p.dTX);
  const int tx = (int)(rem - tyl * (uint32_t)p.TX);
  const uint32_t s = (uint32_t)s0 + sl;
  const uint32_t b = fdiv(s, p.dBands);
  const int band = (int)(s - b * (uint32_t)p.nbands);
  const int oy0 = band * p.R + 4 * (int)tyl;
  const bool tvalid = sl < (uint32_t)p.NI && s < (uint32_t)p.S && oy0 < p.H;
  const int base = tvalid ? (int)((sl * (uint32_t)p.PR + 4 * tyl) * (uint32_t)p.PW) + 4 * tx : 0;   // window top-left in the patch

  int woff[6][6];                                            // float offsets of this lane's 36 window elements in a slice
#pragma unroll
  for (int k = 0; k < 6; ++k)
#pragma unroll
    for (int sc = 0; sc < 6; ++sc) {
      const int pos = base + k * p.PW + sc;
      woff[k][sc] = (pos + (pos >> 3)) * 4 + g;
    }
  f32x4 acc[9][NT];
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int npairs = p.nC4 >> 1;
  for (int m = 0; m < npairs; ++m, ++it) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of this pair have landed (and its stores retired)
    __syncthreads();                                        // ... everybody's; and everybody is done with the previous pair
    // the two waves of a SIMD (same quarter of the two tile groups) issue their DMA share at different points of the pair,
    // so that one of them keeps the MFMA pipe busy while the other sits in the DMA issue
    // what streams in next: the following pair of this item, or the first pair of the block's next item
    const bool last = m + 1 == npairs;
    const bool more = (!last || has_next) && !(W4_EXP & 4);
    auto issue_next = [&]() {
      if (!last) issue_pair(m + 1, (it + 1) & 1, goff, nt0);
      else issue_pair(0, (it + 1) & 1, goffN, (inext / p.nblocks_m) * NT);
    };
    if (more && (grp == 0 || (W4_EXP & 16))) issue_next();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (h == 1 && more && grp == 1 && !(W4_EXP & 16)) issue_next();
      const float4* sbuf = smem + (size_t)(it & 1) * bufF4 + (size_t)h * sliceF4;
      const float* rawf = reinterpret_cast<const float*>(sbuf);
      const float4* ul = sbuf + rawF4;
      // ---- two rows of B^T d from the window, column by column -------------------------------------------------------------
      float t[2][6];
#pragma unroll
      for (int sc = 0; sc < 6; ++sc) {
        float d[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) d[k] = (W4_EXP & 1) ? (float)(k + sc + h) : rawf[woff[k][sc]];
        t[0][sc] = bt_row<RA>(d[0], d[1], d[2], d[3], d[4], d[5]);
        t[1][sc] = bt_row<RA + 1>(d[0], d[1], d[2], d[3], d[4], d[5]);
      }
      // ---- this wave's 9 entries of V and their position GEMMs (one MFMA each per n-tile) ----------------------------------------
      float4 uq[4][NT];                                     // the (up to 4) position quads this wave touches
      constexpr int QD0 = P0 / 4, NQD = (P0 + 8) / 4 - QD0 + 1;
#pragma unroll
      for (int qd = 0; qd < NQD; ++qd)
#pragma unroll
        for (int n = 0; n < NT; ++n) uq[qd][n] = (W4_EXP & 2) ? make_float4(1.f + qd, 2.f + n, 3.f + h, 4.f) : ul[((QD0 + qd) * NT + n) * 64 + lane];
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const int pos = P0 + i;                             // compile-time after unrolling
        const int xi = pos / 6, nu = pos - xi * 6;
        const float* tr = t[xi - RA];
        float v;
        switch (nu) {
          case 0: v = bt_row<0>(tr[0], tr[1], tr[2], tr[3], tr[4], tr[5]); break;
          case 1: v = bt_row<1>(tr[0], tr[1], tr[2], tr[3], tr[4], tr[5]); break;
          case 2: v = bt_row<2>(tr[0], tr[1], tr[2], tr[3], tr[4], tr[5]); break;
          case 3: v = bt_row<3>(tr[0], tr[1], tr[2], tr[3], tr[4], tr[5]); break;
          case 4: v = bt_row<4>(tr[0], tr[1], tr[2], tr[3], tr[4], tr[5]); break;
          default: v = bt_row<5>(tr[0], tr[1], tr[2], tr[3], tr[4], tr[5]); break;
        }
        const int qd = pos / 4 - QD0, cj = pos & 3;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const float4 a4 = uq[qd][n];
          const float a = cj == 0 ? a4.x : cj == 1 ? a4.y : cj == 2 ? a4.z : a4.w;
          if (W4_EXP & 8) acc[i][n][0] += a * v;
          else acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, v, acc[i][n], 0, 0, 0);
        }
      }
    }
  }

  // ---- Z = M A (partial over this wave's columns), exchanged one n-tile at a time; wave Q finishes n-tile Q ------------------
  // exchange area [8 waves][2 rows][4 cols][64] float4 (64 KiB) inside the stage buffer of the item's LAST pair (dead
  // now); the other stage buffer is receiving the next item's first pair
  const int oyb = oy0, oxb = 4 * tx;
  float4* xch = smem + (size_t)((it - 1) & 1) * bufF4;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    __syncthreads();                                         // K loop / previous round finished everywhere
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 9; ++i) {
          const int pos = P0 + i, xi = pos / 6, nu = pos - xi * 6;
          if (xi - RA == r && at_c(j, nu) != 0.f) z += at_c(j, nu) * acc[i][n];
        }
        xch[((wave * 2 + r) * 4 + j) * 64 + lane] = make_float4(z[0], z[1], z[2], z[3]);
      }
    __syncthreads();
    if (nt0 + n < p.nT16) {                                  // every wave finishes output row Q of the 4x4 blocks of n-tile n
      // rows of Z: 0 = q0.r0 | 1 = q0.r1 + q1.r0 | 2 = q1.r1 | 3 = q2.r0 | 4 = q2.r1 + q3.r0 | 5 = q3.r1
      const int w0 = grp * 4;
      auto ld = [&](int q, int r, int j) {
        const float4 z = xch[(((w0 + q) * 2 + r) * 4 + j) * 64 + lane];
        return (f32x4){z.x, z.y, z.z, z.w};
      };
      const float4 sh = *reinterpret_cast<const float4*>(p.bias + (nt0 + n) * 16 + g * 4);
      const float lo = p.act == 1 ? 0.f : -INFINITY;          // ReLU as a clamp: no branch in the store loop
      const bool has_res = p.res != nullptr;
      // per-row output offsets (clamped: dead pixels load/compute harmlessly and are masked at the store only)
      constexpr int i = Q;
      const int oy = oyb + i;
      const bool oky = tvalid && oy < p.H;
      const size_t orow = (size_t)(tvalid ? b : 0u) * p.H + min(oy, p.H - 1);   // lanes without a tile: residual loads stay inside the tensor
      const size_t ooff = orow * p.out_rs + (size_t)(nt0 + n) * p.out_ss + g * 4;
      const size_t roff = orow * p.res_rs + (size_t)(nt0 + n) * p.out_ss + g * 4;
      float4 rr[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) rr[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (has_res) {                                          // wave-uniform; the four loads of the row are issued together
#pragma unroll
        for (int j = 0; j < 4; ++j) rr[j] = *reinterpret_cast<const float4*>(p.res + roff + min(oxb + j, p.W - 1) * 16);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 zr[6];
        zr[0] = ld(0, 0, j); zr[1] = ld(0, 1, j) + ld(1, 0, j); zr[2] = ld(1, 1, j);
        zr[3] = ld(2, 0, j); zr[4] = ld(2, 1, j) + ld(3, 0, j); zr[5] = ld(3, 1, j);
        const int ox = oxb + j;
        const int xo = min(ox, p.W - 1) * 16;
        f32x4 v = (f32x4){sh.x, sh.y, sh.z, sh.w};
#pragma unroll
        for (int k = 0; k < 6; ++k)
          if (at_c(i, k) != 0.f) v += at_c(i, k) * zr[k];
        const float4 r = rr[j];
        if (!p.res_after_act) { v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w; }
        v[0] = fmaxf(v[0], lo); v[1] = fmaxf(v[1], lo); v[2] = fmaxf(v[2], lo); v[3] = fmaxf(v[3], lo);
        if (p.res_after_act) { v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w; }
        if ((W4_EXP & 128) && (v[0] != 12345.f || j > 0)) continue;       // probe: compute everything, store (almost) nothing
        if (oky && ox < p.W) *reinterpret_cast<float4*>(p.out + ooff + xo) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  }
  if (has_next) {
#pragma unroll
    for (int k = 0; k < W4_MAXP; ++k) goff[k] = goffN[k];
  }
  }   // item loop
}

template <int NT>
__global__ void __launch_bounds__(512)
conv_wino4_kernel(const W4Params p) {
  extern __shared__ float4 smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave >> 2, q = wave & 3;
  if (q == 0) wino4_wave<NT, 0>(p, smem, grp, lane, wave);
  else if (q == 1) wino4_wave<NT, 1>(p, smem, grp, lane, wave);
  else if (q == 2) wino4_wave<NT, 2>(p, smem, grp, lane, wave);
  else wino4_wave<NT, 3>(p, smem, grp, lane, wave);
}

struct W4Geo { int R, NI, nbands, S, TX, PR, PW, npos, rawF4, tps; };
bool w4geo(const ConvDesc& d, const ConvCfg& cfg, W4Geo* g) {
  if (d.ks != 3 || d.stride != 1 || cfg.NT < 1 || cfg.NT > 3 || cfg.WM != 2 || cfg.WN != 4 || d.Cin % 16 || d.Cout % 16) return false;
  if (cfg.R < 4 || (cfg.R & 3) || cfg.NI < 1) return false;
  g->TX = (d.W + 3) / 4;
  const int Hc = (d.H + 3) / 4 * 4;
  g->R = std::min(cfg.R, Hc); g->NI = cfg.NI;
  g->nbands = (d.H + g->R - 1) / g->R;
  if (g->NI > 1 && g->nbands > 1) return false;               // several slabs per block only for whole images
  g->S = d.B * g->nbands;
  g->tps = (g->R / 4) * g->TX;
  if (g->NI * g->tps > 32) return false;                      // two groups of 16 tiles
  g->PR = g->R + 2; g->PW = 4 * g->TX + 2;
  g->npos = g->NI * g->PR * g->PW;
  g->rawF4 = (g->npos + g->npos / 8 + 1 + 63) & ~63;          // skewed slots, whole 64-slot DMA pieces
  if (g->rawF4 > W4_MAXP * 512) return false;
  if (2 * std::max<size_t>(2 * ((size_t)g->rawF4 + 9 * cfg.NT * 64), 8 * 2 * 4 * 64) * sizeof(float4) > 160 * 1024) return false;
  if ((long)d.B * d.H * d.W * std::max(std::max(d.in_cs, d.out_cs), d.res_cs) >= (1L << 31)) return false;
  return true;
}

}  // namespace

// packed fragments for ALG 7: [Cin/4][9 quads][Cout16/16][64 lanes][4]; lane = g*16 + co_l, value j = U[4 quad + j][co][4 c4 + g] * scale[co]
size_t conv_wino4_packed_floats(int Cin, int Cout16) { return (size_t)36 * Cin * Cout16; }
void conv_wino4_pack_weights(const float* w_oihw, const float* scale, int Cout, int Cin, int Cout16, float* dst) {
  std::vector<double> u;
  w4::u_transform(w_oihw, Cout, Cin, &u);
  const int nC4 = Cin / 4, nT16 = Cout16 / 16;
  for (int c4 = 0; c4 < nC4; ++c4)
    for (int quad = 0; quad < 9; ++quad)
      for (int nt = 0; nt < nT16; ++nt)
        for (int lane = 0; lane < 64; ++lane) {
          const int g = lane >> 4, co = nt * 16 + (lane & 15), ci = 4 * c4 + g;
          float* o = dst + ((((size_t)c4 * 9 + quad) * nT16 + nt) * 64 + lane) * 4;
          for (int j = 0; j < 4; ++j)
            o[j] = co < Cout ? (float)(u[((size_t)(4 * quad + j) * Cout + co) * Cin + ci] * (scale ? (double)scale[co] : 1.0)) : 0.f;
        }
}

// cfg: {MT = 1, NT (1..3), WM = 2 tile groups, WN = 4 position quarters, R = output rows per slab (multiple of 4), NI, ALG = 7}
size_t conv_wino4_lds_bytes(const ConvDesc& d, const ConvCfg& cfg) {
  W4Geo g;
  if (!w4geo(d, cfg, &g)) return 0;
  // 2 stage buffers of 2 slices each; a buffer also hosts the 64 KiB exchange area after its item's last pair
  const size_t buf = std::max<size_t>(2 * ((size_t)g.rawF4 + 9 * cfg.NT * 64), 8 * 2 * 4 * 64);
  return 2 * buf * sizeof(float4);
}

int conv_wino4_launch(const ConvDesc& d, const ConvCfg& cfg, hipStream_t stream) {
  W4Geo g;
  if (!w4geo(d, cfg, &g) || !d.wfrag_wino4) {
    poco_set_error("conv(winograd 4x4): needs ks = 3, stride 1, NT 1..3, WM = 2, WN = 4, R % 4 == 0, NI*(R/4)*ceil(W/4) <= 32 tiles and the ALG 7 weight fragments");
    return POCO_ERR_ARG;
  }
  if (d.act == 3 || d.act == 2) { poco_set_error("conv(winograd 4x4): activation must be none or ReLU"); return POCO_ERR_ARG; }
  W4Params p{};
  p.in = d.in + l16_chan_off(d.in_co, d.W);
  p.res = d.res ? d.res + l16_chan_off(d.res_co, d.W) : nullptr;
  p.out = d.out + l16_chan_off(d.out_co, d.W);
  p.ufrag = reinterpret_cast<const float4*>(d.wfrag_wino4); p.bias = d.bias;
  p.B = d.B; p.H = d.H; p.W = d.W; p.nC4 = d.Cin / 4; p.nT16 = d.Cout / 16;
  p.in_rs = d.in_cs * d.W; p.in_ss = d.W * 16; p.res_rs = d.res_cs * d.W; p.out_rs = d.out_cs * d.W; p.out_ss = d.W * 16;
  p.R = g.R; p.NI = g.NI; p.S = g.S; p.nbands = g.nbands; p.TX = g.TX; p.PR = g.PR; p.PW = g.PW; p.npos = g.npos; p.rawF4 = g.rawF4;
  p.tiles_per_slab = g.tps;
  p.act = d.act; p.res_after_act = d.res_after_act;
  p.dPW = make_fastdiv(g.PW); p.dSlab = make_fastdiv(g.PR * g.PW); p.dBands = make_fastdiv(g.nbands);
  p.dTX = make_fastdiv(g.TX); p.dTslab = make_fastdiv(g.tps);
  const size_t lds = conv_wino4_lds_bytes(d, cfg);
  p.nblocks_m = (g.S + g.NI - 1) / g.NI; p.nb_n = (p.nT16 + cfg.NT - 1) / cfg.NT;
  // balanced persistent grid: every block walks the same number of items (one block per CU: the stage buffers take ~152 KiB)
  const long items = (long)p.nblocks_m * p.nb_n;
  const long rounds = (items + 255) / 256;
  long g4 = (items + rounds - 1) / rounds;
  if (g4 > 8) g4 = std::min(256L, (g4 + 7) / 8 * 8);           // multiple of 8 for the XCD-aware walk
  const dim3 grid((unsigned)g4, 1);
  auto fn = cfg.NT == 3 ? conv_wino4_kernel<3> : cfg.NT == 2 ? conv_wino4_kernel<2> : conv_wino4_kernel<1>;
  if (lds > 64 * 1024) {
    static thread_local bool configured[4] = {false, false, false, false};
    if (!configured[cfg.NT]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) { poco_set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return POCO_ERR_HIP; }
      configured[cfg.NT] = true;
    }
  }
  hipLaunchKernelGGL(fn, grid, dim3(512), lds, stream, p);
  POCO_HIP_CHECK(hipGetLastError());
  return POCO_OK;
}
