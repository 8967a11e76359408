// Geometry shared by the Winograd F(4x4,3x3) kernels whose blocks own 32-tile items (conv_wino4p.hip = ALG 8, conv_wino4w.hip =
// ALG 13): kernel parameters, the persistent XCD-aware item walk, rectangular / flat / mosaic items (tile -> patch window, patch
// slot -> image pixel), the packed-fp32 helpers of the input transform and the host-side flat-item geometry.  Included INSIDE the
// anonymous namespace of each kernel file (moved here verbatim from conv_wino4p.hip in round 5).
#pragma once

struct W4PParams {
  const float* in;
  const float* res;
  float* out;
  const float4* ufrag;   // [Cin/4][Cout16/16][9][64] float4 (see above)
  const float* bias;
  int B, H, W, nC4, nT16;
  int in_rs, in_ss, res_rs, out_rs, out_ss;
  int R, NI, S, nbands, TX, PR, PW, npos, rawF4, tiles_per_slab;
  int nblocks_m, nb_n;   // work items walked by the persistent blocks: slab groups x n-tile groups
  int act, res_after_act;
  int uoff, voff, xoff;  // float4 offsets of the U ring, the V double buffer and the exchange area in LDS (raw ring at 0)
  FastDiv dPW, dSlab, dBands, dTX, dTslab;
  // FLAT items (cfg.NI == 0, see flat_geo): TY tile rows per image, ntiles = B * TX * TY tiles in all, fragW = 4 TX + 2
  int TY, ntiles, fragW;
  FastDiv dTY, dFragW;
  // MOSAIC (flat items, cfg.R = 4 MS): MS x MS images form one virtual plane of MS (H + 1) - 1 rows x MS (W + 1) - 1 columns in which
  // neighbouring images share their one-pixel zero border (3x3 conv, pad 1: exactly what the convolution pads with).  A 14 x 14
  // plane alone needs 4 x 4 tiles of 4 x 4 outputs (23 % of them padding); 4 x 4 images side by side need 15 x 15 tiles instead of
  // 16 x 16.  TX / TY / ntiles then count the tiles of the virtual planes; Tile::b is the mosaic index, oy0 / tx are virtual.
  int MS, Hp1, Wp1;
  FastDiv dHp1, dWp1;
  // round 5: items are WALKED n-group-innermost (walk index w = m * nb_n + n-group, so that the n-groups of one tile strip run on
  // neighbouring blocks of one XCD at the same time and read their patch from one L2 - the PARE head's 480 -> 128 conv moved
  // 715 MB per launch, 2.9 x its algorithmic bytes, with the n-group-outermost walk); the helpers below keep taking the id
  // m + n-group * nblocks_m (w4p_item_id)
  FastDiv dNbn;
  int ninner;           // 1: walk n-group-innermost (set per launch: where the patch re-reads are heavy, Cin * nb_n >= 1024)
};
// walk index -> item id of the geometry helpers
__device__ __forceinline__ int w4p_item_id(const W4PParams& p, int w) {
  if (!p.ninner) return w;
  const uint32_t m = fdiv((uint32_t)w, p.dNbn);
  return (int)m + (w - (int)m * p.nb_n) * p.nblocks_m;
}

// virtual coordinate v of a mosaic axis -> (image index along the axis, coordinate inside the image); the border lines between
// images map to coordinate H (W), i.e. "outside"
__device__ __forceinline__ void mosaic_split(uint32_t v, int p1, FastDiv d, int* img, int* c) {
  const uint32_t q = fdiv(v, d);
  *img = (int)q;
  *c = (int)(v - q * (uint32_t)p1);
}


__device__ float4 g_zero_page_w4p[1];   // 16 B of zeros: source of the padding lanes

__device__ __forceinline__ void wait_vm(int n) {     // s_waitcnt vmcnt(n), n wave-uniform
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// the persistent, XCD-aware walk over work items shared by both roles: workgroup b runs on XCD b % 8 (round-robin
// dispatch), so XCD x owns the contiguous item range [x*per, (x+1)*per): neighbouring row bands meet in one L2
struct Walk { int first, step, end; };
__device__ __forceinline__ Walk item_walk(const W4PParams& p) {
  const int nitems = p.nblocks_m * p.nb_n;
  Walk w{(int)blockIdx.x, (int)gridDim.x, nitems};
  if ((gridDim.x & 7) == 0 && nitems >= (int)gridDim.x) {
    const int per = (nitems + 7) >> 3;
    w.first = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    w.step = gridDim.x >> 3;
    w.end = min(nitems, ((int)(blockIdx.x & 7) + 1) * per);
  }
  return w;
}

// tile of lane idx (0..15) in group grp of an item: validity, window top-left in the patch, output coordinates
struct Tile { bool valid; int base, b, oy0, tx; };
// FLAT items (round 4): an item is 32 CONSECUTIVE tiles of the flattened (image, tile row, tile column) order, so every one of the
// 32 MFMA tile columns of a block carries a tile whatever the plane width is (rectangular items use 28 of 32 on 56x56 / 28x28
// planes: 2 x 14 and 4 x 7 tiles).  Its patch is a STRIP of 6 rows: the item's tile-row fragments laid side by side, each with its
// own 2-column halo - fragment 0 = tiles tx0.. of the first tile row (4 (TX - tx0) + 2 columns), fragments r >= 1 = whole tile rows
// (fragW = 4 TX + 2 columns each, the last one used as far as the item reaches): PW = 128 + 2 Fmax columns, window (k, c) of a tile at
// base + k PW + c exactly as in a rectangular patch.
struct FlatItem { int gr0, tx0, w0; };
__device__ __forceinline__ FlatItem flat_item(const W4PParams& p, int item) {
  const uint32_t T0 = (uint32_t)(item % p.nblocks_m) * 32u;
  FlatItem f;
  f.gr0 = (int)fdiv(T0, p.dTX);
  f.tx0 = (int)(T0 - (uint32_t)f.gr0 * (uint32_t)p.TX);
  f.w0 = 4 * (p.TX - f.tx0) + 2;
  return f;
}
template <int FLAT>
__device__ __forceinline__ Tile tile_of(const W4PParams& p, int item, int grp, int idx) {
  if constexpr (FLAT) {
    const FlatItem f = flat_item(p, item);
    const uint32_t T = (uint32_t)(item % p.nblocks_m) * 32u + (uint32_t)(grp * 16 + idx);
    const uint32_t gr = fdiv(T, p.dTX);
    Tile t;
    t.tx = (int)(T - gr * (uint32_t)p.TX);
    t.b = (int)fdiv(gr, p.dTY);
    t.oy0 = 4 * (int)(gr - (uint32_t)t.b * (uint32_t)p.TY);
    t.valid = T < (uint32_t)p.ntiles;
    const int r = (int)gr - f.gr0;
    t.base = r == 0 ? 4 * (t.tx - f.tx0) : f.w0 + (r - 1) * p.fragW + 4 * t.tx;
    if (!t.valid) { t.b = 0; t.oy0 = 0; t.tx = 0; t.base = 0; }
    return t;
  }
  const int s0 = (item % p.nblocks_m) * p.NI;
  const uint32_t tidx = (uint32_t)(grp * 16 + idx);
  const uint32_t sl = fdiv(tidx, p.dTslab);
  const uint32_t rem = tidx - sl * (uint32_t)p.tiles_per_slab;
  const uint32_t tyl = fdiv(rem, p.dTX);
  Tile t;
  t.tx = (int)(rem - tyl * (uint32_t)p.TX);
  const uint32_t s = (uint32_t)s0 + sl;
  t.b = (int)fdiv(s, p.dBands);
  const int band = (int)(s - (uint32_t)t.b * (uint32_t)p.nbands);
  t.oy0 = band * p.R + 4 * (int)tyl;
  t.valid = sl < (uint32_t)p.NI && s < (uint32_t)p.S && t.oy0 < p.H;
  t.base = t.valid ? (int)((sl * (uint32_t)p.PR + 4 * tyl) * (uint32_t)p.PW) + 4 * t.tx : 0;
  // a lane without a tile (beyond the block's slabs / the plane) computes along harmlessly and is masked at the store, but
  // its unconditional residual loads must stay inside the tensor: park it on image 0, row 0
  if (!t.valid) { t.b = 0; t.oy0 = 0; t.tx = 0; }
  return t;
}

__device__ __forceinline__ float f4c(const float4& v, int j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; }

// V slot of (tile t, channel g of the slice) inside a 64-slot operand vector.  The producers work in the lane order g + 4 t (a
// 32-lane half = 8 tiles x 4 channels: their window reads hit 32 different banks; with the MFMA operand order t + 16 g a half
// is 16 tiles x 2 channels = 2-way conflicts on every read) and store to slot sigma; the MFMA waves (lane = t + 16 g) read slot
// sigma.  sigma = 4 t + ((g + f(t / 4)) & 3), f = (0, 0, 2, 2): contiguous for 8 consecutive producer lanes (ds_write_b128) and
// conflict-free for the four 16-lane groups of the MFMA waves' ds_read_b128 and the halves of their ds_read_b32.
__device__ __forceinline__ int w4p_sigma(int t, int g) { return 4 * t + ((g + ((t >> 2) & 2)) & 3); }

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, float k, f32x2 c) { return __builtin_elementwise_fma(a, (f32x2){k, k}, c); }
__device__ __forceinline__ f32x2 pk_fma2(f32x2 a, f32x2 k, f32x2 c) { return __builtin_elementwise_fma(a, k, c); }


// ---- staging (LDS-DMA) shared by whoever issues it: `nw` waves take pieces w, w + nw, ... -------------------------------
// global float offsets of this wave's raw-patch pieces (lane = slot inside the piece), -1 = padding / beyond the patch
template <int MAXP, int FLAT>
__device__ __forceinline__ void raw_piece_offsets(const W4PParams& p, int item, int w, int nw, int lane, int* goff) {
  if constexpr (FLAT) {
    // strip position -> (fragment, column) -> (image, row, column); slots are skewed by pos / 16 here (a 6-row strip of up to 146
    // columns has to fit the raw ring next to NT = 3 U slots: 840 positions -> 896 slots with pos / 16, 960 with pos / 8)
    const FlatItem f = flat_item(p, item);
    const uint32_t Tend = min((uint32_t)(item % p.nblocks_m) * 32u + 31u, (uint32_t)p.ntiles - 1u);
    const uint32_t gr_end = fdiv(Tend, p.dTX);
    const int r_end = (int)gr_end - f.gr0, tx_end = (int)(Tend - gr_end * (uint32_t)p.TX);
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
      goff[k] = -1;
      const uint32_t slot = (uint32_t)((w + nw * k) * 64 + lane);
      const uint32_t k17 = __umulhi(slot, 252645136u);               // slot / 17 (exact for slot < 2^28)
      const uint32_t r17 = slot - 17 * k17;
      uint32_t pos = 16 * k17 + r17;
      asm volatile("" : "+v"(pos));                                    // (see below: keeps the item-invariant part out of the K loop's registers)
      if (r17 < 16 && pos < (uint32_t)p.npos) {
        const uint32_t prow = fdiv(pos, p.dPW);
        const int col = (int)(pos - prow * (uint32_t)p.PW);
        int r, cx, txs;
        if (col < f.w0) { r = 0; cx = col; txs = f.tx0; }
        else {
          const uint32_t c2 = (uint32_t)(col - f.w0);
          const uint32_t q = fdiv(c2, p.dFragW);
          r = 1 + (int)q; cx = (int)(c2 - q * (uint32_t)p.fragW); txs = 0;
        }
        const uint32_t gr = (uint32_t)(f.gr0 + r);
        const uint32_t pb = fdiv(gr, p.dTY);
        const int ty = (int)(gr - pb * (uint32_t)p.TY);
        int iy = 4 * ty - 1 + (int)prow, ix = 4 * txs + cx - 1;
        // columns the item's tiles of this fragment do not read are left out of the DMA (they stay zero)
        const bool used = r < r_end || (r == r_end && cx < 4 * (tx_end - txs + 1) + 2);
        uint32_t img = pb;
        bool inimg = true;
        if constexpr (FLAT == 2) {                        // virtual -> (image of the mosaic, pixel); border lines and absent images read as zero
          int my = 0, mx = 0;
          if (iy >= 0 && ix >= 0) {
            mosaic_split((uint32_t)iy, p.Hp1, p.dHp1, &my, &iy);
            mosaic_split((uint32_t)ix, p.Wp1, p.dWp1, &mx, &ix);
          }
          img = (pb * (uint32_t)p.MS + (uint32_t)my) * (uint32_t)p.MS + (uint32_t)mx;
          inimg = my < p.MS && mx < p.MS && img < (uint32_t)p.B;
        }
        if (used && inimg && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
          goff[k] = (int)((img * (uint32_t)p.H + (uint32_t)iy) * (uint32_t)p.in_rs) + ix * 16;
      }
    }
    return;
  }
  const int s0g = (item % p.nblocks_m) * p.NI;
#pragma unroll
  for (int k = 0; k < MAXP; ++k) {
    goff[k] = -1;
    const uint32_t slot = (uint32_t)((w + nw * k) * 64 + lane);
    const uint32_t k9 = __umulhi(slot, 477218589u);                  // slot / 9 (exact for slot < 2^16)
    const uint32_t r9 = slot - 9 * k9;
    uint32_t pos = 8 * k9 + r9;
    // opaque: the patch coordinates of a slot do not depend on the item, and hipcc hoists them (prow - 1, 16 ix, ... as 64-bit
    // v_mad_u64_u32 addends) out of the item loop - eight registers that lived across the K loop and were spilled to scratch at
    // NT = 3 and reloaded with a full vmcnt(0) wait at every item start (28 spilled VGPRs, 36 B of scratch;
    // now 1 and 8 B - threadIdx.x, stored once per kernel).  Recomputing them costs ~20 VALU instructions per item and the launch
    // is 0.5 % faster.  Builds without ANY scratch exist (also make the 4 g of the epilogue addresses opaque, or take the lane id from
    // mbcnt inside each role) and are 2-8 % SLOWER per launch (55.0 / 49.3 / 80.4 against 53.7 / 46.1 / 73.7 us on the three W48
    // shapes): at 166-168 registers every value the allocator cannot park in scratch across the K loop comes out of the operand
    // prefetch depth of the exchange rounds.
    asm volatile("" : "+v"(pos));
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
        goff[k] = (int)((pb * (uint32_t)p.H + (uint32_t)iy) * (uint32_t)p.in_rs) + ix * 16;
    }
  }
}


// FLAT items (cfg.NI == 0, cfg.R = 4 MS, MS = 1 / 2 / 4 / 8 = mosaic side): 32 consecutive tiles per item, strip patch (see tile_of).  Geo fields reused: TX, PW, npos,
// rawF4 (skew pos / 16), S = number of items along m; R = 4, PR = 6; tps / nbands / NI unused.
struct FlatGeo { int TY, ntiles, fragW, MS; };
bool flat_geo(const ConvDesc& d, const ConvCfg& cfg, w4::Geo* g, FlatGeo* f) {
  if (cfg.NI != 0 || (cfg.R != 4 && cfg.R != 8 && cfg.R != 16 && cfg.R != 32)) return false;
  f->MS = cfg.R / 4;                                  // mosaic side: MS x MS images per virtual plane (1 = plain flat items)
  const int VH = f->MS * (d.H + 1) - 1, VW = f->MS * (d.W + 1) - 1;
  const int nmos = (d.B + f->MS * f->MS - 1) / (f->MS * f->MS);
  g->TX = (VW + 3) / 4;
  f->TY = (VH + 3) / 4;
  if ((long)nmos * g->TX * f->TY >= (1L << 24) || VH >= 32768 || VW >= 32768) return false;       // (fdiv is exact for n * d < 2^32)
  f->ntiles = nmos * g->TX * f->TY;
  f->fragW = 4 * g->TX + 2;
  const int fmax = (g->TX - 1 + 32 + g->TX - 1) / g->TX;        // tile-row fragments of an item that starts in the last column
  g->R = 4; g->NI = 0; g->nbands = f->TY; g->PR = 6; g->tps = g->TX;
  g->PW = 128 + 2 * fmax;
  g->npos = 6 * g->PW;
  g->rawF4 = (g->npos + g->npos / 16 + 1 + 63) & ~63;
  g->S = (f->ntiles + 31) / 32;
  if ((long)d.B * d.H * d.W * std::max(std::max(d.in_cs, d.out_cs), d.res_cs) >= (1L << 31)) return false;
  return true;
}
