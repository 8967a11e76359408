// ALG 13 (round 5): Winograd F(4x4,3x3) on the fp32 MFMA with WHOLE-POSITION MFMA waves - no exchange, no item start
// (3x3 stride-1 convs on planes >= 14x14; pocolib/models/backbone/hrnet.py:42-58, hrnet_cls.py BasicBlock convs).
//
// ALG 8 (conv_wino4p.hip) splits the 36 Winograd positions of a 16-tile group over four MFMA waves (9 positions x NT n-tiles
// each: 108 accumulators at the 168-register budget of a 12-wave block), so the output transform A^T M A needs all four
// waves: NT exchange rounds through LDS per item (8 ds_write_b128 + 8 ds_read_b128 + ~110 VALU per wave and round, two
// block-wide barriers each), an exchange area that overlays the V buffers and U ring - so the next item's first slices can
// only be prepared after the rounds - and the s_memtime traces of round 5 put 12-13 k clk of exchange + 2-7 k clk of item
// start next to 33 k clk of K loop on the 56x56 48->48 launch (43 % of a launch is fixed cost).
//
// Here a block is 2 NT + 2 waves (NT = 3: eight, i.e. two per SIMD and a 256-register budget):
//   * MFMA wave (grp, n) owns ALL 36 positions of one 16-tile group for ONE 16-channel n-tile: 36 accumulators (144
//     registers), per 4-channel slice nine quads of { ds_read_b128 V, ds_read_b128 U, 4 MFMAs }.  Its output transform
//     A^T M A is register-only, it applies bias / residual / ReLU and stores its 4x4 pixel blocks itself: no exchange, no
//     barrier, no LDS traffic at the end of an item.
//   * producer wave pw = the input transform V = B^T d B of group pw, all 36 positions per (tile, channel) lane, the packed-fp32
//     scheme of ALG 8 (15 ds_read2_b32 window reads, ~72 v_pk_* instructions, 9 ds_write_b128 in the pair order of the
//     results, slot swizzle w4p_sigma).
//   * the slice pipeline is CONTINUOUS ACROSS ITEMS: global slice t = (item k, slice s) uses U ring slot t % 3, raw ring slot
//     t % 3 and V buffer t & 1; at slice t the MFMA waves request U(t + 2) and raw(t + 4) and the producers build V(t + 1) and
//     read the window of slice t + 2 - whichever item those belong to.  One barrier per slice, nothing else: an item boundary
//     costs the MFMA waves their register-only epilogue and nobody a pipeline refill.  (Padding positions of a patch differ
//     from item to item: the wave that requests the first three slices of an item also zero-fills the padding lanes of the ring
//     slots they go to.)
//   * items are walked n-group-innermost (item = m * nb_n + n-group): the n-groups of one tile strip run on neighbouring
//     blocks of one XCD at the same time and read their patch from one L2.
//
// Weights: U = G g G^T in float64 on the host (BN scale folded), packed per (4-channel slice, n-tile) as one 9 KiB block of nine
// quads [q][64 lanes] float4, lane = (co & 15) + 16 (ci & 3), quad q / element i = position (w4w_row, w4w_nu) - the order in
// which the producer's packed transform leaves its register pairs.
#include "conv_wino4_common.h"
#include <cstdlib>

namespace {

using w4::at_c;

#include "conv_wino4p_geo.h"

#ifndef W4W_DMAW
#define W4W_DMAW 9    // who requests the LDS-DMA of the coming slices (see W4WDuty)
#endif

#ifndef W4W_EXP
#define W4W_EXP 0     // timing probes (results are garbage): 1 no LDS-DMA in the K loop, 2 producers skip transform + window reads, 4 MFMA waves
#endif                // skip their operand reads, 8 MFMA waves skip the MFMAs
#ifndef W4W_DMALAST
#define W4W_DMALAST 0     // (1: 72.5 vs 71.6 us on 14x14 192->192, 855 vs 803 on 480->128 - the requests need their two slices to land)
#endif
#ifndef W4W_WIN2
#define W4W_WIN2 0
#endif
#ifndef W4W_GATHER
#define W4W_GATHER 1  // raw pieces requested as one exec-masked asm block per issuer (w4::dma_gather); 0 = one dma16_sv per piece
#endif
#ifndef W4W_PF
#define W4W_PF 4      // operand quads requested ahead of the MFMAs that use them (PF + 1 register sets of 8).  Same box, us per launch on the
                      // five solo shapes: PF 2 47.5 / 46.8 / 77.5 / 86.7 / 881, PF 4 46.9 / 46.3 / 77.0 / 84.1 / 868, PF 6 48.3 / 47.5 / 78.6 / 87.3 / 890, PF 8 (spills) 52.8 / ...
#endif
#ifndef W4W_HOLD
#define W4W_HOLD 0    // quads of a slice whose MFMAs run behind the slice barrier (see the K loop): 2 is 1-2 % slower than 0 (registers)
#endif
#ifndef W4W_LAYOUT
#define W4W_LAYOUT 1
#endif
#ifndef W4W_TRACE
#define W4W_TRACE 0   // 1: block 0 sums s_memtime phases of its waves over its first item (tools/w4w_trace.py)
#endif
#if W4W_TRACE
__device__ unsigned long long g_w4w_trace[64];
#define W4W_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define W4W_ACC(slot, a, b) do { if (trace) tr[slot] += (b) - (a); } while (0)
#else
#define W4W_T(var)
#define W4W_ACC(slot, a, b)
#endif

constexpr int W4W_UBLK = 9 * 64;     // float4 per (slice, n-tile) block of U; V of one group and slice has the same shape

// position (xi, nu) held by element i of quad q = 3 m + a:  a = 0: row 2m, nu (0 5 1 3) | a = 1: row 2m nu (2 4), row 2m+1 nu (0 5)
// | a = 2: row 2m+1, nu (1 3 2 4) - the register pairs (nu0, nu5), (nu1, nu3), (nu2, nu4) of the packed input transform, row by row
__host__ __device__ constexpr int w4w_row(int q, int i) { return 2 * (q / 3) + ((q % 3) == 0 ? 0 : (q % 3) == 1 ? (i < 2 ? 0 : 1) : 1); }
__host__ __device__ constexpr int w4w_nu(int q, int i) {
  constexpr int E[3][4] = {{0, 5, 1, 3}, {2, 4, 0, 5}, {1, 3, 2, 4}};
  return E[q % 3][i];
}
// inverse: which accumulator (4 q + i) holds position (xi, nu)
__host__ __device__ constexpr int w4w_slot(int xi, int nu) {
  const int m = xi / 2;
  if ((xi & 1) == 0) {
    if (nu == 2 || nu == 4) return 4 * (3 * m + 1) + (nu == 2 ? 0 : 1);
    return 4 * (3 * m) + (nu == 0 ? 0 : nu == 5 ? 1 : nu == 1 ? 2 : 3);
  }
  if (nu == 0 || nu == 5) return 4 * (3 * m + 1) + (nu == 0 ? 2 : 3);
  return 4 * (3 * m + 2) + (nu == 1 ? 0 : nu == 3 ? 1 : nu == 2 ? 2 : 3);
}

__device__ __forceinline__ void w4w_wait_vm(int n) {     // s_waitcnt vmcnt(n), n wave-uniform, 0 .. 31
#define W4W_WVM(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n) {
    W4W_WVM(0) W4W_WVM(1) W4W_WVM(2) W4W_WVM(3) W4W_WVM(4) W4W_WVM(5) W4W_WVM(6) W4W_WVM(7) W4W_WVM(8) W4W_WVM(9) W4W_WVM(10) W4W_WVM(11)
    W4W_WVM(12) W4W_WVM(13) W4W_WVM(14) W4W_WVM(15) W4W_WVM(16) W4W_WVM(17) W4W_WVM(18) W4W_WVM(19) W4W_WVM(20) W4W_WVM(21) W4W_WVM(22)
    W4W_WVM(23) W4W_WVM(24) W4W_WVM(25) W4W_WVM(26) W4W_WVM(27) W4W_WVM(28) W4W_WVM(29) W4W_WVM(30) W4W_WVM(31)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef W4W_WVM
}

// y = A^T x for the six values x0 .. x5 of one transform row / column ([Lavin & Gray]: A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0;
// 0 1 -1 8 -8 1]) in even / odd form: 10 operations
__device__ __forceinline__ void w4w_at(const f32x4& x0, const f32x4& x1, const f32x4& x2, const f32x4& x3, const f32x4& x4, const f32x4& x5,
                                       f32x4 (&y)[4]) {
  const f32x4 s1 = x1 + x2, d1 = x1 - x2, s2 = x3 + x4, d2 = x3 - x4;
  y[0] = x0 + s1 + s2;
  y[1] = d1 + 2.f * d2;
  y[2] = s1 + 4.f * s2;
  y[3] = d1 + 8.f * d2 + x5;
}

struct W4WParams {
  W4PParams g;          // geometry, tensors, LDS offsets (xoff unused)
  FastDiv dNbn;         // item -> (tile strip m, n-group): n-group innermost
};

// walk index -> the item id the shared geometry helpers expect (m + n-group * nblocks_m)
__device__ __forceinline__ int w4w_item_id(const W4WParams& p, int it, int* ngroup) {
  const uint32_t m = fdiv((uint32_t)it, p.dNbn);
  *ngroup = it - (int)m * p.g.nb_n;
  return (int)m + *ngroup * p.g.nblocks_m;
}

// ---------------------------------------------------------------------------------------------------------------------
// LDS-DMA duty (global_load_lds_dwordx4) of the coming slices, shared by NW issuing waves: issuer dw takes raw pieces dw, dw + NW, ...
// and U pieces dw, dw + NW, ... of every slice.  Who issues (W4W_DMAW): an LDS-DMA piece costs the issuing wave ~100 clk of issue
// time and blocks the MFMA pipe of its SIMD for ~47 clk (s_memtime traces), and with 2 NT MFMA waves + 2 producers the
// SIMDs are unevenly loaded - NT = 3: SIMDs 0 / 1 carry two MFMA waves (72 MFMAs = 2304 clk per slice, the floor of the kernel),
// SIMDs 2 / 3 one MFMA wave + one producer.
// ---------------------------------------------------------------------------------------------------------------------

// WHO requests what (W4W_DMAW) and WHERE the producers sit (W4W_LAYOUT).  The 9 NT U pieces of a slice are dealt round-robin to
// NWU issuers, the rawF4 / 64 patch pieces to NWR issuers; an issuer is an MFMA wave (by its MFMA index m = 0 .. 2 NT - 1) or a
// producer.  Waves land on SIMD (wave id % 4):
//   W4W_LAYOUT 0: MFMA waves = wave ids 0 .. 2 NT - 1, producers = the last two.  NT = 3: SIMDs 0 / 1 carry two MFMA waves, SIMDs 2 / 3
//                 one MFMA wave + one producer.
//   W4W_LAYOUT 1 (NT = 3 only): producers = wave ids 3 and 7, i.e. BOTH on SIMD 3, and every other SIMD carries two MFMA waves:
//                 the MFMA work is balanced over three SIMDs (72 MFMAs = 2304 clk per slice each) and the producers' VALU / LDS /
//                 LDS-DMA instructions have a SIMD of their own (fp32 MFMAs run on the SIMD's vector ALUs; an LDS-DMA piece blocks the
//                 MFMA pipe of the SIMD it is issued from for ~47 clk - on SIMD 3 there is none).
//   mode   U issuers                 raw issuers
//   0      MFMA 2, 3 + producers     MFMA 2, 3 + producers
//   1      all MFMA                  all MFMA
//   6      producers                 all MFMA
//   7      producers                 producers
//   8      all MFMA                  producers
//   9      NT = 3: mode 7 (the producers request everything: they own SIMD 3 in layout 1 and the three MFMA SIMDs do nothing but MFMAs and
//          operand reads - 14x14 192->192 76.8 -> 71.3 us, 28x28 96->96 46.4 -> 44.5, 480->128 882 -> 807); NT < 3: mode 6 (the producers
//          share their SIMDs with MFMA waves there: 56x56 64->64 83.7 us against 99.7 with mode 7)
struct W4WPlan { int mu0, nmu, kpu, mr0, nmr, kpr; };
template <int NT> constexpr W4WPlan w4w_plan() {
  constexpr int M = 2 * NT;
  if (W4W_DMAW == 1) return {0, M, 0, 0, M, 0};
  if (W4W_DMAW == 6) return {0, 0, 1, 0, M, 0};
  if (W4W_DMAW == 7) return {0, 0, 1, 0, 0, 1};
  if (W4W_DMAW == 8) return {0, M, 0, 0, 0, 1};
  if (W4W_DMAW == 9) return NT == 3 ? W4WPlan{0, 0, 1, 0, 0, 1} : W4WPlan{0, 0, 1, 0, M, 0};      // shipped: see the table above
  if (NT == 1) return {0, 2, 1, 0, 2, 1};
  return {2, 2, 1, 2, 2, 1};
}
template <int NT> constexpr int w4w_nwu() { return w4w_plan<NT>().nmu + 2 * w4w_plan<NT>().kpu; }
template <int NT> constexpr int w4w_nwr() { return w4w_plan<NT>().nmr + 2 * w4w_plan<NT>().kpr; }
// wave id -> role: producer index (0 / 1) or -1; MFMA index m
template <int NT> __device__ __forceinline__ int w4w_producer_of(int wave) {
  if constexpr (W4W_LAYOUT == 1 && NT == 3) return (wave & 3) == 3 ? wave >> 2 : -1;
  return wave >= 2 * NT ? wave - 2 * NT : -1;
}
template <int NT> __device__ __forceinline__ int w4w_mfma_index(int wave) {
  if constexpr (W4W_LAYOUT == 1 && NT == 3) return wave - (wave >> 2);
  return wave;
}

// np (wave-uniform, 0 .. MAXN) gather pieces
template <int MAXN>
__device__ __forceinline__ void w4w_gather_n(int np, const void* sbase, const unsigned (&voff)[8], const unsigned long long (&mask)[8], unsigned dst0,
                                             unsigned step) {
  if constexpr (MAXN >= 1) {
    if (np == MAXN) w4::dma_gather<MAXN>(sbase, voff, mask, dst0, step);
    else w4w_gather_n<MAXN - 1>(np, sbase, voff, mask, dst0, step);
  }
}

template <int NT, int FLAT>
struct W4WDuty {
  static constexpr int NWU = w4w_nwu<NT>(), NWR = w4w_nwr<NT>();
  static constexpr int MAXP = (16 + NWR - 1) / NWR;                    // raw pieces per raw issuer (rawF4 <= 1024 slots)
  static constexpr int NUP = (9 * NT + NWU - 1) / NWU;                 // U pieces per U issuer and slice
  int goff[MAXP];
  bool live[MAXP];
  unsigned long long lmask[MAXP];                                       // lanes of a piece that carry an in-image position
  int dwu, dwr;                                                         // this wave's index among the U / raw issuers, -1 = none (wave-uniform)
  // the patch of item `it`: global float offsets of this issuer's raw pieces (lane = slot inside the piece), -1 = padding
  __device__ __forceinline__ void setup(const W4WParams& pp, int it, int lane) {
    if (dwr < 0) return;
    int ng;
    const int id = w4w_item_id(pp, it, &ng);
    raw_piece_offsets<MAXP, FLAT>(pp.g, id, dwr, NWR, lane, goff);
#pragma unroll
    for (int k = 0; k < MAXP; ++k) { lmask[k] = __ballot(goff[k] >= 0); live[k] = lmask[k] != 0ull; }
  }
  // 4-channel slice c4 of the patch -> raw ring slot; zero = also clear the padding lanes of the slot (first use by this item)
  __device__ __forceinline__ int issue_raw(const W4PParams& p, float4* smem, int lane, int c4, int slot, bool zero) const {
    if (dwr < 0) return 0;
    const int dw = dwr;
    constexpr int NW = NWR;
    int cnt = 0;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float4*)smem;
    const float* sbase = p.in + (size_t)(c4 >> 2) * p.in_ss + (c4 & 3) * 4;
    const unsigned sb = lds_base + (unsigned)(slot * p.rawF4) * 16u;
    const int npieces_raw = p.rawF4 >> 6;
    if (zero) {
#pragma unroll
      for (int k = 0; k < MAXP; ++k) {
        const int piece = dw + NW * k;
        if (piece < npieces_raw && goff[k] < 0) {
          float4 zz = make_float4(0.f, 0.f, 0.f, 0.f);
          asm volatile("" : "+v"(zz.x), "+v"(zz.y), "+v"(zz.z), "+v"(zz.w));      // (materialised here: hipcc kept one zero vector live across the K loop and spilled it)
          smem[slot * p.rawF4 + piece * 64 + lane] = zz;
        }
      }
    }
    if constexpr (W4W_GATHER != 0 && w4w_plan<NT>().nmr == 0) {
      // (only where the producers alone request the patch: on the MFMA waves the always-issued empty pieces and exec switches cost
      // more than the per-piece branches - 48.8 vs 47.4 us)
      // all of this issuer's pieces in one (two) asm block(s); the number of pieces dw, dw + NW, ... < npieces_raw is wave-uniform
      static_assert(MAXP <= 8, "at most 8 raw pieces per issuer");
      unsigned voff[8];
      unsigned long long mask[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        voff[k] = k < MAXP ? (unsigned)goff[k] * 4u : 0u;
        mask[k] = k < MAXP ? lmask[k] : 0ull;
      }
      const int np = dw < npieces_raw ? (npieces_raw - 1 - dw) / NW + 1 : 0;            // pieces of this issuer
      const unsigned dst0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(sb + (unsigned)dw * 1024u));
      w4w_gather_n<MAXP>(np, sbase, voff, mask, dst0, NW * 1024u);
      return np;
    }
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
      const int piece = dw + NW * k;
      if (piece < npieces_raw && live[k]) {
        if (goff[k] >= 0)
          w4::dma16_sv(sbase, (unsigned)goff[k] * 4u, (unsigned)__builtin_amdgcn_readfirstlane((int)(sb + (unsigned)piece * 1024u)));
        ++cnt;
      }
    }
    return cnt;
  }
  // U of slice c4, n-tiles nt0 .. nt0 + NT - 1 -> U ring slot.  The 9 NT one-KiB pieces of a (slice, n-group) are CONTIGUOUS in the
  // packed fragments ([c4][n-tile][9 quads][64] float4), so piece i is base + i KiB: one 64-bit scalar add per piece.  An n-group that
  // reaches beyond the tensor (Cout = 112: 7 n-tiles at NT = 3) reads on into the next slice's fragments / the slack behind the
  // last one (conv_wino4w_packed_floats) - those waves' results are never stored.
  __device__ __forceinline__ int issue_u(const W4PParams& p, float4* smem, int lane, int nt0, int c4, int slot) const {
    if (dwu < 0) return 0;
    const int dw = dwu;
    constexpr int NW = NWU;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float4*)smem;
    const unsigned dst0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + (unsigned)(p.uoff + slot * NT * W4W_UBLK) * 16u + (unsigned)dw * 1024u));
    const float4* src = p.ufrag + (((size_t)c4 * p.nT16 + nt0) * 9 + dw) * 64;
    // pieces dw, dw + NW, ...: the first NUP - 1 exist for every issuer, the last one only while dw + NW (NUP - 1) < 9 NT
    unsigned voff[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) voff[k] = (unsigned)lane * 16u + (unsigned)(k * NW) * 1024u;
    constexpr int NA = NUP <= 7 ? NUP : 7;                      // first call: up to 7 pieces
    const bool full = dw + NW * (NUP - 1) < 9 * NT;             // (wave-uniform)
    if constexpr (NUP <= 7) {
      if (full) w4::dma_stream<NA>(src, voff, dst0, NW * 1024u);
      else if constexpr (NA > 1) w4::dma_stream<(NA > 1 ? NA - 1 : 1)>(src, voff, dst0, NW * 1024u);
      return full ? NUP : NUP - 1;
    } else {
      static_assert(NUP <= 14, "two calls of up to 7 pieces");
      w4::dma_stream<7>(src, voff, dst0, NW * 1024u);
      constexpr int NB = NUP - 7;
      const float4* src2 = src + (size_t)(7 * NW) * 64;
      const unsigned dst2 = dst0 + 7u * NW * 1024u;
      if (full) w4::dma_stream<NB>(src2, voff, dst2, NW * 1024u);
      else if constexpr (NB > 1) w4::dma_stream<(NB > 1 ? NB - 1 : 1)>(src2, voff, dst2, NW * 1024u);
      return full ? NUP : NUP - 1;
    }
  }
  // once per block, before P0: raw(0..2), U(0), U(1) of the first item
  __device__ __forceinline__ void prologue(const W4WParams& pp, float4* smem, int lane, int it0) {
    const W4PParams& p = pp.g;
    setup(pp, it0, lane);
#pragma unroll
    for (int c = 0; c < 3; ++c)
      if (c < p.nC4) issue_raw(p, smem, lane, c, c, true);
    int ng0;
    (void)w4w_item_id(pp, it0, &ng0);
    issue_u(p, smem, lane, ng0 * NT, 0, 0);
    if (p.nC4 > 1) issue_u(p, smem, lane, ng0 * NT, 1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  // global slice t = (item it, slice s): U(t + 2) -> slot r2 (of U(t - 1)); raw(t + 4) -> slot r1 (of raw(t + 1), whose window was read
  // during slice t - 1) - of this item or of the next one, whose patch takes over at s = S - 4.  Returns the number of requests.
  __device__ __forceinline__ int slice_requests(const W4WParams& pp, float4* smem, int lane, int it_next, bool hasB, int s, int r1,
                                                int r2, int nt0A, int nt0B) {
    const W4PParams& p = pp.g;
    const int S = p.nC4;
    int nvm = 0;
    if (s + 2 < S) nvm += issue_u(p, smem, lane, nt0A, s + 2, r2);
    else if (hasB) nvm += issue_u(p, smem, lane, nt0B, s + 2 - S, r2);
    if (s + 4 == S && hasB) setup(pp, it_next, lane);
    if (s + 4 < S) nvm += issue_raw(p, smem, lane, s + 4, r1, false);
    else if (hasB) nvm += issue_raw(p, smem, lane, s + 4 - S, r1, s + 4 - S < 3);
    return nvm;
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// MFMA waves
// ---------------------------------------------------------------------------------------------------------------------
template <int NT, int FLAT>
__device__ __forceinline__ void w4w_mfma_wave(const W4WParams& pp, float4* smem, int wave, int lane) {      // wave = MFMA index 0 .. 2 NT - 1
  const W4PParams& p = pp.g;
  const int grp = wave >= NT ? 1 : 0, nn = wave - grp * NT;           // (wave-uniform)
  const int idx = lane & 15, g = lane >> 4;
  const int vlane = w4p_sigma(idx, g);
  constexpr W4WPlan plan = w4w_plan<NT>();
  W4WDuty<NT, FLAT> duty;
  duty.dwu = wave >= plan.mu0 && wave < plan.mu0 + plan.nmu ? wave - plan.mu0 : -1;
  duty.dwr = wave >= plan.mr0 && wave < plan.mr0 + plan.nmr ? wave - plan.mr0 : -1;
  const bool is_dma = duty.dwu >= 0 || duty.dwr >= 0;                  // (wave-uniform)
  const int uF4 = NT * W4W_UBLK;
  const int S = p.nC4;
  const Walk wk = item_walk(p);
  if (wk.first >= wk.end) return;                                     // (whole block: every role takes the same exit)

  if (is_dma) duty.prologue(pp, smem, lane, wk.first);
  __syncthreads();                                        // P0: the first fetches have landed
  __syncthreads();                                        // P1: V(0) is written, the windows of slices 0 and 1 are in the producers' registers
  if (is_dma && S > 3) duty.issue_raw(p, smem, lane, 3, 0, false);     // (the window of slice 0 has been read)

  int ring = 0, vb = 0;                                   // global slice t: t % 3, t & 1
  for (int it = wk.first; it < wk.end; it += wk.step) {
    const bool hasB = it + wk.step < wk.end;
    int ngA, ngB = 0;
    const int idA = w4w_item_id(pp, it, &ngA);
    if (hasB) (void)w4w_item_id(pp, it + wk.step, &ngB);
    const int nt0A = ngA * NT, nt0B = ngB * NT;
    const int nt = nt0A + nn;                             // this wave's n-tile (may lie beyond the tensor in the last n-group)

    f32x4 acc[9][4];
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[q][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 hu[W4W_HOLD > 0 ? W4W_HOLD : 1], hv[W4W_HOLD > 0 ? W4W_HOLD : 1];       // operands of the deferred quads

#if W4W_TRACE
    const bool trace = blockIdx.x == 0 && it == wk.first;
    unsigned long long tr[4] = {0, 0, 0, 0};
#endif
    for (int s = 0; s < S; ++s) {
      W4W_T(c0);
      const int r1 = ring == 2 ? 0 : ring + 1, r2 = r1 == 2 ? 0 : r1 + 1;
      int nvm = 0;
      const float4* U = smem + p.uoff + ring * uF4 + nn * W4W_UBLK + lane;
      const float4* V = smem + p.voff + vb * (2 * W4W_UBLK) + grp * W4W_UBLK + vlane;
      // Operands two quads ahead of the MFMAs that use them.  The MFMAs of the last W4W_HOLD quads of a slice are DEFERRED ACROSS THE
      // SLICE BARRIER: their operands are read before it (the barrier protects the ring slots against the next requests, and it is
      // the READS that must be over), the MFMAs themselves run behind it, while the first operands of the next slice are on their way -
      // right after a barrier every wave of the block asks the LDS for operands at once, and the MFMA pipes used to idle through that
      // round trip (~300 clk of a ~2900-clk slice on every SIMD).  The last slice of an item keeps nothing back (its accumulators go to
      // the epilogue).
      constexpr int PF = W4W_PF, NH = W4W_HOLD;
      float4 ub[PF + 1], vq[PF + 1];
#pragma unroll
      for (int q = 0; q < PF; ++q) { ub[q] = (W4W_EXP & 4) ? make_float4(1.f, 2.f, (float)s, 3.f) : U[q * 64]; vq[q] = (W4W_EXP & 4) ? make_float4(1.f, 2.f, 3.f, (float)q) : V[q * 64]; }
      // The LDS-DMA of the coming slices: requested while the first operands are on their way.
      // U(t + 2) -> slot of U(t - 1); raw(t + 4) -> slot of raw(t + 1), whose window was read during slice t - 1.
      if (is_dma) {
        __builtin_amdgcn_sched_barrier(0);
        if (!(W4W_EXP & 1)) nvm = duty.slice_requests(pp, smem, lane, it + wk.step, hasB, s, r1, r2, nt0A, nt0B);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (NH > 0 && s > 0) {
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            acc[9 - NH + h][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4c(hu[h], i), f4c(hv[h], i), acc[9 - NH + h][i], 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        if (q + PF < 9) {
          ub[(q + PF) % (PF + 1)] = (W4W_EXP & 4) ? make_float4(1.f, (float)s, 2.f, 3.f) : U[(q + PF) * 64];
          vq[(q + PF) % (PF + 1)] = (W4W_EXP & 4) ? make_float4(1.f, 2.f, 3.f, (float)q) : V[(q + PF) * 64];
        }
        const float4 u = ub[q % (PF + 1)], v = vq[q % (PF + 1)];
        if (q < 9 - NH || s + 1 == S) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (W4W_EXP & 8) acc[q][i][0] += f4c(u, i) * f4c(v, i);
            else acc[q][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4c(u, i), f4c(v, i), acc[q][i], 0, 0, 0);
          }
        } else {
          hu[q - (9 - NH)] = u; hv[q - (9 - NH)] = v;
        }
      }
      W4W_T(c1);
      if (is_dma) w4w_wait_vm(nvm);                       // what this wave requested BEFORE this slice has landed: U(t + 1), raw(t + 3)
      ring = r1;
      vb ^= 1;
      W4W_T(c2);
      __syncthreads();                                    // everybody is done with slice t; V(t + 1), U(t + 1), raw(t + 2) are in place
      __builtin_amdgcn_sched_barrier(0);
      W4W_T(c3);
      W4W_ACC(0, c0, c1); W4W_ACC(1, c1, c2); W4W_ACC(2, c2, c3);
    }
    W4W_T(e0);

    // ---- epilogue, registers only: Y = A^T M A, bias, residual, ReLU, 16-byte stores of the lane's 4x4 pixels x 4 channels ----
    {
      const Tile tl = tile_of<FLAT>(p, idA, grp, idx);
      const int ntc = min(nt, p.nT16 - 1);
      const int g4 = g * 4;
      const float lo = p.act == 1 ? 0.f : -INFINITY;
      const bool has_res = p.res != nullptr;
      const int oyb = tl.oy0, oxb = 4 * tl.tx;
      bool okx[4], oky[4];
      unsigned ooff[4], roff[4], xoffb[4];              // byte offsets of the rows and columns (clamped: dead pixels compute harmlessly, masked at the store)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        okx[j] = tl.valid && oxb + j < p.W;
        xoffb[j] = (unsigned)(min(oxb + j, p.W - 1) * 16 + g4) * 4u;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        oky[i] = oyb + i < p.H;
        const int orow = tl.b * p.H + min(oyb + i, p.H - 1);
        ooff[i] = (unsigned)orow * (unsigned)p.out_rs * 4u;
        roff[i] = (unsigned)orow * (unsigned)p.res_rs * 4u;
      }
      const float4 sh = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p.bias + ntc * 16) + (unsigned)(g4 * 4));
      const char* rb = reinterpret_cast<const char*>(p.res + (size_t)ntc * p.out_ss);
      char* ob = reinterpret_cast<char*>(p.out + (size_t)ntc * p.out_ss);
      const bool ntok = nt < p.nT16;
      // Z[xi][.] = A^T applied along nu to row xi of M, row by row: a row of M (24 registers) dies as soon as its row of Z (16)
      // exists - the allocator sees 144 -> 96 live values instead of 144 + 96.  A^T in its even / odd form (w4w_at: 10 packed
      // operations per 6 -> 4 transform instead of the 14 of the term-by-term sums; the epilogue is VALU-bound, two MFMA waves per SIMD)
      f32x4 z[6][4];
#pragma unroll
      for (int xi = 0; xi < 6; ++xi) {
        w4w_at(acc[w4w_slot(xi, 0) >> 2][w4w_slot(xi, 0) & 3], acc[w4w_slot(xi, 1) >> 2][w4w_slot(xi, 1) & 3],
               acc[w4w_slot(xi, 2) >> 2][w4w_slot(xi, 2) & 3], acc[w4w_slot(xi, 3) >> 2][w4w_slot(xi, 3) & 3],
               acc[w4w_slot(xi, 4) >> 2][w4w_slot(xi, 4) & 3], acc[w4w_slot(xi, 5) >> 2][w4w_slot(xi, 5) & 3], z[xi]);
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(z[xi][j][0]), "+v"(z[xi][j][1]), "+v"(z[xi][j][2]), "+v"(z[xi][j][3]));    // (row xi is finished here)
      }
      // one output column j at a time: Y[i][j] = sum_xi A^T[i][xi] Z[xi][j]; the residual of column j + 1 travels meanwhile
      float4 rr[2][4];
      if (has_res) {
#pragma unroll
        for (int i = 0; i < 4; ++i) rr[0][i] = *reinterpret_cast<const float4*>(rb + (roff[i] + xoffb[0]));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (has_res && j + 1 < 4) {
#pragma unroll
          for (int i = 0; i < 4; ++i) rr[(j + 1) & 1][i] = *reinterpret_cast<const float4*>(rb + (roff[i] + xoffb[j + 1]));
        }
        f32x4 yc[4];
        w4w_at(z[0][j], z[1][j], z[2][j], z[3][j], z[4][j], z[5][j], yc);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          f32x4 v = yc[i] + (f32x4){sh.x, sh.y, sh.z, sh.w};
          if (has_res) {
            const float4 r = rr[j & 1][i];
            if (p.res_after_act) {
              v[0] = fmaxf(v[0], lo) + r.x; v[1] = fmaxf(v[1], lo) + r.y; v[2] = fmaxf(v[2], lo) + r.z; v[3] = fmaxf(v[3], lo) + r.w;
            } else {
              v[0] = fmaxf(v[0] + r.x, lo); v[1] = fmaxf(v[1] + r.y, lo); v[2] = fmaxf(v[2] + r.z, lo); v[3] = fmaxf(v[3] + r.w, lo);
            }
          } else {
            v[0] = fmaxf(v[0], lo); v[1] = fmaxf(v[1], lo); v[2] = fmaxf(v[2], lo); v[3] = fmaxf(v[3], lo);
          }
          if (ntok && okx[j] && oky[i]) *reinterpret_cast<float4*>(ob + (ooff[i] + xoffb[j])) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    }
#if W4W_TRACE
    W4W_T(e1);
    W4W_ACC(3, e0, e1);
    if (trace && lane == 0) { for (int k = 0; k < 4; ++k) g_w4w_trace[4 * wave + k] = tr[k]; g_w4w_trace[63] = (unsigned long long)S; }
#endif
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// producer waves: input transform V = B^T d B, all 36 positions of tile group pw
// ---------------------------------------------------------------------------------------------------------------------
template <int NT, int FLAT>
__device__ __forceinline__ void w4w_producer(const W4WParams& pp, float4* smem, int pw, int lane) {
  const W4PParams& p = pp.g;
  const int grp = pw;
  const int idx = lane >> 2, g = lane & 3;   // transform lane order: 8 tiles x 4 channels per 32-lane half (see w4p_sigma)
  const int vlane = w4p_sigma(idx, g);
  const int rawF4 = p.rawF4;
  const int S = p.nC4;
  const Walk wk = item_walk(p);
  if (wk.first >= wk.end) return;

  // float offsets of the lane's window (column pairs: one ds_read2_b32 each, see conv_wino4p.hip) inside a raw slot
  struct Win { int woff[6][3]; };
  auto win_of = [&](int it, Win& w) __attribute__((always_inline)) {
    int ng;
    const int id = w4w_item_id(pp, it, &ng);
    const Tile tl = tile_of<FLAT>(p, id, grp, idx);
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int pos = tl.base + k * p.PW + 2 * c;
        w.woff[k][c] = (pos + (pos >> (FLAT ? 4 : 3))) * 4 + g;
      }
  };
  // W4W_WIN2 = 1: TWO window register sets used alternately (by the parity of the global slice: no copies) - the window of slice t + 2 is
  // requested at the top of slice t and its latency hides behind the transform of window t + 1.
  // (Tried in the first layout: TWO window register sets, the window of slice t + 2 requested at the top of slice t ahead of the transform of window
  // t + 1 - 3-7 % slower per launch: the 15 reads then queue in front of the MFMA waves' first operand reads of the slice.)
  f32x2 dA[6][3];
#if W4W_WIN2
  f32x2 dB[6][3];
#endif
  auto load_window = [&](f32x2 (&d)[6][3], const Win& w, int rslot) __attribute__((always_inline)) {
    const float* rawf = reinterpret_cast<const float*>(smem + rslot * rawF4);
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float* q = rawf + w.woff[k][c];
        d[k][c] = (f32x2){q[0], q[4]};
      }
  };
  // V = B^T d B in packed fp32: stage 1 down the window columns (on the column pairs as read), stage 2 along the rows, which
  // produces the pairs (nu0, nu5), (nu1, nu3), (nu2, nu4) per row; 9 ds_write_b128 in exactly that order (w4w_row / w4w_nu)
  auto transform = [&](const f32x2 (&d)[6][3], int vbuf) __attribute__((always_inline)) {
    f32x2 t[6][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const f32x2 d0 = d[0][c], d1 = d[1][c], d2 = d[2][c], d3 = d[3][c], d4 = d[4][c], d5 = d[5][c];
      t[0][c] = pk_fma(d0, 4.f, pk_fma(d2, -5.f, d4));                       // 4 d0 - 5 d2 + d4
      const f32x2 a = pk_fma(d2, -4.f, d4), cc = pk_fma(d1, 4.f, -d3);       // rows 1, 2 = (d4 - 4 d2) -+ (4 d1 - d3)
      t[1][c] = a - cc;
      t[2][c] = a + cc;
      const f32x2 b = d4 - d2, e = d1 - d3;                                  // rows 3, 4 = (d4 - d2) -+ 2 (d1 - d3)
      t[3][c] = pk_fma(e, -2.f, b);
      t[4][c] = pk_fma(e, 2.f, b);
      t[5][c] = pk_fma(d1, 4.f, pk_fma(d3, -5.f, d5));                       // 4 d1 - 5 d3 + d5
    }
    float4* Vg = smem + p.voff + vbuf * (2 * W4W_UBLK) + grp * W4W_UBLK + vlane;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      f32x2 Ap[2], Bp[2], Cp[2];                              // rows 2m, 2m + 1: (v0, v5), (v1, v3), (v2, v4)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const f32x2 T0 = t[2 * m + r][0], T1 = t[2 * m + r][1], T2 = t[2 * m + r][2];
        Ap[r] = pk_fma(T0, 4.f, pk_fma(T1, -5.f, T2));                            // 4 t0 - 5 t2 + t4 | 4 t1 - 5 t3 + t5
        const f32x2 ab = pk_fma2(T1.xx, (f32x2){-4.f, -1.f}, T2.xx);             // t4 - 4 t2 | t4 - t2
        const f32x2 cf = pk_fma2(T0.yy, (f32x2){4.f, 2.f}, T1.yy * (f32x2){-1.f, -2.f});   // 4 t1 - t3 | 2 t1 - 2 t3
        Bp[r] = ab - cf;
        Cp[r] = ab + cf;
      }
      Vg[(3 * m) * 64] = make_float4(Ap[0].x, Ap[0].y, Bp[0].x, Bp[0].y);
      Vg[(3 * m + 1) * 64] = make_float4(Cp[0].x, Cp[0].y, Ap[1].x, Ap[1].y);
      Vg[(3 * m + 2) * 64] = make_float4(Bp[1].x, Bp[1].y, Cp[1].x, Cp[1].y);
    }
  };

  // this wave's share of the LDS-DMA duty (W4W_DMAW): issuer index behind the MFMA waves'
  constexpr W4WPlan plan = w4w_plan<NT>();
  constexpr bool kIssue = plan.kpu > 0 || plan.kpr > 0;
  W4WDuty<NT, FLAT> duty;
  duty.dwu = plan.kpu > 0 ? plan.nmu + pw : -1;
  duty.dwr = plan.kpr > 0 ? plan.nmr + pw : -1;
  Win A, B;
  win_of(wk.first, A);
  if constexpr (kIssue) duty.prologue(pp, smem, lane, wk.first);
  __syncthreads();                                        // P0: raw(0..2), U(0..1) of the first item have landed
  load_window(dA, A, 0);
  transform(dA, 0);
  if (S > 1) load_window(dA, A, 1);                       // dA = window of slice t + 1 at the top of slice t
  __syncthreads();                                        // P1
  if constexpr (kIssue) {
    if (S > 3) duty.issue_raw(p, smem, lane, 3, 0, false);
  }
  // fp32 MFMAs run on the SIMD's vector ALUs: without a higher issue priority the MFMA wave of this SIMD starves this wave's
  // VALU / LDS instructions until it reaches the slice barrier
  __builtin_amdgcn_s_setprio(3);
  int ring = 0, vb = 0;
  for (int it = wk.first; it < wk.end; it += wk.step) {
    const bool hasB = it + wk.step < wk.end;
    if (hasB) win_of(it + wk.step, B);
    int ngA, ngB = 0;
    (void)w4w_item_id(pp, it, &ngA);
    if (hasB) (void)w4w_item_id(pp, it + wk.step, &ngB);
#if W4W_TRACE
    const bool trace = blockIdx.x == 0 && it == wk.first;
    unsigned long long tr[4] = {0, 0, 0, 0};
#endif
    for (int s = 0; s < S; ++s) {
      W4W_T(q0);
      const int r1 = ring == 2 ? 0 : ring + 1, r2 = r1 == 2 ? 0 : r1 + 1;
      int nvm = 0;
      if constexpr (kIssue && !W4W_DMALAST) { if (!(W4W_EXP & 1)) nvm = duty.slice_requests(pp, smem, lane, it + wk.step, hasB, s, r1, r2, ngA * NT, ngB * NT); }
      // V(t + 1) from the window fetched during the previous slice, then the window of slice t + 2 (raw(t + 2) landed before the
      // barrier that ended slice t - 1) - of this item or of the next one
      if (!(W4W_EXP & 2)) {
#if W4W_WIN2
      auto step = [&](f32x2 (&cur)[6][3], f32x2 (&nxt)[6][3]) __attribute__((always_inline)) {
        if (s + 2 < S) load_window(nxt, A, r2);
        else if (hasB) load_window(nxt, B, r2);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < S || hasB) transform(cur, vb ^ 1);
      };
      if (vb) step(dB, dA); else step(dA, dB);
      W4W_T(q1);
#else
      if (s + 1 < S || hasB) transform(dA, vb ^ 1);
      W4W_T(q1);
      if (s + 2 < S) load_window(dA, A, r2);
      else if (hasB) load_window(dA, B, r2);
#endif
      }
      // W4W_DMALAST: the requests of the coming slices go out behind the window reads, whose latency (~650 clk under the MFMA waves'
      // operand traffic) they cover - U(t + 2) / raw(t + 4) have two slices to land either way
      if constexpr (kIssue && W4W_DMALAST) {
        __builtin_amdgcn_sched_barrier(0);
        if (!(W4W_EXP & 1)) nvm = duty.slice_requests(pp, smem, lane, it + wk.step, hasB, s, r1, r2, ngA * NT, ngB * NT);
      }
      if (kIssue) w4w_wait_vm(nvm);                       // what this wave requested BEFORE this slice has landed
      ring = r1;
      vb ^= 1;
      W4W_T(q2);
      __syncthreads();
      W4W_T(q3);
      W4W_ACC(0, q0, q1); W4W_ACC(1, q1, q2); W4W_ACC(2, q2, q3);
    }
#if W4W_TRACE
    if (trace && lane == 0) for (int k = 0; k < 3; ++k) g_w4w_trace[32 + 4 * pw + k] = tr[k];
#endif
    if (hasB) A = B;
  }
}

template <int NT, int FLAT>      // FLAT: 0 rectangular items, 1 flat items
__global__ void __launch_bounds__(128 * NT + 128)
conv_wino4w_kernel(const W4WParams p) {
  extern __shared__ float4 smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pw = w4w_producer_of<NT>(wave);
  if (pw < 0) w4w_mfma_wave<NT, FLAT>(p, smem, w4w_mfma_index<NT>(wave), lane);
  else w4w_producer<NT, FLAT>(p, smem, pw, lane);
}

struct W4WLayout { int uoff, voff, totalF4; };
bool w4w_geo(const ConvDesc& d, const ConvCfg& cfg, w4::Geo* g, W4WLayout* L, FlatGeo* fg = nullptr) {
  if (d.ks != 3 || d.stride != 1 || cfg.NT < 1 || cfg.NT > 3 || cfg.WM != 2 || cfg.WN != 1 || d.Cin % 16 || d.Cout % 16) return false;
  FlatGeo ftmp;
  if (cfg.NI == 0) { if (cfg.R != 4 || !flat_geo(d, cfg, g, fg ? fg : &ftmp)) return false; }      // (no mosaic items here)
  else if (!w4::geo(d, cfg, 32, g)) return false;
  if (g->rawF4 > 1024) return false;
  // 32-bit byte offsets in the epilogue
  if ((long)d.B * d.H * d.W * std::max(std::max(d.in_cs, d.out_cs), d.res_cs) >= (1L << 30)) return false;
  const int uF4 = cfg.NT * W4W_UBLK;
  L->uoff = 3 * g->rawF4;
  L->voff = L->uoff + 3 * uF4;
  L->totalF4 = L->voff + 4 * W4W_UBLK;
  return (size_t)L->totalF4 * sizeof(float4) <= 160 * 1024;
}

}  // namespace

// packed fragments for ALG 13: [Cin/4][Cout16/16][9 quads][64 lanes] float4; lane = g*16 + co_l holds
// U[6 w4w_row(q, i) + w4w_nu(q, i)][co][4 c4 + g] * scale[co], i = 0..3
size_t conv_wino4w_packed_floats(int Cin, int Cout16) { return (size_t)36 * Cin * Cout16 + (size_t)2 * 9 * 256; }     // + 2 n-tile blocks of slack (issue_u)
void conv_wino4w_pack_weights(const float* w_oihw, const float* scale, int Cout, int Cin, int Cout16, float* dst) {
  std::vector<double> u;
  w4::u_transform(w_oihw, Cout, Cin, &u);
  const int nC4 = Cin / 4, nT16 = Cout16 / 16;
  auto val = [&](int pos, int co, int ci) -> float {
    return co < Cout ? (float)(u[((size_t)pos * Cout + co) * Cin + ci] * (scale ? (double)scale[co] : 1.0)) : 0.f;
  };
  std::fill(dst + (size_t)36 * Cin * Cout16, dst + conv_wino4w_packed_floats(Cin, Cout16), 0.f);
  for (int c4 = 0; c4 < nC4; ++c4)
    for (int nt = 0; nt < nT16; ++nt) {
      float* blk = dst + ((size_t)c4 * nT16 + nt) * 9 * 256;
      for (int lane = 0; lane < 64; ++lane) {
        const int g = lane >> 4, co = nt * 16 + (lane & 15), ci = 4 * c4 + g;
        for (int q = 0; q < 9; ++q)
          for (int i = 0; i < 4; ++i) blk[(q * 64 + lane) * 4 + i] = val(6 * w4w_row(q, i) + w4w_nu(q, i), co, ci);
      }
    }
}

// cfg: {MT = CU-share divisor, NT (1..3) n-tiles per item = MFMA waves per tile group, WM = 2 tile groups, WN = 1, R, NI as ALG 8, ALG = 13}
size_t conv_wino4w_lds_bytes(const ConvDesc& d, const ConvCfg& cfg) {
  w4::Geo g;
  W4WLayout L;
  if (!w4w_geo(d, cfg, &g, &L)) return 0;
  return (size_t)L.totalF4 * sizeof(float4);
}

int conv_wino4w_launch(const ConvDesc& d, const ConvCfg& cfg, hipStream_t stream) {
  w4::Geo g;
  W4WLayout L;
  FlatGeo fg{};
  const bool flat = cfg.NI == 0;
  if (!w4w_geo(d, cfg, &g, &L, &fg) || !d.wfrag_wino4w) {
    poco_set_error("conv(winograd 4x4, whole-position waves): needs ks = 3, stride 1, NT 1..3, WM = 2, WN = 1, R % 4 == 0, "
                   "NI*(R/4)*ceil(W/4) <= 32 tiles (or R = 4 MS, NI = 0: flat items), a patch of <= 1024 slots that fits the LDS next "
                   "to the U ring, tensors below 2^30 elements and the ALG 13 weight fragments");
    return POCO_ERR_ARG;
  }
  if (d.act == 3 || d.act == 2) { poco_set_error("conv(winograd 4x4): activation must be none or ReLU"); return POCO_ERR_ARG; }
  W4WParams pp{};
  W4PParams& p = pp.g;
  p.in = d.in + l16_chan_off(d.in_co, d.W);
  p.res = d.res ? d.res + l16_chan_off(d.res_co, d.W) : nullptr;
  p.out = d.out + l16_chan_off(d.out_co, d.W);
  p.ufrag = reinterpret_cast<const float4*>(d.wfrag_wino4w); p.bias = d.bias;
  p.B = d.B; p.H = d.H; p.W = d.W; p.nC4 = d.Cin / 4; p.nT16 = d.Cout / 16;
  p.in_rs = d.in_cs * d.W; p.in_ss = d.W * 16; p.res_rs = d.res_cs * d.W; p.out_rs = d.out_cs * d.W; p.out_ss = d.W * 16;
  p.R = g.R; p.NI = g.NI; p.S = g.S; p.nbands = g.nbands; p.TX = g.TX; p.PR = g.PR; p.PW = g.PW; p.npos = g.npos; p.rawF4 = g.rawF4;
  p.tiles_per_slab = g.tps;
  p.act = d.act; p.res_after_act = d.res_after_act;
  p.uoff = L.uoff; p.voff = L.voff; p.xoff = 0;
  p.dPW = make_fastdiv(g.PW); p.dSlab = make_fastdiv(g.PR * g.PW); p.dBands = make_fastdiv(g.nbands);
  p.dTX = make_fastdiv(g.TX); p.dTslab = make_fastdiv(g.tps);
  p.nblocks_m = flat ? g.S : (g.S + g.NI - 1) / g.NI; p.nb_n = (p.nT16 + cfg.NT - 1) / cfg.NT;
  pp.dNbn = make_fastdiv(p.nb_n);
  if (flat) {
    p.TY = fg.TY; p.ntiles = fg.ntiles; p.fragW = fg.fragW;
    p.dTY = make_fastdiv(fg.TY); p.dFragW = make_fastdiv(fg.fragW);
    p.MS = fg.MS; p.Hp1 = d.H + 1; p.Wp1 = d.W + 1;
    p.dHp1 = make_fastdiv(d.H + 1); p.dWp1 = make_fastdiv(d.W + 1);
  }
  // balanced persistent grid: every block walks the same number of items (one block per CU: the LDS); cfg.MT = CU-share divisor
  const int mt = std::max(1, cfg.MT);
  const long cus = std::max(8, poco_num_cus() / mt);
  const long items = (long)p.nblocks_m * p.nb_n;
  const long rounds = (items + cus - 1) / cus;
  long g4 = (items + rounds - 1) / rounds;
  if (g4 > 8) g4 = std::min(cus, (g4 + 7) / 8 * 8);            // multiple of 8 for the XCD-aware walk
  const size_t lds = (size_t)L.totalF4 * sizeof(float4);
  const int mode = flat ? 1 : 0;
  void (*fn)(const W4WParams) =
      mode == 1 ? (cfg.NT == 3 ? conv_wino4w_kernel<3, 1> : cfg.NT == 2 ? conv_wino4w_kernel<2, 1> : conv_wino4w_kernel<1, 1>)
                  : (cfg.NT == 3 ? conv_wino4w_kernel<3, 0> : cfg.NT == 2 ? conv_wino4w_kernel<2, 0> : conv_wino4w_kernel<1, 0>);
  if (lds > 64 * 1024) {
    static thread_local bool configured[12] = {};
    if (!configured[cfg.NT + 4 * mode]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) { poco_set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return POCO_ERR_HIP; }
      configured[cfg.NT + 4 * mode] = true;
    }
  }
  hipLaunchKernelGGL(fn, dim3((unsigned)g4, 1), dim3(128 * cfg.NT + 128), lds, stream, pp);
  POCO_HIP_CHECK(hipGetLastError());
  return POCO_OK;
}

#if W4W_TRACE
extern "C" int poco_w4w_trace(unsigned long long* host_out, int n) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_w4w_trace), sizeof(unsigned long long) * (size_t)std::min(n, 64)) == hipSuccess ? 0 : 1;
}
#endif
