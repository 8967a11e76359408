// ALG 13: Winograd F(4x4,3x3) on the fp32 MFMA with WHOLE-POSITION MFMA waves (round 5) whose weights go STRAIGHT INTO REGISTERS
// (round 6) - no exchange, no item start, no U ring (3x3 stride-1 convs on planes >= 14x14; pocolib/models/backbone/hrnet.py:42-58,
// hrnet_cls.py BasicBlock convs, resnet.py:101-121 conv2).
//
// ALG 8 (conv_wino4p.hip) splits the 36 Winograd positions of a 16-tile group over four MFMA waves (9 positions x NT n-tiles
// each: 108 accumulators at the 168-register budget of a 12-wave block), so the output transform A^T M A needs all four
// waves: NT exchange rounds through LDS per item and an exchange area that overlays the V buffers and U ring (VERDICT r4 weak #2:
// 43 % of a launch was fixed cost).  Here a block of 8 waves (two per SIMD, a 256-register budget) is
//   * 2 NT MFMA waves: wave (grp, n) owns ALL 36 positions of one 16-tile group for ONE 16-channel n-tile: 36 accumulators (144
//     registers), per 4-channel slice nine quads of { ds_read_b128 V, 4 MFMAs, global_load_dwordx4 U of the next slice }.  Its
//     output transform A^T M A is register-only, it applies bias / residual / ReLU and stores its 4x4 pixel blocks itself: no
//     exchange, no barrier, no LDS traffic at the end of an item.
//   * 2 producer waves: pw = the input transform V = B^T d B of group pw, all 36 positions per (tile, channel) lane, in packed fp32
//     (15 ds_read2_b32 window reads, ~72 v_pk_* instructions, 9 ds_write_b128 in the pair order of the results, slot swizzle
//     w4p_sigma).  NT = 3: both on SIMD 3 (wave ids 3 and 7), which carries no MFMA wave, and they also request the raw patch;
//   * NT = 2: two more waves (ids 6, 7 = SIMDs 2 / 3) that do nothing but request the raw patch (LDS-DMA), so that every SIMD
//     carries one MFMA wave + one helper.
//   * the slice pipeline is CONTINUOUS ACROSS ITEMS: global slice t = (item k, slice s) uses raw ring slot t % RD and V buffer
//     t & 1; at slice t the requesters ask for raw(t + RD + 1), the producers build V(t + 1) and read the window of slice t + 2,
//     and the MFMA waves request U(t + 1) quad by quad - whichever item those belong to.  One barrier per slice, nothing else.
//     (Padding positions of a patch differ from item to item: whoever requests the first RD slices of an item also zero-fills the
//     padding lanes of the ring slots they go to.)
//   * items are walked n-group-innermost (item = m * nb_n + n-group): the n-groups of one tile strip run on neighbouring
//     blocks of one XCD at the same time and read their patch from one L2.
//
// Round 6, what the s_memtime trace of the round-5 kernel showed (tools/w4w_trace.py; 56x56 48->48, NT = 3, clk per slice): the MFMA
// waves were done issuing after 1190 (older wave of a SIMD) / 1900 clk (younger) and then sat 1600-2300 clk at the slice barrier -
// waiting for the PRODUCERS, whose chain per slice was ~1500 clk of LDS-DMA requests (20 one-KiB pieces each, 13.5 of them U) + 650-900
// of transform + 330 of window reads + 330 of vmcnt wait = ~3100 clk on a SIMD of their own.  Handing the U requests to the MFMA waves
// instead does not help (+1 ... +5 %: every LDS-DMA piece blocks the MFMA pipe of its SIMD for ~47 clk).  But the U stream does not
// need the LDS at all: the packed fragments of a (slice, n-tile) ARE the A operands of one MFMA wave - nine float4 per lane, one
// contiguous KiB per quad - and nobody else uses them except the same-n wave of the other tile group (an L1 / L2 hit).  So every
// MFMA wave loads its nine quads with global_load_dwordx4 straight into the registers its MFMAs read: no U ring (81 KB of LDS), no U
// requests (27 of a slice's 41 LDS-DMA pieces), nine instead of 18 ds_read_b128 per wave and slice.  Same box, us per launch, 64
// crops: 56x56 48->48 48.7 -> 46.9 (flat items 50.0 -> 47.5), 28x28 96->96 44.8 -> 41.9 (46.1 -> 42.6), 14x14 192->192 71.4 -> 64.7,
// 56x56 480->128 840 -> 807.  The freed LDS buys a 4-deep raw ring (a request has three slices to land instead of two: the producers'
// 330-400 clk of vmcnt wait per slice).
//
// Weights: U = G g G^T in float64 on the host (BN scale folded), packed per (4-channel slice, n-tile) as one 9 KiB block of nine
// quads [q][64 lanes] float4, lane = (co & 15) + 16 (ci & 3), quad q / element i = position (w4w_row, w4w_nu) - the order in
// which the producer's packed transform leaves its register pairs.
#include "conv_wino4_common.h"
#include <cstdlib>
#include <type_traits>

namespace {

using w4::at_c;

#include "conv_wino4p_geo.h"

#ifndef W4W_EXP
#define W4W_EXP 0     // timing probes (results are garbage): 1 no patch LDS-DMA in the K loop, 2 producers skip transform + window reads, 4 MFMA waves
#endif                // skip their operand reads, 8 MFMA waves skip the MFMAs, 16 MFMA waves skip their U loads
#ifndef W4W_PF
#define W4W_PF 4      // V operand quads requested ahead of the MFMAs that use them (PF + 1 register sets of 4)
#endif
#ifndef W4W_HOLD
#define W4W_HOLD 1    // NT < 3: the four MFMAs of a slice's LAST quad are issued behind the slice barrier, in front of the next slice's first quad, whose V
#endif                // operands are still on their way from the LDS then (their own operands are in registers; the last slice of an item keeps nothing back).
                      // Same box: 56x56 64->64 85 -> 79 us, 56x56 32->32 28.1 -> 26.7; at NT = 3 it is 2-3 % SLOWER (two MFMA waves per SIMD cover the gap already)
#ifndef W4W_ASMMAX
#define W4W_ASMMAX 1  // the epilogue's ReLU clamp as ONE v_max_f32 per value (inline asm): fmaxf() compiles to two (hipcc first quiets a possible signalling NaN
#endif                // with v_max x, x, x - 64 extra VALU instructions per wave and item in a VALU-bound epilogue); packed residual / bias adds on f32x4
#ifndef W4W_RESEARLY
#define W4W_RESEARLY 1 // residual (conv2 of a BasicBlock): the loads of output columns 0 / 1 go out at the START of the epilogue, under the first transform stage,
#endif                 // those of columns 2 / 3 as soon as columns 0 / 1 are stored; 0 = round 5: column j + 1 requested when column j is computed
#ifndef W4W_ATPK
#define W4W_ATPK 1    // output transform on register pairs (see w4w_at)
#endif
#ifndef W4W_PEEL
#define W4W_PEEL 1    // the first slice of an item accumulates onto a literal zero (the MFMA's C operand) instead of onto 144 registers zeroed with 144 v_mov_b32
#endif
#ifndef W4W_RD
#define W4W_RD 3      // depth of the raw-patch ring in LDS (3: a request has two slices to land, 4: three - measured 1-3 % slower per launch)
#endif
#ifndef W4W_STATICNP
#define W4W_STATICNP 1   // every requester issues exactly MAXP = 8 patch pieces per slice (pieces that do not exist go out with an empty exec
#endif                   // mask: no traffic, but they count in vmcnt), so the landing waits are the constants vmcnt(0 / 8 / 16) instead of a dispatch on a run-time count
#ifndef W4W_YIELD
#define W4W_YIELD 1      // NT < 3: s_sleep(W4W_YIELD) behind every MFMA quad of an MFMA wave that shares its SIMD with a producer - a producer advances about one
                         // VALU instruction per MFMA while the MFMA wave of its SIMD streams (traced: transform 1880 clk instead of 650); 1: 56x56 64->64 88.8 -> 86.3 us, 2 / 4: slower
#endif
#ifndef W4W_DMAWAVES
#define W4W_DMAWAVES 1   // NT = 2: two extra waves (SIMDs 2 / 3) request the raw patch; 0: the producers do (as at NT = 1 / 3)
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

// s_waitcnt vmcnt(n), n wave-uniform, 0 .. 16 (more: 0 = over-wait).  The count is an immediate, so this is a dispatch: as a dense
// switch hipcc emitted a 32-way decision tree through condition-code copies (~300 clk per call on a producer's critical path, traced);
// a hand-written binary tree is four scalar compares.
__device__ __forceinline__ void w4w_wait_vm(int n) {
#define W4W_W(k) asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory")
  if (n < 8) {
    if (n < 4) { if (n < 2) { if (n == 0) W4W_W(0); else W4W_W(1); } else { if (n == 2) W4W_W(2); else W4W_W(3); } }
    else { if (n < 6) { if (n == 4) W4W_W(4); else W4W_W(5); } else { if (n == 6) W4W_W(6); else W4W_W(7); } }
  } else if (n < 16) {
    if (n < 12) { if (n < 10) { if (n == 8) W4W_W(8); else W4W_W(9); } else { if (n == 10) W4W_W(10); else W4W_W(11); } }
    else { if (n < 14) { if (n == 12) W4W_W(12); else W4W_W(13); } else { if (n == 14) W4W_W(14); else W4W_W(15); } }
  } else if (n == 16) W4W_W(16);
  else W4W_W(0);
#undef W4W_W
}

// y = A^T x for the six values x0 .. x5 of one transform row / column ([Lavin & Gray]: A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0;
// 0 1 -1 8 -8 1]) in even / odd form: 10 operations
__device__ __forceinline__ void w4w_at2(f32x2 x0, f32x2 x1, f32x2 x2, f32x2 x3, f32x2 x4, f32x2 x5, f32x2 (&y)[4]) {
  const f32x2 s1 = x1 + x2, d1 = x1 - x2, s2 = x3 + x4, d2 = x3 - x4;
  y[0] = x0 + s1 + s2;
  y[1] = pk_fma(d2, 2.f, d1);
  y[2] = pk_fma(s2, 4.f, s1);
  y[3] = pk_fma(d2, 8.f, d1) + x5;
}
// (on register PAIRS: as f32x4 arithmetic hipcc emitted the two differences as eight scalar v_sub_f32 per transform - 80 of an epilogue's ~480
// vector instructions; on f32x2 they are v_pk_add_f32 with a negated operand)
__device__ __forceinline__ void w4w_at(const f32x4& x0, const f32x4& x1, const f32x4& x2, const f32x4& x3, const f32x4& x4, const f32x4& x5,
                                       f32x4 (&y)[4]) {
#if W4W_ATPK
  f32x2 lo[4], hi[4];
  w4w_at2(x0.xy, x1.xy, x2.xy, x3.xy, x4.xy, x5.xy, lo);
  w4w_at2(x0.zw, x1.zw, x2.zw, x3.zw, x4.zw, x5.zw, hi);
#pragma unroll
  for (int i = 0; i < 4; ++i) y[i] = (f32x4){lo[i].x, lo[i].y, hi[i].x, hi[i].y};
#else
  const f32x4 s1 = x1 + x2, d1 = x1 - x2, s2 = x3 + x4, d2 = x3 - x4;
  y[0] = x0 + s1 + s2;
  y[1] = d1 + 2.f * d2;
  y[2] = s1 + 4.f * s2;
  y[3] = d1 + 8.f * d2 + x5;
#endif
}

struct W4WParams {
  W4PParams g;          // geometry, tensors, LDS offsets (xoff unused)
  FastDiv dNbn;         // walk index -> (tile strip m, n-group), n-group innermost ...
  FastDiv dNbm;         // ... or tile strip innermost
  int minner;           // 1: tile strips innermost (see conv_wino4w_launch)
};

// walk index -> the item id the shared geometry helpers expect (m + n-group * nblocks_m).  The persistent walk hands every XCD a contiguous
// range of walk indices: n-group-innermost, an XCD owns some tile strips with ALL their n-groups (their patch comes through its L2 once, the
// whole U stream once per XCD); strip-innermost, it owns some n-groups with all strips (its share of U once, every patch once per XCD).
__device__ __forceinline__ int w4w_item_id(const W4WParams& p, int it, int* ngroup) {
  if (p.minner) {
    const uint32_t n = fdiv((uint32_t)it, p.dNbm);
    *ngroup = (int)n;
    return it;                                            // (it = n * nblocks_m + m already is the helpers' id)
  }
  const uint32_t m = fdiv((uint32_t)it, p.dNbn);
  *ngroup = it - (int)m * p.g.nb_n;
  return (int)m + *ngroup * p.g.nblocks_m;
}

// ---------------------------------------------------------------------------------------------------------------------
// wave roles.  Waves land on SIMD (wave id % 4).
//   NT = 3: producers = wave ids 3 and 7, i.e. BOTH on SIMD 3, every other SIMD carries two MFMA waves (72 MFMAs = 2304 clk per slice, the
//           floor of the kernel); the producers also request the patch (an LDS-DMA piece blocks the MFMA pipe of the SIMD it is issued from
//           for ~47 clk - on SIMD 3 there is none).
//   NT = 2: MFMA waves = ids 0 .. 3 (one per SIMD), producers = ids 4, 5 (SIMDs 0 / 1), patch requesters = ids 6, 7 (SIMDs 2 / 3).
//   NT = 1: MFMA waves = ids 0, 1, producers = ids 2, 3 (their own SIMDs; they request the patch too).
// ---------------------------------------------------------------------------------------------------------------------
template <int NT> constexpr int w4w_dma_waves() { return (NT == 2 && W4W_DMAWAVES) ? 2 : 0; }
template <int NT> constexpr int w4w_block_waves() { return 2 * NT + 2 + w4w_dma_waves<NT>(); }
enum { W4W_ROLE_MFMA = 0, W4W_ROLE_PRODUCER = 1, W4W_ROLE_DMA = 2 };
template <int NT> __device__ __forceinline__ int w4w_role(int wave, int* index) {
  if constexpr (NT == 3) {
    if ((wave & 3) == 3) { *index = wave >> 2; return W4W_ROLE_PRODUCER; }
    *index = wave - (wave >> 2);
    return W4W_ROLE_MFMA;
  } else {
    if (wave < 2 * NT) { *index = wave; return W4W_ROLE_MFMA; }
    if (wave < 2 * NT + 2) { *index = wave - 2 * NT; return W4W_ROLE_PRODUCER; }
    *index = wave - 2 * NT - 2;
    return W4W_ROLE_DMA;
  }
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

// ---------------------------------------------------------------------------------------------------------------------
// The raw-patch requests (LDS-DMA, global_load_lds_dwordx4), shared by two requesters: requester dw takes pieces dw, dw + 2, ... of every
// slice, all of them in ONE exec-masked asm block (w4::dma_gather: padding lanes masked off).  A CURSOR (item, slice, ring slot) walks
// the global slice sequence RD + 1 slices ahead of the MFMAs: RD slices before the first barrier, one behind the second (the window of
// slice 0 has been read by then), one per slice from there on - raw(t + RD + 1) goes to the slot of raw(t + 1), whose window was read
// during slice t - 1.  Its window is read during slice t + RD - 1, so it must have landed at the barrier that ends slice t + RD - 2:
// at the end of a slice the requests of the last RD - 2 slices may still be in flight (wait()).
// ---------------------------------------------------------------------------------------------------------------------
template <int FLAT>
struct W4WRaw {
  static constexpr int NW = 2, MAXP = 8, RD = W4W_RD;      // (rawF4 <= 1024 slots = 16 pieces)
  static_assert(RD == 3 || RD == 4, "raw ring depth");
  int goff[MAXP];
  unsigned long long lmask[MAXP];                           // lanes of a piece that carry an in-image position
  int dw, it, s, slot, step, end, prev;
  __device__ __forceinline__ void setup(const W4WParams& pp, int lane) {      // the patch of item `it`: global float offsets of this requester's pieces
    int ng;
    const int id = w4w_item_id(pp, it, &ng);
    raw_piece_offsets<MAXP, FLAT>(pp.g, id, dw, NW, lane, goff);
#pragma unroll
    for (int k = 0; k < MAXP; ++k) lmask[k] = __ballot(goff[k] >= 0);
  }
  __device__ __forceinline__ void init(const W4WParams& pp, const Walk& wk, int dw_, int lane) {
    dw = dw_; it = wk.first; s = 0; slot = 0; step = wk.step; end = wk.end; prev = 0;
    setup(pp, lane);
  }
  // the cursor's slice -> its ring slot; returns the number of requests (wave-uniform)
  __device__ __forceinline__ int issue(const W4WParams& pp, float4* smem, int lane) {
    if (it >= end) return 0;
    const W4PParams& p = pp.g;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float4*)smem;
    const float* sbase = p.in + (size_t)(s >> 2) * p.in_ss + (s & 3) * 4;
    const unsigned sb = lds_base + (unsigned)(slot * p.rawF4) * 16u;
    const int npieces_raw = p.rawF4 >> 6;
    if (s < RD) {                                           // first use of this slot by this item: clear its padding lanes
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
    unsigned voff[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) voff[k] = (unsigned)goff[k] * 4u;
    int np = dw < npieces_raw ? (npieces_raw - 1 - dw) / NW + 1 : 0;                  // pieces of this requester (a piece whose mask is empty still issues)
    const unsigned dst0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(sb + (unsigned)dw * 1024u));
    if constexpr (W4W_STATICNP != 0) {
      np = MAXP;
      if (!(W4W_EXP & 1)) w4::dma_gather<MAXP>(sbase, voff, lmask, dst0, NW * 1024u);
    } else {
      if (!(W4W_EXP & 1)) w4w_gather_n<MAXP>(np, sbase, voff, lmask, dst0, NW * 1024u);
    }
    slot = slot + 1 == RD ? 0 : slot + 1;
    if (++s == p.nC4) {
      s = 0; it += step;
      if (it < end) setup(pp, lane);
    }
    return (W4W_EXP & 1) ? 0 : np;
  }
  // s_waitcnt vmcnt(n) for a count of this requester
  __device__ __forceinline__ static void wait_n(int n) {
    if constexpr (W4W_STATICNP != 0) {                      // n is 0, MAXP or 2 MAXP
      if (n == 2 * MAXP) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (n == MAXP) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      static_assert(MAXP == 8, "the constants above");
    } else w4w_wait_vm(n);
  }
  // Before the first barrier (P0): raw(0 .. RD - 1) requested, raw(0) and raw(1) landed (their windows are read between P0 and P1);
  // before the second (P1): raw(2) landed (its window is read during slice 0).  Younger requests stay in flight - the cold start of
  // a launch is a burst of RD slices per CU, and only the first two gate the first transform.
  int n_pro[RD];
  __device__ __forceinline__ void prologue(const W4WParams& pp, float4* smem, int lane) {
#pragma unroll
    for (int c = 0; c < RD; ++c) n_pro[c] = issue(pp, smem, lane);
    int young = 0;
#pragma unroll
    for (int c = 2; c < RD; ++c) young += n_pro[c];
    wait_n(young);
  }
  __device__ __forceinline__ void prologue_p1(const W4WParams& pp, float4* smem, int lane) {     // between P0 and P1
    int young = 0;
#pragma unroll
    for (int c = 3; c < RD; ++c) young += n_pro[c];
    wait_n(young);
  }
  // behind P1: raw(RD) -> slot 0 (the window of slice 0 has been read)
  __device__ __forceinline__ void after_p1(const W4WParams& pp, float4* smem, int lane) {
    const int n = issue(pp, smem, lane);
    // in flight now: raw(3 .. RD - 1) and raw(RD); the wait at the end of slice 0 allows raw(4 ..) = this batch (+ slice 0's own)
    prev = n;
  }
  // end of a slice in which this wave made `nvm` requests: everything older than the last RD - 2 slices' requests has landed
  __device__ __forceinline__ void wait(int nvm) {
    if constexpr (RD == 3) wait_n(nvm);
    else { wait_n(nvm + prev); prev = nvm; }
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// MFMA waves.  U straight from memory into the A-operand registers: the register ring is IN PLACE - quad q of slice t + 1 is requested right
// behind the four MFMAs that consumed quad q of slice t (an MFMA reads its A / B operands in its first passes, the load's data arrives
// hundreds of clocks later), so exactly nine loads are in flight per wave and `s_waitcnt vmcnt(8)` in front of quad q means "quad q has
// landed" (loads return in order; anything else this wave has in flight - the epilogue's loads and stores - was issued later or is
// older, so the count can only over-wait).  The loads are inline asm: hipcc neither counts them nor drains them at the slice barrier,
// which is a bare s_barrier here (the LDS reads of a slice are consumed by its MFMAs; __syncthreads() would add a vmcnt(0) behind every
// item's stores).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void w4w_uload(f32x4& dst, const void* sbase, unsigned voff, int q) {
  // quad q = byte offset q KiB: three lane-offset registers (+0 / +4 / +8 KiB) x immediate offsets 0 .. 3 KiB (13-bit signed field)
#define W4W_UL(imm) asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #imm : "+v"(dst) : "v"(voff), "s"(sbase))
  switch (q & 3) { case 0: W4W_UL(0); break; case 1: W4W_UL(1024); break; case 2: W4W_UL(2048); break; default: W4W_UL(3072); break; }
#undef W4W_UL
}

template <int NT, int FLAT>
__device__ __forceinline__ void w4w_mfma_wave(const W4WParams& pp, float4* smem, int wave, int lane) {      // wave = MFMA index 0 .. 2 NT - 1
  const W4PParams& p = pp.g;
  const int grp = wave >= NT ? 1 : 0, nn = wave - grp * NT;           // (wave-uniform)
  const int idx = lane & 15, g = lane >> 4;
  const int vlane = w4p_sigma(idx, g);
  const int S = p.nC4;
  const Walk wk = item_walk(p);
  if (wk.first >= wk.end) return;                                     // (whole block: every role takes the same exit)

  // U(0) of the block's first item: nine one-KiB quads, lane * 16 B each (the packed fragments of a (slice, n-tile) are 9 KiB in a row)
  f32x4 ur[9];
  const unsigned uvo[3] = {(unsigned)lane * 16u, (unsigned)lane * 16u + 4096u, (unsigned)lane * 16u + 8192u};
  {
    int ng0;
    (void)w4w_item_id(pp, wk.first, &ng0);
    const float4* u0 = p.ufrag + (size_t)(ng0 * NT + nn) * W4W_UBLK;
#pragma unroll
    for (int q = 0; q < 9; ++q) { ur[q] = (f32x4){0.f, 0.f, 0.f, 0.f}; if (!(W4W_EXP & 16)) w4w_uload(ur[q], u0, uvo[q >> 2], q); }
  }
  __syncthreads();                                        // P0: the first patch slices have landed
  __syncthreads();                                        // P1: V(0) is written, the windows of slices 0 and 1 are in the producers' registers

  int vb = 0;                                             // global slice t: V buffer t & 1
  for (int it = wk.first; it < wk.end; it += wk.step) {
    const bool hasB = it + wk.step < wk.end;
    int ngA, ngB = 0;
    const int idA = w4w_item_id(pp, it, &ngA);
    if (hasB) (void)w4w_item_id(pp, it + wk.step, &ngB);
    const int nt0A = ngA * NT, nt0B = ngB * NT;
    const int nt = nt0A + nn;                             // this wave's n-tile (may lie beyond the tensor in the last n-group)

    f32x4 acc[9][4];
#pragma unroll
    for (int q = (W4W_PEEL ? 9 - (NT < 3 ? W4W_HOLD : 0) : 0); q < 9; ++q)        // (W4W_PEEL: only the quads whose first MFMAs are deferred need a zeroed accumulator)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[q][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

#if W4W_TRACE
    const bool trace = blockIdx.x == 0 && it == wk.first;
    unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
#if W4W_HOLD
    constexpr int NH = NT < 3 ? W4W_HOLD : 0;               // quads held back across the slice barrier
    float4 vhold[NH > 0 ? NH : 1];
    const float4* uhold = p.ufrag;
#endif
    auto slice = [&](const int s, auto first_tag) __attribute__((always_inline)) {
      constexpr bool kFirst = W4W_PEEL && decltype(first_tag)::value;      // slice 0 of an item: C operand = literal zero
      W4W_T(c0);
      const float4* V = smem + p.voff + vb * (2 * W4W_UBLK) + grp * W4W_UBLK + vlane;
      // U of global slice t + 1: the next slice of this item, slice 0 of the block's next item (an n-group that reaches beyond the tensor -
      // Cout = 112: 7 n-tiles at NT = 3 - reads on into the next slice's fragments / the slack behind the last one,
      // conv_wino4w_packed_floats: those waves' results are never stored), or - behind the block's very last slice - a reload of the current
      // one that nobody uses (the number of loads in flight stays nine, so the vmcnt(8) below stays exact)
      const float4* un = p.ufrag + (s + 1 < S ? (size_t)(s + 1) * p.nT16 + (nt0A + nn) : hasB ? (size_t)(nt0B + nn) : (size_t)s * p.nT16 + (nt0A + nn)) * W4W_UBLK;
      constexpr int PF = W4W_PF;
      float4 vq[PF + 1];
#pragma unroll
      for (int q = 0; q < PF; ++q) vq[q] = (W4W_EXP & 4) ? make_float4(1.f, 2.f, 3.f, (float)q) : V[q * 64];
#if W4W_HOLD
      if (NH > 0 && s > 0) {                               // the previous slice's last quads (operands in registers since before the barrier)
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          constexpr int q0 = 9 - NH;
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[q0 + h][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ur[q0 + h][i], f4c(vhold[h], i), acc[q0 + h][i], 0, 0, 0);
          w4w_uload(ur[q0 + h], uhold, uvo[(q0 + h) >> 2], q0 + h);
        }
      }
#endif
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        if (q + PF < 9) vq[(q + PF) % (PF + 1)] = (W4W_EXP & 4) ? make_float4(1.f, 2.f, 3.f, (float)q) : V[(q + PF) * 64];
        const float4 v = vq[q % (PF + 1)];
        if (!(W4W_EXP & 16)) asm volatile("s_waitcnt vmcnt(8)" : "+v"(ur[q]));          // quad q of this slice has landed (eight younger loads may be in flight)
#if W4W_HOLD
        if (NH > 0 && q >= 9 - NH && s + 1 < S) { vhold[q - (9 - NH)] = v; uhold = un; continue; }      // deferred across the barrier
#endif
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (W4W_EXP & 8) acc[q][i][0] = (kFirst ? 0.f : acc[q][i][0]) + ur[q][i] * f4c(v, i);
          else if constexpr (kFirst) acc[q][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ur[q][i], f4c(v, i), (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          else acc[q][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ur[q][i], f4c(v, i), acc[q][i], 0, 0, 0);
        }
        if (!(W4W_EXP & 16)) w4w_uload(ur[q], un, uvo[q >> 2], q);                        // ... and its registers take quad q of the next slice
        if constexpr (W4W_YIELD > 0 && NT < 3) { if (wave < 2) __builtin_amdgcn_s_sleep(W4W_YIELD); }
      }
      W4W_T(c1);
      vb ^= 1;
      W4W_T(c2);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // (the slice's LDS reads were consumed by its MFMAs: free)
      __builtin_amdgcn_s_barrier();                       // everybody is done with slice t; V(t + 1), raw(t + 2) are in place
      __builtin_amdgcn_sched_barrier(0);
      W4W_T(c3);
      W4W_ACC(0, c0, c1); W4W_ACC(1, c1, c2); W4W_ACC(2, c2, c3);
    };
    slice(0, std::true_type{});
    for (int s = 1; s < S; ++s) slice(s, std::false_type{});
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
      unsigned xoffr[4];                                // FLAT == 2: the residual's column offsets (they carry an image term: its row stride may differ)
      int imgrow[4], ximg[4];                           // FLAT == 2: first image of the mosaic row / image column of the pixel
      if constexpr (FLAT == 2) {
        // mosaic (cfg.R = 4 MS): the tile's VIRTUAL rows / columns -> (image, pixel).  A tile may straddle two images and the border line
        // between them; image = (mosaic * MS + yimg) * MS + ximg, so the element offset is separable into a row and a column part
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int xc;
          mosaic_split((uint32_t)(oxb + j), p.Wp1, p.dWp1, &ximg[j], &xc);
          okx[j] = tl.valid && xc < p.W && ximg[j] < p.MS;
          const unsigned xin = (unsigned)(min(xc, p.W - 1) * 16 + g4);
          xoffb[j] = ((unsigned)(ximg[j] * p.H) * (unsigned)p.out_rs + xin) * 4u;
          xoffr[j] = ((unsigned)(ximg[j] * p.H) * (unsigned)p.res_rs + xin) * 4u;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          int yimg, yc;
          mosaic_split((uint32_t)(oyb + i), p.Hp1, p.dHp1, &yimg, &yc);
          oky[i] = yc < p.H && yimg < p.MS;
          imgrow[i] = (tl.b * p.MS + yimg) * p.MS;
          const int orow = imgrow[i] * p.H + min(yc, p.H - 1);
          ooff[i] = (unsigned)orow * (unsigned)p.out_rs * 4u;
          roff[i] = (unsigned)orow * (unsigned)p.res_rs * 4u;
        }
      } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        okx[j] = tl.valid && oxb + j < p.W;
        xoffb[j] = (unsigned)(min(oxb + j, p.W - 1) * 16 + g4) * 4u;
        xoffr[j] = xoffb[j]; ximg[j] = 0;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        oky[i] = oyb + i < p.H;
        const int orow = tl.b * p.H + min(oyb + i, p.H - 1);
        ooff[i] = (unsigned)orow * (unsigned)p.out_rs * 4u;
        roff[i] = (unsigned)orow * (unsigned)p.res_rs * 4u;
        imgrow[i] = 0;
      }
      }
      // FLAT == 2: pixel (i, j) exists iff its row and column do and its image is one of the B (the last mosaic may be partly empty)
      auto pix_ok = [&](int i, int j) __attribute__((always_inline)) -> bool {
        if constexpr (FLAT == 2) return okx[j] && oky[i] && imgrow[i] + ximg[j] < p.B;
        else return okx[j] && oky[i];
      };
      const float4 sh = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p.bias + ntc * 16) + (unsigned)(g4 * 4));
      const char* rb = reinterpret_cast<const char*>(p.res + (size_t)ntc * p.out_ss);
      char* ob = reinterpret_cast<char*>(p.out + (size_t)ntc * p.out_ss);
      const bool ntok = nt < p.nT16;
      // Z[xi][.] = A^T applied along nu to row xi of M, row by row: a row of M (24 registers) dies as soon as its row of Z (16)
      // exists - the allocator sees 144 -> 96 live values instead of 144 + 96.  A^T in its even / odd form (w4w_at: 10 packed
      // operations per 6 -> 4 transform instead of the 14 of the term-by-term sums; the epilogue is VALU-bound, two MFMA waves per SIMD)
#if W4W_RESEARLY
      float4 rr[2][4];
      auto res_load = [&](int j) __attribute__((always_inline)) {       // column j -> rr[j & 1]
        if (has_res) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            unsigned ro = roff[i] + xoffr[j];
            if constexpr (FLAT == 2) ro = pix_ok(i, j) ? ro : 0u;          // (an absent image of the last mosaic lies beyond the tensor)
            rr[j & 1][i] = *reinterpret_cast<const float4*>(rb + ro);
          }
        }
      };
      res_load(0);
      res_load(1);
#endif
      W4W_T(ea);
      f32x4 z[6][4];
#pragma unroll
      for (int xi = 0; xi < 6; ++xi) {
        w4w_at(acc[w4w_slot(xi, 0) >> 2][w4w_slot(xi, 0) & 3], acc[w4w_slot(xi, 1) >> 2][w4w_slot(xi, 1) & 3],
               acc[w4w_slot(xi, 2) >> 2][w4w_slot(xi, 2) & 3], acc[w4w_slot(xi, 3) >> 2][w4w_slot(xi, 3) & 3],
               acc[w4w_slot(xi, 4) >> 2][w4w_slot(xi, 4) & 3], acc[w4w_slot(xi, 5) >> 2][w4w_slot(xi, 5) & 3], z[xi]);
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(z[xi][j][0]), "+v"(z[xi][j][1]), "+v"(z[xi][j][2]), "+v"(z[xi][j][3]));    // (row xi is finished here)
      }
      W4W_T(eb);
      W4W_ACC(4, e0, ea); W4W_ACC(5, ea, eb);
      // one output column j at a time: Y[i][j] = sum_xi A^T[i][xi] Z[xi][j]; the residual of column j + 1 travels meanwhile
#if !W4W_RESEARLY
      float4 rr[2][4];
      if (has_res) {
#pragma unroll
        for (int i = 0; i < 4; ++i) rr[0][i] = *reinterpret_cast<const float4*>(rb + (roff[i] + xoffr[0]));
      }
#endif
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#if !W4W_RESEARLY
        if (has_res && j + 1 < 4) {
#pragma unroll
          for (int i = 0; i < 4; ++i) rr[(j + 1) & 1][i] = *reinterpret_cast<const float4*>(rb + (roff[i] + xoffr[j + 1]));
        }
#endif
        f32x4 yc[4];
        w4w_at(z[0][j], z[1][j], z[2][j], z[3][j], z[4][j], z[5][j], yc);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          f32x4 v = yc[i] + (f32x4){sh.x, sh.y, sh.z, sh.w};
#if W4W_ASMMAX
          auto clampv = [&](f32x4& x) __attribute__((always_inline)) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { float y; asm("v_max_f32 %0, %1, %2" : "=v"(y) : "v"(x[c]), "v"(lo)); x[c] = y; }
          };
          if (has_res) {
            const float4 r = rr[j & 1][i];
            const f32x4 rv = {r.x, r.y, r.z, r.w};
            if (p.res_after_act) { clampv(v); v = v + rv; }
            else { v = v + rv; clampv(v); }
          } else clampv(v);
#else
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
#endif
          if (ntok && pix_ok(i, j)) *reinterpret_cast<float4*>(ob + (ooff[i] + xoffb[j])) = make_float4(v[0], v[1], v[2], v[3]);
        }
#if W4W_RESEARLY
        if (j + 2 < 4) res_load(j + 2);                     // (its registers are free now; two columns of work ahead of its use)
#endif
      }
    }
#if W4W_TRACE
    W4W_T(e1);
    W4W_ACC(3, e0, e1);
    if (trace && lane == 0) { for (int k = 0; k < 4; ++k) g_w4w_trace[4 * wave + k] = tr[k]; g_w4w_trace[63] = (unsigned long long)S;
                              g_w4w_trace[48 + 2 * wave] = tr[4]; g_w4w_trace[49 + 2 * wave] = tr[5]; }
#endif
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// producer waves: input transform V = B^T d B, all 36 positions of tile group pw (+ the patch requests where no DMA waves exist)
// ---------------------------------------------------------------------------------------------------------------------
template <int NT, int FLAT>
__device__ __forceinline__ void w4w_producer(const W4WParams& pp, float4* smem, int pw, int lane) {
  const W4PParams& p = pp.g;
  const int grp = pw;
  const int idx = lane >> 2, g = lane & 3;   // transform lane order: 8 tiles x 4 channels per 32-lane half (see w4p_sigma)
  const int vlane = w4p_sigma(idx, g);
  const int rawF4 = p.rawF4;
  const int S = p.nC4;
  constexpr int RD = W4W_RD;
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
  // (Tried in the first layout: TWO window register sets, the window of slice t + 2 requested at the top of slice t ahead of the transform of window
  // t + 1 - 3-7 % slower per launch: the 15 reads then queue in front of the MFMA waves' first operand reads of the slice.)
  f32x2 dA[6][3];
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

  constexpr bool kIssue = w4w_dma_waves<NT>() == 0;        // this wave requests the patch itself
  W4WRaw<FLAT> raw;
  Win A, B;
  win_of(wk.first, A);
  if constexpr (kIssue) { raw.init(pp, wk, pw, lane); raw.prologue(pp, smem, lane); }
  __syncthreads();                                        // P0: raw(0), raw(1) of the first item have landed
  load_window(dA, A, 0);
  transform(dA, 0);
  if (S > 1) load_window(dA, A, 1);                       // dA = window of slice t + 1 at the top of slice t
  if constexpr (kIssue) raw.prologue_p1(pp, smem, lane);  // raw(2) has landed
  __syncthreads();                                        // P1
  if constexpr (kIssue) raw.after_p1(pp, smem, lane);     // raw(RD) -> slot 0 (the window of slice 0 has been read)
  // fp32 MFMAs run on the SIMD's vector ALUs: without a higher issue priority an MFMA wave of this SIMD starves this wave's
  // VALU / LDS instructions until it reaches the slice barrier
  __builtin_amdgcn_s_setprio(3);
  int ring = 0, vb = 0;                                   // global slice t: t % RD, t & 1
  for (int it = wk.first; it < wk.end; it += wk.step) {
    const bool hasB = it + wk.step < wk.end;
    if (hasB) win_of(it + wk.step, B);
#if W4W_TRACE
    const bool trace = blockIdx.x == 0 && it == wk.first;
    unsigned long long tr[5] = {0, 0, 0, 0, 0};
#endif
    for (int s = 0; s < S; ++s) {
      W4W_T(q0);
      const int r1 = ring + 1 == RD ? 0 : ring + 1, r2 = r1 + 1 == RD ? 0 : r1 + 1;
      int nvm = 0;
      if constexpr (kIssue) nvm = raw.issue(pp, smem, lane);          // raw(t + RD + 1) -> slot r1
      W4W_T(qa);
      // V(t + 1) from the window fetched during the previous slice, then the window of slice t + 2 (raw(t + 2) landed before the
      // barrier that ended slice t - 1) - of this item or of the next one
      if (!(W4W_EXP & 2)) {
        if (s + 1 < S || hasB) transform(dA, vb ^ 1);
      }
      W4W_T(q1);
      if (!(W4W_EXP & 2)) {
        if (s + 2 < S) load_window(dA, A, r2);
        else if (hasB) load_window(dA, B, r2);
      }
      W4W_T(qb);
      if constexpr (kIssue) raw.wait(nvm);
      ring = r1;
      vb ^= 1;
      W4W_T(q2);
      __syncthreads();
      W4W_T(q3);
      W4W_ACC(0, q0, qa); W4W_ACC(1, qa, q1); W4W_ACC(2, q1, qb); W4W_ACC(3, qb, q2); W4W_ACC(4, q2, q3);
    }
#if W4W_TRACE
    if (trace && lane == 0) for (int k = 0; k < 5; ++k) g_w4w_trace[32 + 5 * pw + k] = tr[k];
#endif
    if (hasB) A = B;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// patch requesters (NT = 2): nothing but the LDS-DMA of the coming slices, in step with the block's barriers
// ---------------------------------------------------------------------------------------------------------------------
template <int NT, int FLAT>
__device__ __forceinline__ void w4w_dma_wave(const W4WParams& pp, float4* smem, int dw, int lane) {
  const W4PParams& p = pp.g;
  const Walk wk = item_walk(p);
  if (wk.first >= wk.end) return;
  W4WRaw<FLAT> raw;
  raw.init(pp, wk, dw, lane);
  raw.prologue(pp, smem, lane);
  __syncthreads();                                        // P0
  raw.prologue_p1(pp, smem, lane);
  __syncthreads();                                        // P1
  raw.after_p1(pp, smem, lane);
#if W4W_TRACE
  unsigned long long tr[2] = {0, 0};
  int nsl = 0;
#endif
  for (int it = wk.first; it < wk.end; it += wk.step)
    for (int s = 0; s < p.nC4; ++s) {
      W4W_T(d0);
      const int nvm = raw.issue(pp, smem, lane);
      W4W_T(d1);
      raw.wait(nvm);
      __syncthreads();
      W4W_T(d2);
#if W4W_TRACE
      if (blockIdx.x == 0 && it == wk.first) { tr[0] += d1 - d0; tr[1] += d2 - d1; ++nsl; }
#endif
    }
#if W4W_TRACE
  if (blockIdx.x == 0 && lane == 0) { g_w4w_trace[58 + 2 * dw] = tr[0]; g_w4w_trace[59 + 2 * dw] = tr[1]; }
#endif
}

template <int NT, int FLAT>      // FLAT: 0 rectangular items, 1 flat items
__global__ void __launch_bounds__(64 * w4w_block_waves<NT>())
conv_wino4w_kernel(const W4WParams p) {
  extern __shared__ float4 smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int index;
  const int role = w4w_role<NT>(wave, &index);
  if (role == W4W_ROLE_MFMA) w4w_mfma_wave<NT, FLAT>(p, smem, index, lane);
  else if (role == W4W_ROLE_PRODUCER) w4w_producer<NT, FLAT>(p, smem, index, lane);
  else if constexpr (w4w_dma_waves<NT>() > 0) w4w_dma_wave<NT, FLAT>(p, smem, index, lane);
}

struct W4WLayout { int uoff, voff, totalF4; };
bool w4w_geo(const ConvDesc& d, const ConvCfg& cfg, w4::Geo* g, W4WLayout* L, FlatGeo* fg = nullptr) {
  if (d.ks != 3 || d.stride != 1 || cfg.NT < 1 || cfg.NT > 3 || cfg.WM != 2 || cfg.WN != 1 || d.Cin % 16 || d.Cout % 16) return false;
  FlatGeo ftmp;
  if (cfg.NI == 0) { if (!flat_geo(d, cfg, g, fg ? fg : &ftmp)) return false; }                     // flat items; cfg.R = 4 MS: over mosaics of MS x MS images
  else if (!w4::geo(d, cfg, 32, g)) return false;
  if (g->rawF4 > 1024) return false;
  // 32-bit byte offsets in the epilogue
  if ((long)d.B * d.H * d.W * std::max(std::max(d.in_cs, d.out_cs), d.res_cs) >= (1L << 30)) return false;
  L->uoff = W4W_RD * g->rawF4;                                  // (no U ring: the MFMA waves load U into registers)
  L->voff = L->uoff;
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
                   "NI*(R/4)*ceil(W/4) <= 32 tiles (or R = 4 MS, NI = 0: flat items), a patch of <= 1024 slots, tensors below 2^30 elements and the ALG 13 weight fragments");
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
  pp.dNbm = make_fastdiv(p.nblocks_m);
  // Walk order by what it costs in L2 fills (8 XCDs, each with its own 4 MB L2): n-group-innermost = activations once + 8 x the U stream,
  // strip-innermost = U once + 8 x the activations.  56x56 48->48: U 0.3 MB against 38 MB of activations; 7x7 384->384: U 21 MB against 4.8 MB
  // (PMC, 64 crops: 182 MB per launch n-group-innermost)
  {
    const double ubytes = 36.0 * d.Cin * d.Cout * 4.0, abytes = (double)d.B * d.H * d.W * d.Cin * 4.0;
    pp.minner = (p.nb_n > 1 && ubytes > abytes) ? 1 : 0;
  }
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
  const int mode = !flat ? 0 : fg.MS > 1 ? 2 : 1;
  void (*fn)(const W4WParams) =
      mode == 2 ? (cfg.NT == 3 ? conv_wino4w_kernel<3, 2> : cfg.NT == 2 ? conv_wino4w_kernel<2, 2> : conv_wino4w_kernel<1, 2>)
      : mode == 1 ? (cfg.NT == 3 ? conv_wino4w_kernel<3, 1> : cfg.NT == 2 ? conv_wino4w_kernel<2, 1> : conv_wino4w_kernel<1, 1>)
                  : (cfg.NT == 3 ? conv_wino4w_kernel<3, 0> : cfg.NT == 2 ? conv_wino4w_kernel<2, 0> : conv_wino4w_kernel<1, 0>);
  if (lds > 64 * 1024) {
    static thread_local bool configured[16] = {};
    if (!configured[cfg.NT + 4 * mode]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) { poco_set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return POCO_ERR_HIP; }
      configured[cfg.NT + 4 * mode] = true;
    }
  }
  hipLaunchKernelGGL(fn, dim3((unsigned)g4, 1), dim3(64 * (cfg.NT == 3 ? w4w_block_waves<3>() : cfg.NT == 2 ? w4w_block_waves<2>() : w4w_block_waves<1>())), lds, stream, pp);
  POCO_HIP_CHECK(hipGetLastError());
  return POCO_OK;
}

#if W4W_TRACE
extern "C" int poco_w4w_trace(unsigned long long* host_out, int n) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_w4w_trace), sizeof(unsigned long long) * (size_t)std::min(n, 64)) == hipSuccess ? 0 : 1;
}
#endif
