// ALG 8: Winograd F(4x4,3x3) on the fp32 MFMA with SPECIALISED WAVES (3x3 stride-1 convs on planes >= 28x28;
// pocolib/models/backbone/hrnet.py:42-58, hrnet_cls.py BasicBlock convs).
//
// Same arithmetic and block geometry as ALG 7 (conv_wino4.hip): 36 position GEMMs M_p[co][tile] += U_p[co][ci] V_p[ci][tile]
// per 4x4 output tile, K walked in slices of four input channels = one v_mfma_f32_16x16x4_f32 per (position, n-tile), a
// block owns two groups of 16 tiles, a wave the 9 positions 9q..9q+8 of one group.  What differs is WHO does what.
//
// ALG 7 lets every wave read its tiles' 6x6 windows, transform them and then issue its MFMAs; its K loop is a serial
// chain per wave and slice (ISA: ~28 ds_read_b32 -> s_waitcnt lgkmcnt(0) -> ~90 VALU -> 27 MFMAs with nine more
// lgkmcnt(0) stalls for the U fragments in between), the two waves of a SIMD run that chain in lockstep, and the MFMA
// pipe is busy 45 % of the loop (14 us per 16 channels against a 5.9 us MFMA floor).
//
// Here a block is 12 waves:
//   * waves 0-7, the MFMA waves (q = wave & 3, group = wave >> 2): per slice they read 3 + 3 NT operand vectors
//     (V and U, both stored in LDS in exactly the lane order of the MFMA B / A operands: two float4 quads + one float
//     per wave and n-tile) and issue 9 NT MFMAs.  No window reads, no transform, no vector-memory instruction, no
//     address arithmetic: ~50 instructions per 27 MFMAs instead of ~190.
//   * waves 8-11, the producers (group = (wave-8) >> 1, transform rows 3h..3h+2 for h = (wave-8) & 1): they stream the
//     raw patch and the U fragments of the coming slices into LDS rings with LDS-DMA (global_load_lds_dwordx4), and
//     compute V = B^T d B of the NEXT slice for their group ONCE (a lane = one (tile, channel) pair in the order channel + 4 tile:
//     15 ds_read2_b32 fetch the window as column pairs, 18 + 21 packed-fp32 instructions transform it, 4 ds_write_b128 + 2
//     ds_write_b32 store the 18 positions in the pair order of the results, see w4p_nu / w4p_sigma) while the MFMA waves work
//     on the current one.  Their
//     VALU / LDS / VMEM instructions issue in the slots the MFMA pipe leaves free on their SIMD (one MFMA occupies the
//     pipe for 8 issue slots).  In ALG 7 the four waves of a group each read the whole window (144 reads per pair).
//   * one s_barrier per slice hands V(s+1), U(s+1) and raw(s+2) over; raw ring 3 deep (LDS-DMA issued four slices ahead,
//     waited for with a counted s_waitcnt vmcnt(n) by the wave that issued it), U and V double-buffered.
//   * end of item: as ALG 7 - the MFMA waves exchange Z = M A through LDS one n-tile at a time and every wave finishes
//     one output row of the 4x4 blocks (bias, residual, ReLU, 16-byte stores); the exchange area overlays ring slots
//     that are dead by then.  The producers fetch the next item's first slices in the meantime.
//
// Weights: U = G g G^T in float64 on the host (BN scale folded), packed per (4-channel slice, n-tile) as one 9 KiB block
// in LDS order: [q = 0..3][2 quads][64 lanes] float4 (positions 9q..9q+7), then [q][64 lanes] float (position 9q+8);
// lane = (co & 15) + 16 (ci & 3).  One block = nine 1 KiB LDS-DMA pieces.
#include "conv_wino4_common.h"
#include <cstdlib>

namespace {

using w4::at_c;
using w4::bt_row;

#include "conv_wino4p_geo.h"

#ifndef W4P_EXP
#define W4P_EXP 0     // timing probes (tools/build_exp.sh conv_wino4p.hip W4P_EXP n; results are then garbage): 1 producers skip the
#endif                // transform, 2 no LDS-DMA inside the K loop, 4 MFMA waves skip the MFMAs, 8 ... skip their operand reads,
                      // 16 no per-slice barrier work at all in the producers (neither DMA nor transform), 128 producers skip the window
                      // reads only, 512 / 1024 no U / no patch LDS-DMA inside the K loop
#ifndef W4P_EPI
#define W4P_EPI 15     // round 5 epilogue / item-start restructure (bit mask; 0 = the round-4 kernel, for same-box A/B builds):
#endif                //  1: the exchange area sits at the END of the LDS (over U slot 2 + V) instead of over U slots 1, 2: U(1) of the next item
                      //     is fetched together with U(0) during the exchange rounds, no exposed fetch at the item start
                      //  2: bias / residual loads of round n+1 are issued BEFORE the stores of round n and awaited before them (round 0's after
                      //     its Z is written): no vmcnt wait of a round covers the stores of the previous round any more
                      //  8: the U pieces of a slice requested as one streamed asm block per wave (w4::dma_stream)
                      //  4: no barrier in front of exchange round 0 (the last slice barrier already separates the last operand reads from the
                      //     exchange writes), and the producers deal the next item's first fetches over the 2 NT barrier gaps of the exchange
                      //     instead of issuing all of them in front of its first barrier (where the eight MFMA waves waited for them)
#ifndef W4P_NINNER
#define W4P_NINNER 1  // items walked n-group-innermost (conv_wino4p_geo.h: w4p_item_id); 0 = the round-4 order, for A/B builds
#endif
#ifndef W4P_TRACE
#define W4P_TRACE 0   // 1: block 0 accumulates s_memtime phase sums of its 8 MFMA waves and 4 producer waves (tools/w4p_trace.py; reading the
                      // counter drains lgkmcnt, so a phase that ends with LDS reads in flight includes their latency)
#endif
#ifndef W4P_TRACE_ITEM
#define W4P_TRACE_ITEM 0   // which item of block 0's walk is traced (0 = the first, cold one)
#endif
#if W4P_TRACE
__device__ unsigned long long g_w4p_trace[96];
#define W4P_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define W4P_ACC(slot, a, b) do { if (trace) tr[slot] += (b) - (a); } while (0)
#else
#define W4P_T(var)
#define W4P_ACC(slot, a, b)
#endif
constexpr int W4P_NCONS = 8, W4P_NPROD = 4;
constexpr int W4P_MAXP = 2;          // raw-patch DMA pieces per MFMA wave (npos slots <= 8 * 2 * 64)
constexpr int W4P_UBLK = 9 * 64;     // float4 per (slice, n-tile) block of U; V of one group has the same shape
constexpr int W4P_XCH = 8 * 2 * 4 * 64;   // float4 of the Z exchange area (64 KiB)

// Which Winograd position (xi, nu) sits in slot i (0..3 = first quad, 4..7 = second quad, 8 = the single) of MFMA wave q.
// Wave q owns nine positions of transform rows RA(q) = (0, 1, 3, 4)[q] and RA(q) + 1, as in ALG 7, but INSIDE a wave the slots
// follow the register pairs the packed-fp32 input transform produces per row - (nu0, nu5), (nu1, nu3), (nu2, nu4) - so that
// the producers write their v_pk_* results to LDS without a single v_mov.  Rows r = 0, 1, 2 of a producer (xi = 3 RH + r):
//   even q = 2 RH    : [r0: nu 0 5 1 3] [r0: nu 2 4 | r1: nu 0 5] (r1: nu 1)
//   odd  q = 2 RH + 1: [r1: nu 2 4 | r2: nu 0 5] [r2: nu 1 3 2 4] (r1: nu 3)
__host__ __device__ constexpr int w4p_ra(int q) { return (9 * q) / 6; }
__host__ __device__ constexpr int w4p_row(int q, int i) {                     // 0 / 1: transform row RA(q) + that
  return (q & 1) ? ((i < 2 || i == 8) ? 0 : 1) : (i < 6 ? 0 : 1);
}
__host__ __device__ constexpr int w4p_nu(int q, int i) {
  constexpr int E[9] = {0, 5, 1, 3, 2, 4, 0, 5, 1}, O[9] = {2, 4, 0, 5, 1, 3, 2, 4, 3};
  return (q & 1) ? O[i] : E[i];
}


// ---------------------------------------------------------------------------------------------------------------------
// MFMA waves (they also issue the LDS-DMA of the coming slices in the shadow of their MFMAs: a wave that is stuck behind a
// busy MFMA pipe issues vector-memory instructions for free, while a producer wave needs ~60 clk per such instruction)
// ---------------------------------------------------------------------------------------------------------------------
template <int NT, int Q, int FLAT>
__device__ __forceinline__ void w4p_consumer(const W4PParams& p, float4* smem, int grp, int lane, int wave) {
  // this wave's nine positions: w4p_row / w4p_nu (transform rows RA(Q) and RA(Q) + 1)
  const int idx = lane & 15, g = lane >> 4;
  const int vlane = w4p_sigma(idx, g);      // this lane's slot in the V operand vectors
  const int uF4 = NT * W4P_UBLK;
  const int rawF4 = p.rawF4;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float4*)smem;
  const int npieces_raw = rawF4 >> 6;
  const Walk wk = item_walk(p);
  for (int item = wk.first; item < wk.end; item += wk.step) {
    const int id = W4P_NINNER ? w4p_item_id(p, item) : item;
    const int nt0 = (id / p.nblocks_m) * NT;
    const Tile tl = tile_of<FLAT>(p, id, grp, idx);
    // ---- staging duties of this wave: raw pieces wave, wave + 8 and U pieces wave, wave + 8, ... ------------------------------
    int goff[W4P_MAXP];
    raw_piece_offsets<W4P_MAXP, FLAT>(p, id, wave, W4P_NCONS, lane, goff);
    bool live[W4P_MAXP];                                    // pieces with at least one in-image position (wave-uniform)
#pragma unroll
    for (int k = 0; k < W4P_MAXP; ++k) live[k] = __ballot(goff[k] >= 0) != 0ull;
    // (padding lanes of a piece are switched off in the DMA: the producers zeroed those slots for the whole item, so the source
    // is ONE wave-uniform base + a 32-bit lane offset - no 64-bit VALU address arithmetic, no per-lane source select)
    auto issue_raw = [&](int c4, int slot) __attribute__((always_inline)) -> int {   // 4-channel slice c4 of the patch -> raw ring slot
      int cnt = 0;
      const float* sbase = p.in + (size_t)(c4 >> 2) * p.in_ss + (c4 & 3) * 4;
      const unsigned sb = lds_base + (unsigned)(slot * rawF4) * 16u;
#pragma unroll
      for (int k = 0; k < W4P_MAXP; ++k) {
        const int piece = wave + W4P_NCONS * k;
        if (piece < npieces_raw && live[k]) {
          if (goff[k] >= 0)
            w4::dma16_sv(sbase, (unsigned)goff[k] * 4u, (unsigned)__builtin_amdgcn_readfirstlane((int)(sb + (unsigned)piece * 1024u)));
          ++cnt;
        }
      }
      return cnt;
    };
    auto issue_u = [&](int c4, int slot) __attribute__((always_inline)) -> int {     // U of slice c4, n-tiles nt0.. -> U ring slot
      if constexpr ((W4P_EPI & 8) != 0) {
        // round 5: the 9 NT pieces of a (slice, n-group) are contiguous in the packed fragments, so this wave's pieces wave, wave + 8, ...
        // go out as ONE asm block (w4::dma_stream: per piece s_add_u32 m0 / s_nop / global_load_lds_dwordx4 instead of a dma16_sv call
        // with its own 64-bit scalar address arithmetic and m0 save / restore).  An n-group that reaches beyond the tensor reads on into
        // the next slice's fragments / the slack behind the last one (conv_wino4p_packed_floats); those n-tiles are never stored.
        constexpr int NUP = (9 * NT + W4P_NCONS - 1) / W4P_NCONS;
        const unsigned dst0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + (unsigned)(p.uoff + slot * uF4) * 16u + (unsigned)wave * 1024u));
        const float4* src = p.ufrag + (((size_t)c4 * p.nT16 + nt0) * 9 + wave) * 64;
        unsigned voff[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) voff[k] = (unsigned)lane * 16u + (unsigned)(k * W4P_NCONS) * 1024u;
        const bool full = wave + W4P_NCONS * (NUP - 1) < 9 * NT;            // (wave-uniform)
        if (full) w4::dma_stream<NUP>(src, voff, dst0, W4P_NCONS * 1024u);
        else if constexpr (NUP > 1) w4::dma_stream<(NUP > 1 ? NUP - 1 : 1)>(src, voff, dst0, W4P_NCONS * 1024u);
        return full ? NUP : NUP - 1;
      }
      int cnt = 0;
      const unsigned sb = lds_base + (unsigned)(p.uoff + slot * uF4) * 16u;
#pragma unroll
      for (int i0 = 0; i0 < 9 * NT; i0 += W4P_NCONS) {
        const int i = i0 + wave;
        if (i < 9 * NT) {
          const int n = i / 9, j = i - n * 9;
          const float4* src = p.ufrag + (((size_t)c4 * p.nT16 + min(nt0 + n, p.nT16 - 1)) * 9 + j) * 64;
          w4::dma16_sv(src, (unsigned)lane * 16u, (unsigned)__builtin_amdgcn_readfirstlane((int)(sb + (unsigned)i * 1024u)));
          ++cnt;
        }
      }
      return cnt;
    };
    const int S = p.nC4;
    f32x4 acc[9][NT];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

#if W4P_TRACE
    const bool trace = blockIdx.x == 0 && item == wk.first + W4P_TRACE_ITEM * wk.step;
    unsigned long long tr[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    W4P_T(c_a);
    // (the producers fetched raw(0..2), U(0) during the previous item's exchange rounds and U(1) just now)
    __syncthreads();                                        // B0: the first fetches have landed (and the padding slots are zero)
    __syncthreads();                                        // B1: V(0) is written; the windows of slices 0 and 1 are in the producers' registers
    if (S > 3) issue_raw(3, 0);
    W4P_T(c_b);
    W4P_ACC(0, c_a, c_b);
    int ring = 0;                                           // s % 3
    for (int s = 0; s < S; ++s) {
      W4P_T(c0);
      const int r1 = ring == 2 ? 0 : ring + 1, r2 = r1 == 2 ? 0 : r1 + 1;
      int nvm = 0;
      const float4* U = smem + p.uoff + ring * uF4 + (2 * Q) * 64 + lane;
      const float4* V = smem + p.voff + (s & 1) * (2 * W4P_UBLK) + grp * W4P_UBLK + (2 * Q) * 64 + vlane;
      const bool noread = (W4P_EXP & 8) != 0;
      const float4 vq0 = noread ? make_float4(1.f, 2.f, 3.f, (float)s) : V[0], vq1 = noread ? make_float4(1.f, 2.f, 3.f, 4.f) : V[64];
      const float vs = noread ? 2.f : reinterpret_cast<const float*>(V - (2 * Q) * 64 - vlane + 512)[Q * 64 + vlane];
#pragma unroll
      for (int n = 0; n < NT; ++n) {                        // U of one n-tile at a time: 9 operand registers live, not 27
        const float4 uq0 = noread ? make_float4(1.f, 2.f + n, 3.f, (float)s) : U[n * W4P_UBLK];
        const float4 uq1 = noread ? make_float4(1.f, 2.f, 3.f + n, 4.f) : U[n * W4P_UBLK + 64];
        const float us = noread ? 3.f : reinterpret_cast<const float*>(U - (2 * Q) * 64 - lane + n * W4P_UBLK + 512)[Q * 64 + lane];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
          const float a = i < 4 ? f4c(uq0, i) : i < 8 ? f4c(uq1, i - 4) : us;
          const float v = i < 4 ? f4c(vq0, i) : i < 8 ? f4c(vq1, i - 4) : vs;
          if (W4P_EXP & 4) acc[i][n][0] += a * v;
          else acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, v, acc[i][n], 0, 0, 0);
        }
        // The LDS-DMA of the coming slices goes out AFTER the first MFMAs of the slice are queued: a vector-memory instruction
        // takes this wave ~60 clk to issue; behind queued MFMAs that is hidden, ahead of them the pipe would idle.
        if (n == 0 && !(W4P_EXP & 64)) {
          __builtin_amdgcn_sched_barrier(0);
          if (s + 2 < S && !(W4P_EXP & (2 | 512))) nvm += issue_u(s + 2, r2);       // slot of U(s-1): consumed before the barrier that ended iteration s-1
          if (s + 4 < S && !(W4P_EXP & (2 | 1024))) nvm += issue_raw(s + 4, r1);     // slot of raw(s+1): its window was read during iteration s-1
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      W4P_T(c05);
      wait_vm(nvm);                                         // what this wave issued BEFORE this iteration has landed: U(s+1), raw(s+3)
      ring = r1;
      W4P_T(c1);
      __syncthreads();                                      // everybody is done with slice s; V(s+1), U(s+1), raw(s+2) are in place
      W4P_T(c2);
      W4P_ACC(1, c0, c1); W4P_ACC(2, c1, c2); W4P_ACC(4, c05, c1);
    }
    W4P_T(c_e0);

#if W4P_EPI & 2
    // ---- Z = M A (partial over this wave's columns), exchanged one n-tile at a time; wave Q finishes output column Q --------
    // exchange area [8 waves][2 rows][4 cols][64] float4 (64 KiB) over ring slots that are dead now.
    // Round 5: what a round waits for.  Round 4 requested bias and residual of round n after its second barrier and consumed them
    // right away: loads and stores share the one in-order VM counter on gfx9, so that wait (vmcnt(0)) also covered the four stores of
    // round n-1 - issued a few hundred clocks earlier by all 256 CUs at once - and every round paid a store round trip under burst
    // load (s_memtime: 4.5 k clk per round for ~110 instructions per wave).  Now the loads of round n+1 go out BEFORE the stores of
    // round n and are awaited before them too (an empty asm that "uses" the registers pins hipcc's wait there: newest operations,
    // so vmcnt(0) at that point covers only stores issued a whole round ago), and round 0's go out as soon as its Z is written
    // (36 accumulators dead).  The store geometry of the wave's column does not depend on the n-tile and is computed once.
    const int oyb = tl.oy0, oxb = 4 * tl.tx;
    float4* xch = smem + p.xoff;
    const int g4 = g * 4;
    const float lo = p.act == 1 ? 0.f : -INFINITY;            // ReLU as a clamp: no branch in the store loop
    const bool has_res = p.res != nullptr;
    constexpr int jc = Q;
    const int ox = oxb + jc;
    int xo, ximg = 0;
    bool okx;
    if constexpr (FLAT == 2) {
      // mosaic: the tile's virtual rows / column -> (image, pixel); a tile may straddle two images and the border line between them
      int xc;
      mosaic_split((uint32_t)ox, p.Wp1, p.dWp1, &ximg, &xc);
      okx = tl.valid && xc < p.W && ximg < p.MS;
      xo = min(xc, p.W - 1) * 16;
    } else {
      xo = min(ox, p.W - 1) * 16;
      okx = tl.valid && ox < p.W;
    }
    // per-lane 32-bit BYTE offsets of the wave's four output rows (clamped: dead pixels compute harmlessly and are masked at the
    // store); the n-tile's channel slice is added to the wave-uniform base, so every access is SGPR base + VGPR offset and no
    // 64-bit address lives in (or is spilled from) the vector registers
    unsigned ooff[4], roff[4];
    bool oky[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int orow;
      if constexpr (FLAT == 2) {
        int yimg, yc;
        mosaic_split((uint32_t)(oyb + i), p.Hp1, p.dHp1, &yimg, &yc);
        const int img = (tl.b * p.MS + yimg) * p.MS + ximg;
        oky[i] = yc < p.H && yimg < p.MS && img < p.B;
        orow = oky[i] && okx ? img * p.H + yc : 0;
      } else {
        oky[i] = oyb + i < p.H;
        orow = tl.b * p.H + min(oyb + i, p.H - 1);
      }
      ooff[i] = ((unsigned)orow * (unsigned)p.out_rs + (unsigned)(g4 + xo)) * 4u;
      roff[i] = has_res ? ((unsigned)orow * (unsigned)p.res_rs + (unsigned)(g4 + xo)) * 4u : 0u;
    }
    // bias + residual of n-tile nt0 + nn (clamped into the tensor: the last group of an item may be partly empty).  Without a
    // residual the four loads read the 16-byte zero page: no branch, no phi copies (hipcc waited for the loads inside the branch)
    const unsigned boff = (unsigned)(g4 * 4);
    auto issue_loads = [&](int nn, float4& sh, float4 (&rr)[4]) __attribute__((always_inline)) {
      const int nt = __builtin_amdgcn_readfirstlane(min(nt0 + nn, p.nT16 - 1));
      sh = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p.bias + nt * 16) + boff);
      const char* rb = has_res ? reinterpret_cast<const char*>(p.res + (size_t)nt * p.out_ss) : reinterpret_cast<const char*>(g_zero_page_w4p);
#pragma unroll
      for (int i = 0; i < 4; ++i) rr[i] = *reinterpret_cast<const float4*>(rb + roff[i]);
    };
    auto pin = [&](float4& sh, float4 (&rr)[4]) __attribute__((always_inline)) {      // hipcc's wait for these registers goes HERE
      asm volatile("" : "+v"(sh.x), "+v"(sh.y), "+v"(sh.z), "+v"(sh.w));
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(rr[i].x), "+v"(rr[i].y), "+v"(rr[i].z), "+v"(rr[i].w));
    };
    float4 sh, rr[4];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      W4P_T(e0);
      if (n > 0 || !(W4P_EPI & 4)) __syncthreads();         // previous round's reads finished everywhere (round 0: the last slice barrier did that)
      W4P_T(e1);
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int i = 0; i < 9; ++i) {
            const int nu = w4p_nu(Q, i);
            if (w4p_row(Q, i) == r && at_c(j, nu) != 0.f) z += at_c(j, nu) * acc[i][n];
          }
          xch[((wave * 2 + r) * 4 + j) * 64 + lane] = make_float4(z[0], z[1], z[2], z[3]);
        }
      if (n == 0) issue_loads(0, sh, rr);
      W4P_T(e2);
      __syncthreads();
      W4P_T(e3);
      W4P_ACC(8, e0, e1); W4P_ACC(9, e1, e2); W4P_ACC(10, e2, e3);
      // the next round's loads: right behind the barrier when the registers allow it (<= 36 accumulators still live), else after
      // this round's output transform (NT = 3, round 0: their latency is exposed once per item)
      const bool early = (NT - 1 - n) <= 1;
      float4 shn, rrn[4];
      if (n + 1 < NT && early) issue_loads(n + 1, shn, rrn);
      // Every wave finishes one output COLUMN j = Q of the 4x4 blocks: Y[i][Q] = sum_k A^T[i][k] Z[k][Q] needs the six rows of Z
      // for that one column only = 8 exchange reads (rows 1 and 4 are split between two waves) instead of the 32 an output
      // row would need (the exchange rounds are LDS-bandwidth-bound: 8 KiB written + 8 KiB read per wave and round).
      // rows of Z: 0 = q0.r0 | 1 = q0.r1 + q1.r0 | 2 = q1.r1 | 3 = q2.r0 | 4 = q2.r1 + q3.r0 | 5 = q3.r1
      const int w0 = grp * 4;
      auto ld = [&](int q, int r) {
        const float4 z = xch[(((w0 + q) * 2 + r) * 4 + jc) * 64 + lane];
        return (f32x4){z.x, z.y, z.z, z.w};
      };
      f32x4 zr[6];
      zr[0] = ld(0, 0); zr[1] = ld(0, 1) + ld(1, 0); zr[2] = ld(1, 1);
      zr[3] = ld(2, 0); zr[4] = ld(2, 1) + ld(3, 0); zr[5] = ld(3, 1);
      f32x4 y[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 v = (f32x4){sh.x, sh.y, sh.z, sh.w};
#pragma unroll
        for (int k = 0; k < 6; ++k)
          if (at_c(i, k) != 0.f) v += at_c(i, k) * zr[k];
        y[i] = v;
      }
      // residual before the activation (BasicBlock, hrnet.py:42-58) or behind it (hrnet_cls.py:475-477): a wave-uniform BRANCH - as
      // two selects per value (what hipcc makes of `if (flag) v += r` on both sides of the clamp) it was 32 v_cndmask per round
      if (p.res_after_act) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 r = rr[i];
          f32x4 v = y[i];
          v[0] = fmaxf(v[0], lo) + r.x; v[1] = fmaxf(v[1], lo) + r.y; v[2] = fmaxf(v[2], lo) + r.z; v[3] = fmaxf(v[3], lo) + r.w;
          y[i] = v;
          asm volatile("" : "+v"(y[i][0]), "+v"(y[i][1]), "+v"(y[i][2]), "+v"(y[i][3]));
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 r = rr[i];
          f32x4 v = y[i];
          v[0] = fmaxf(v[0] + r.x, lo); v[1] = fmaxf(v[1] + r.y, lo); v[2] = fmaxf(v[2] + r.z, lo); v[3] = fmaxf(v[3] + r.w, lo);
          y[i] = v;
          asm volatile("" : "+v"(y[i][0]), "+v"(y[i][1]), "+v"(y[i][2]), "+v"(y[i][3]));      // (computed here, not sunk behind the wait below)
        }
      }
      W4P_T(e4);
      if (n + 1 < NT) {
        if (!early) issue_loads(n + 1, shn, rrn);
        pin(shn, rrn);
      }
      W4P_T(e5);
      if (nt0 + n < p.nT16) {
        char* ob = reinterpret_cast<char*>(p.out + (size_t)(nt0 + n) * p.out_ss);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          // (streaming stores - __builtin_nontemporal_store - make a launch 1.5 us faster alone and the whole forward 0.5 % slower:
          // the next conv finds less of its input in the L2.  W4P_EXP & 2048: probe without the stores.)
          if (okx && oky[i] && !((W4P_EXP & 2048) && p.B < 100000))
            *reinterpret_cast<float4*>(ob + ooff[i]) = make_float4(y[i][0], y[i][1], y[i][2], y[i][3]);
        }
      }
      W4P_T(e6);
      W4P_ACC(11, e3, e4); W4P_ACC(12, e4, e5); W4P_ACC(13, e5, e6);
      if (n + 1 < NT) {
        sh = shn;
#pragma unroll
        for (int i = 0; i < 4; ++i) rr[i] = rrn[i];
      }
    }
#else
    // ---- Z = M A (partial over this wave's columns), exchanged one n-tile at a time; wave Q finishes output column Q --------
    // exchange area [8 waves][2 rows][4 cols][64] float4 (64 KiB) over ring slots that are dead now
    const int oyb = tl.oy0, oxb = 4 * tl.tx;
    float4* xch = smem + p.xoff;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      W4P_T(e0);
      if (n > 0 || !(W4P_EPI & 4)) __syncthreads();         // previous round's reads finished everywhere
      W4P_T(e1);
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int i = 0; i < 9; ++i) {
            const int nu = w4p_nu(Q, i);
            if (w4p_row(Q, i) == r && at_c(j, nu) != 0.f) z += at_c(j, nu) * acc[i][n];
          }
          xch[((wave * 2 + r) * 4 + j) * 64 + lane] = make_float4(z[0], z[1], z[2], z[3]);
        }
      W4P_T(e2);
      __syncthreads();
      W4P_T(e3);
      W4P_ACC(8, e0, e1); W4P_ACC(9, e1, e2); W4P_ACC(10, e2, e3);
      if (nt0 + n < p.nT16) {
        // Every wave finishes one output COLUMN j = Q of the 4x4 blocks: Y[i][Q] = sum_k A^T[i][k] Z[k][Q] needs the six rows of Z
        // for that one column only = 8 exchange reads (rows 1 and 4 are split between two waves) instead of the 32 an output
        // row would need (the exchange rounds are LDS-bandwidth-bound: 8 KiB written + 8 KiB read per wave and round).
        // rows of Z: 0 = q0.r0 | 1 = q0.r1 + q1.r0 | 2 = q1.r1 | 3 = q2.r0 | 4 = q2.r1 + q3.r0 | 5 = q3.r1
        const int w0 = grp * 4;
        constexpr int j = Q;
        auto ld = [&](int q, int r) {
          const float4 z = xch[(((w0 + q) * 2 + r) * 4 + j) * 64 + lane];
          return (f32x4){z.x, z.y, z.z, z.w};
        };
        f32x4 zr[6];
        zr[0] = ld(0, 0); zr[1] = ld(0, 1) + ld(1, 0); zr[2] = ld(1, 1);
        zr[3] = ld(2, 0); zr[4] = ld(2, 1) + ld(3, 0); zr[5] = ld(3, 1);
        const int g4 = g * 4;
        const float4 sh = *reinterpret_cast<const float4*>(p.bias + (nt0 + n) * 16 + g4);
        const float lo = p.act == 1 ? 0.f : -INFINITY;        // ReLU as a clamp: no branch in the store loop
        const bool has_res = p.res != nullptr;
        const int ox = oxb + j;
        int xo, ximg = 0;
        bool okx;
        size_t ooff[4];
        float4 rr[4];
        bool oky[4];
        if constexpr (FLAT == 2) {
          // mosaic: the tile's virtual rows / column -> (image, pixel); a tile may straddle two images and the border line between them
          int xc;
          mosaic_split((uint32_t)ox, p.Wp1, p.dWp1, &ximg, &xc);
          okx = tl.valid && xc < p.W && ximg < p.MS;
          xo = min(xc, p.W - 1) * 16;
        } else {
          xo = min(ox, p.W - 1) * 16;
          okx = tl.valid && ox < p.W;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {                         // per-row output offsets (clamped: dead pixels compute harmlessly, masked at the store)
          size_t orow;
          if constexpr (FLAT == 2) {
            int yimg, yc;
            mosaic_split((uint32_t)(oyb + i), p.Hp1, p.dHp1, &yimg, &yc);
            const int img = (tl.b * p.MS + yimg) * p.MS + ximg;
            oky[i] = yc < p.H && yimg < p.MS && img < p.B;
            orow = oky[i] && okx ? (size_t)img * p.H + yc : 0;
          } else {
            oky[i] = oyb + i < p.H;
            orow = (size_t)tl.b * p.H + min(oyb + i, p.H - 1);
          }
          ooff[i] = orow * p.out_rs + (size_t)(nt0 + n) * p.out_ss + g4 + xo;
          rr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (has_res) rr[i] = *reinterpret_cast<const float4*>(p.res + orow * p.res_rs + (size_t)(nt0 + n) * p.out_ss + g4 + xo);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          f32x4 v = (f32x4){sh.x, sh.y, sh.z, sh.w};
#pragma unroll
          for (int k = 0; k < 6; ++k)
            if (at_c(i, k) != 0.f) v += at_c(i, k) * zr[k];
          const float4 r = rr[i];
          if (!p.res_after_act) { v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w; }
          v[0] = fmaxf(v[0], lo); v[1] = fmaxf(v[1], lo); v[2] = fmaxf(v[2], lo); v[3] = fmaxf(v[3], lo);
          if (p.res_after_act) { v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w; }
          // (streaming stores - __builtin_nontemporal_store - make a launch 1.5 us faster alone and the whole forward 0.5 % slower:
          // the next conv finds less of its input in the L2.  W4P_EXP & 2048: probe without the stores.)
          if (okx && oky[i] && !((W4P_EXP & 2048) && p.B < 100000))
            *reinterpret_cast<float4*>(p.out + ooff[i]) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
      W4P_T(e6);
      W4P_ACC(11, e3, e6);
    }
#endif
    __syncthreads();                                        // the exchange area is free again (next item's rings / V)
#if W4P_TRACE
    W4P_T(c_e1);
    W4P_ACC(3, c_e0, c_e1);
    if (trace && lane == 0) {
      if (wave == 0) { for (int k = 0; k < 4; ++k) g_w4p_trace[k] = tr[k]; g_w4p_trace[4] = (unsigned long long)p.nC4; }
      g_w4p_trace[16 + 2 * wave] = tr[1]; g_w4p_trace[17 + 2 * wave] = tr[2]; g_w4p_trace[48 + wave] = tr[4];
      if (wave == 0 || wave == 5) for (int k = 0; k < 6; ++k) g_w4p_trace[(wave == 0 ? 56 : 64) + k] = tr[8 + k];
    }
#endif
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// producer waves: LDS-DMA of the raw patch + U fragments, input transform V = B^T d B for rows 3 RH .. 3 RH + 2
// ---------------------------------------------------------------------------------------------------------------------
template <int NT, int RH, int FLAT>
__device__ __forceinline__ void w4p_producer(const W4PParams& p, float4* smem, int pw, int lane) {
  const int grp = pw >> 1;
  const int idx = lane >> 2, g = lane & 3;   // transform lane order: 8 tiles x 4 channels per 32-lane half (see w4p_sigma)
  const int vlane = w4p_sigma(idx, g);
  const int rawF4 = p.rawF4;
  const int uF4 = NT * W4P_UBLK;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float4*)smem;
  const int npieces_raw = rawF4 >> 6;
  const Walk wk = item_walk(p);
  // First fetches of an item: raw(0..2) and U(0) (U(1) follows once the exchange area, which overlays its slot, is free).
  // The producers issue them for the NEXT item while the MFMA waves run the exchange rounds of the current one, and zero
  // the padding positions of the item's patch in all ring slots (every later DMA into the ring skips those lanes).
  auto prefetch_item = [&](int item) __attribute__((always_inline)) {
    constexpr int MAXP = 4;                                 // raw pieces pw, pw + 4, ... (rawF4 <= 1024 slots)
    int goff[MAXP];
    const int id = W4P_NINNER ? w4p_item_id(p, item) : item;
    raw_piece_offsets<MAXP, FLAT>(p, id, pw, W4P_NPROD, lane, goff);
    const int nt0 = (id / p.nblocks_m) * NT;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
      const int piece = pw + W4P_NPROD * k;
      if (piece < npieces_raw) {
        if (goff[k] < 0) {
#pragma unroll
          for (int slot = 0; slot < 3; ++slot) smem[slot * rawF4 + piece * 64 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
#pragma unroll
          for (int c4 = 0; c4 < 3; ++c4)
            if (c4 < p.nC4)
              w4::dma16_sv(p.in + (size_t)(c4 >> 2) * p.in_ss + (c4 & 3) * 4, (unsigned)goff[k] * 4u,
                           (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + (unsigned)(c4 * rawF4 + piece * 64) * 16u)));
        }
      }
    }
#pragma unroll
    for (int i0 = 0; i0 < 9 * NT; i0 += W4P_NPROD) {
      const int i = i0 + pw;
      if (i < 9 * NT) {
        const int n = i / 9, j = i - n * 9;
        const float4* src = p.ufrag + (((size_t)0 * p.nT16 + min(nt0 + n, p.nT16 - 1)) * 9 + j) * 64;
        w4::dma16_sv(src, (unsigned)lane * 16u, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + (unsigned)(p.uoff + i * 64) * 16u)));
        if ((W4P_EPI & 1) && p.nC4 > 1)                      // U(1) -> ring slot 1 (free during the exchange rounds in the round-5 layout)
          w4::dma16_sv(src + (size_t)p.nT16 * 9 * 64, (unsigned)lane * 16u,
                       (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + (unsigned)(p.uoff + uF4 + i * 64) * 16u)));
      }
    }
  };
  // the same fetches as a list of jobs - [zero the padding slots | raw(0..2): 3 x 4 pieces | U(0), U(1): 2 x ceil(9 NT / 4) pieces] -
  // of which part `part` of `nparts` is issued (both compile-time constants after unrolling)
  auto prefetch_part = [&](int item, const int (&goff)[4], int part, int nparts) __attribute__((always_inline)) {
    constexpr int NU = (9 * NT + W4P_NPROD - 1) / W4P_NPROD;
    constexpr int NJ = 12 + 2 * NU;
    const int lo = part * NJ / nparts, hi = (part + 1) * NJ / nparts;
    const int nt0 = ((W4P_NINNER ? w4p_item_id(p, item) : item) / p.nblocks_m) * NT;
    if (part == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int piece = pw + W4P_NPROD * k;
        if (piece < npieces_raw && goff[k] < 0) {
#pragma unroll
          for (int slot = 0; slot < 3; ++slot) smem[slot * rawF4 + piece * 64 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (j < lo || j >= hi) continue;
      if (j < 12) {
        const int c4 = j >> 2, k = j & 3;
        const int piece = pw + W4P_NPROD * k;
        if (piece < npieces_raw && goff[k] >= 0 && c4 < p.nC4)
          w4::dma16_sv(p.in + (size_t)(c4 >> 2) * p.in_ss + (c4 & 3) * 4, (unsigned)goff[k] * 4u,
                       (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + (unsigned)(c4 * rawF4 + piece * 64) * 16u)));
      } else {
        const int c4 = (j - 12) / NU, i = pw + W4P_NPROD * ((j - 12) % NU);
        if (i < 9 * NT && c4 < p.nC4 && c4 < ((W4P_EPI & 1) ? 2 : 1)) {       // (U(1) only where its slot is outside the exchange area)
          const int n = i / 9, jj = i - n * 9;
          const float4* src = p.ufrag + (((size_t)c4 * p.nT16 + min(nt0 + n, p.nT16 - 1)) * 9 + jj) * 64;
          w4::dma16_sv(src, (unsigned)lane * 16u, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + (unsigned)(p.uoff + c4 * uF4 + i * 64) * 16u)));
        }
      }
    }
  };
  auto issue_u1 = [&](int item) __attribute__((always_inline)) {       // U(1) -> ring slot 1
    const int nt0 = ((W4P_NINNER ? w4p_item_id(p, item) : item) / p.nblocks_m) * NT;
#pragma unroll
    for (int i0 = 0; i0 < 9 * NT; i0 += W4P_NPROD) {
      const int i = i0 + pw;
      if (i < 9 * NT) {
        const int n = i / 9, j = i - n * 9;
        const float4* src = p.ufrag + (((size_t)1 * p.nT16 + min(nt0 + n, p.nT16 - 1)) * 9 + j) * 64;
        w4::dma16_sv(src, (unsigned)lane * 16u, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + (unsigned)(p.uoff + uF4 + i * 64) * 16u)));
      }
    }
  };
  if (wk.first < wk.end) prefetch_item(wk.first);
  // fp32 MFMAs run on the SIMD's vector ALUs: the two MFMA waves of this SIMD always have one ready and would starve this
  // wave's VALU / LDS / VMEM instructions until they reach the slice barrier (measured: MFMA time + producer time add up).
  // With a higher issue priority the producer's instructions slot in between their MFMAs and its stalls cost nothing.
  if (!(W4P_EXP & 32)) __builtin_amdgcn_s_setprio(3);

  for (int item = wk.first; item < wk.end; item += wk.step) {
    const int id = W4P_NINNER ? w4p_item_id(p, item) : item;
    const int nt0 = (id / p.nblocks_m) * NT;
    // ---- this lane's (tile, channel) pair: float offsets of its 36 window elements in a raw slot -----------------------
    const Tile tl = tile_of<FLAT>(p, id, grp, idx);
    // pos(k, 0) is even (patch width and tile origins are even), so the columns (2c, 2c + 1) of a window row never straddle a
    // multiple of 8 in the skewed slot order: their slots are neighbours (16 B apart) and ONE ds_read2_b32 fetches the pair
    // into a 64-bit register pair = one operand of the packed-fp32 transform below.  18 addresses / reads instead of 36.
    int woff[6][3];
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int pos = tl.base + k * p.PW + 2 * c;
        woff[k][c] = (pos + (pos >> (FLAT ? 4 : 3))) * 4 + g;
      }
    f32x2 d[6][3];                                            // the window of the slice that is transformed next (column pairs)
    auto load_window = [&](int rslot) {
      const float* rawf = reinterpret_cast<const float*>(smem + rslot * rawF4);
#pragma unroll
      for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float* w = rawf + woff[k][c];
          d[k][c] = (f32x2){w[0], w[4]};
        }
    };
    // V = B^T d B in packed fp32 (v_pk_fma_f32 / v_pk_add_f32: half the issue slots of scalar VALU, and fp32 MFMAs share the
    // vector ALUs with it).  Stage 1 (down the window columns) works on the column pairs as they were read; stage 2 (along a
    // row) produces the pairs (nu0, nu5), (nu1, nu3), (nu2, nu4): 18 + 21 packed instructions for this wave's 18 positions.
    auto transform = [&](int vbuf) {                          // d -> rows 3 RH .. 3 RH + 2 of V (18 positions) of group grp
      f32x2 t[3][3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const f32x2 d0 = d[0][c], d1 = d[1][c], d2 = d[2][c], d3 = d[3][c], d4 = d[4][c], d5 = d[5][c];
        if constexpr (RH == 0) {
          t[0][c] = pk_fma(d0, 4.f, pk_fma(d2, -5.f, d4));                       // 4 d0 - 5 d2 + d4
          const f32x2 a = pk_fma(d2, -4.f, d4), cc = pk_fma(d1, 4.f, -d3);       // rows 1, 2 = (d4 - 4 d2) -+ (4 d1 - d3)
          t[1][c] = a - cc;
          t[2][c] = a + cc;
        } else {
          const f32x2 b = d4 - d2, e = d1 - d3;                                  // rows 3, 4 = (d4 - d2) -+ 2 (d1 - d3)
          t[0][c] = pk_fma(e, -2.f, b);
          t[1][c] = pk_fma(e, 2.f, b);
          t[2][c] = pk_fma(d1, 4.f, pk_fma(d3, -5.f, d5));                       // 4 d1 - 5 d3 + d5
        }
      }
      f32x2 A[3], Bp[3], Cp[3];                               // per row: (v0, v5), (v1, v3), (v2, v4)
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const f32x2 T0 = t[r][0], T1 = t[r][1], T2 = t[r][2];
        A[r] = pk_fma(T0, 4.f, pk_fma(T1, -5.f, T2));                            // 4 t0 - 5 t2 + t4 | 4 t1 - 5 t3 + t5
        const f32x2 ab = pk_fma2(T1.xx, (f32x2){-4.f, -1.f}, T2.xx);             // t4 - 4 t2 | t4 - t2
        const f32x2 cf = pk_fma2(T0.yy, (f32x2){4.f, 2.f}, T1.yy * (f32x2){-1.f, -2.f});   // 4 t1 - t3 | 2 t1 - 2 t3
        Bp[r] = ab - cf;
        Cp[r] = ab + cf;
      }
      // slot order of the MFMA waves q = 2 RH and 2 RH + 1: see w4p_nu
      float4* Vg = smem + p.voff + vbuf * (2 * W4P_UBLK) + grp * W4P_UBLK;
      float* Vs = reinterpret_cast<float*>(Vg + 512);
      constexpr int qa = 2 * RH, qb = 2 * RH + 1;
      Vg[(2 * qa) * 64 + vlane] = make_float4(A[0].x, A[0].y, Bp[0].x, Bp[0].y);
      Vg[(2 * qa + 1) * 64 + vlane] = make_float4(Cp[0].x, Cp[0].y, A[1].x, A[1].y);
      Vs[qa * 64 + vlane] = Bp[1].x;
      Vs[qb * 64 + vlane] = Bp[1].y;
      Vg[(2 * qb) * 64 + vlane] = make_float4(Cp[1].x, Cp[1].y, A[2].x, A[2].y);
      Vg[(2 * qb + 1) * 64 + vlane] = make_float4(Bp[2].x, Bp[2].y, Cp[2].x, Cp[2].y);
    };

    const int S = p.nC4;
    if (!(W4P_EPI & 1) && S > 1) issue_u1(item);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                        // B0: raw(0..2), U(0..1) have landed
    load_window(0);
    transform(0);
    if (S > 1) load_window(1);
    __syncthreads();                                        // B1 (the compiler waits for this wave's LDS accesses before a barrier)
    int ring = 0;                                           // s % 3
#if W4P_TRACE
    const bool trace = blockIdx.x == 0 && item == wk.first + W4P_TRACE_ITEM * wk.step;
    unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (int s = 0; s < S; ++s) {
      const int r1 = ring == 2 ? 0 : ring + 1, r2 = r1 == 2 ? 0 : r1 + 1;
      W4P_T(q0);
      // V(s+1) from the window fetched during the previous iteration, then the window of slice s+2 (raw(s+2) landed before the
      // barrier that ended iteration s-1)
      if (s + 1 < S && !(W4P_EXP & 17)) transform((s + 1) & 1);
      W4P_T(q1);
      if (s + 2 < S && !(W4P_EXP & (17 | 128))) load_window(r2);
      W4P_T(q4);
      ring = r1;
      __syncthreads();
      W4P_T(q6);
      W4P_ACC(0, q0, q1); W4P_ACC(3, q1, q4); W4P_ACC(5, q4, q6);
    }
#if W4P_TRACE
    if (trace && lane == 0) {
      if (pw == 0) for (int k = 0; k < 6; ++k) g_w4p_trace[8 + k] = tr[k];
      g_w4p_trace[32 + 4 * pw] = tr[0]; g_w4p_trace[33 + 4 * pw] = tr[3]; g_w4p_trace[34 + 4 * pw] = tr[5];
    }
#endif
    const bool more = item + wk.step < wk.end;
    if constexpr ((W4P_EPI & 4) != 0) {
      // the next item's first fetches, dealt over the 2 NT gaps between the barriers of the exchange rounds (no barrier in front of
      // round 0): ~5 LDS-DMA instructions (~60 clk each) per gap, so that this wave is never the last one at a barrier
      int goffn[4];
      if (more) raw_piece_offsets<4, FLAT>(p, W4P_NINNER ? w4p_item_id(p, item + wk.step) : item + wk.step, pw, W4P_NPROD, lane, goffn);
#pragma unroll
      for (int gap = 0; gap < 2 * NT; ++gap) {
        if (gap > 0) __syncthreads();
        if (more) prefetch_part(item + wk.step, goffn, gap, 2 * NT);
      }
    } else {
      if (more) prefetch_item(item + wk.step);                           // ... while the MFMA waves exchange and store
#pragma unroll
      for (int n = 0; n < NT; ++n) { __syncthreads(); __syncthreads(); }   // the MFMA waves' exchange rounds
    }
    __syncthreads();
  }
}

template <int NT, int FLAT>      // FLAT: 0 rectangular items, 1 flat items, 2 flat items over a mosaic of images
__global__ void __launch_bounds__(768)
conv_wino4p_kernel(const W4PParams p) {
  extern __shared__ float4 smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave < W4P_NCONS) {
    const int grp = wave >> 2, q = wave & 3;
    if (q == 0) w4p_consumer<NT, 0, FLAT>(p, smem, grp, lane, wave);
    else if (q == 1) w4p_consumer<NT, 1, FLAT>(p, smem, grp, lane, wave);
    else if (q == 2) w4p_consumer<NT, 2, FLAT>(p, smem, grp, lane, wave);
    else w4p_consumer<NT, 3, FLAT>(p, smem, grp, lane, wave);
  } else {
    const int pw = wave - W4P_NCONS;
    if (pw & 1) w4p_producer<NT, 1, FLAT>(p, smem, pw, lane);
    else w4p_producer<NT, 0, FLAT>(p, smem, pw, lane);
  }
}

struct W4PLayout { int uoff, voff, xoff, totalF4; };
bool w4p_geo(const ConvDesc& d, const ConvCfg& cfg, w4::Geo* g, W4PLayout* L, FlatGeo* fg = nullptr) {
  if (d.ks != 3 || d.stride != 1 || cfg.NT < 1 || cfg.NT > 3 || cfg.WM != 2 || cfg.WN != 4 || d.Cin % 16 || d.Cout % 16) return false;
  FlatGeo ftmp;
  if (cfg.NI == 0) { if (!flat_geo(d, cfg, g, fg ? fg : &ftmp)) return false; }
  else if (!w4::geo(d, cfg, 32, g)) return false;
  if (g->rawF4 > W4P_NCONS * W4P_MAXP * 64) return false;
  const int uF4 = cfg.NT * W4P_UBLK, vF4 = 2 * W4P_UBLK;
  L->uoff = 3 * g->rawF4;
  L->voff = L->uoff + 3 * uF4;
  const int end = L->voff + 2 * vF4;
  // Round 4: the exchange area overlaid U ring slots 1, 2 and the V buffers (slot 0 and the raw ring received the next item's
  // first fetches during the exchange rounds), so U(1) could only be requested at the item start and its whole round trip was
  // exposed there.  Round 5 (W4P_EPI & 1): the area starts behind U slot 1 - at NT = 3 with flat items U slot 2 + V + the spare KiB
  // are exactly 64 KiB and the block uses exactly 160 KiB - so both U(0) and U(1) travel during the exchange rounds.
  L->xoff = L->uoff + ((W4P_EPI & 1) ? 2 : 1) * uF4;
  L->totalF4 = std::max(end, L->xoff + W4P_XCH);
  return (size_t)L->totalF4 * sizeof(float4) <= 160 * 1024;
}

}  // namespace

// packed fragments for ALG 8: [Cin/4][Cout16/16][9 pieces][64 lanes] float4; piece 2q+a (q = 0..3, a = 0..1): lane = g*16 + co_l holds
// U[pos(q, 4a + j)][co][4 c4 + g] * scale[co], j = 0..3; piece 8: float index q*64 + lane = U[pos(q, 8)][co][4 c4 + g] * scale[co];
// pos(q, i) = 6 (RA(q) + w4p_row(q, i)) + w4p_nu(q, i) (the slot order of the packed input transform, see w4p_nu)
size_t conv_wino4p_packed_floats(int Cin, int Cout16) { return (size_t)36 * Cin * Cout16 + (size_t)2 * 9 * 256; }     // + 2 n-tile blocks of slack (streamed U requests)
void conv_wino4p_pack_weights(const float* w_oihw, const float* scale, int Cout, int Cin, int Cout16, float* dst) {
  std::vector<double> u;
  w4::u_transform(w_oihw, Cout, Cin, &u);
  const int nC4 = Cin / 4, nT16 = Cout16 / 16;
  auto val = [&](int pos, int co, int ci) -> float {
    return co < Cout ? (float)(u[((size_t)pos * Cout + co) * Cin + ci] * (scale ? (double)scale[co] : 1.0)) : 0.f;
  };
  std::fill(dst + (size_t)36 * Cin * Cout16, dst + conv_wino4p_packed_floats(Cin, Cout16), 0.f);
  for (int c4 = 0; c4 < nC4; ++c4)
    for (int nt = 0; nt < nT16; ++nt) {
      float* blk = dst + ((size_t)c4 * nT16 + nt) * 9 * 256;
      for (int lane = 0; lane < 64; ++lane) {
        const int g = lane >> 4, co = nt * 16 + (lane & 15), ci = 4 * c4 + g;
        for (int q = 0; q < 4; ++q) {
          auto pos = [&](int i) { return 6 * (w4p_ra(q) + w4p_row(q, i)) + w4p_nu(q, i); };
          for (int a = 0; a < 2; ++a)
            for (int j = 0; j < 4; ++j) blk[((2 * q + a) * 64 + lane) * 4 + j] = val(pos(4 * a + j), co, ci);
          blk[8 * 256 + q * 64 + lane] = val(pos(8), co, ci);
        }
      }
    }
}

// cfg: {MT = 1, NT (1..3), WM = 2 tile groups, WN = 4 position quarters, R = output rows per slab (multiple of 4), NI, ALG = 8}
size_t conv_wino4p_lds_bytes(const ConvDesc& d, const ConvCfg& cfg) {
  w4::Geo g;
  W4PLayout L;
  if (!w4p_geo(d, cfg, &g, &L)) return 0;
  return (size_t)L.totalF4 * sizeof(float4);
}

int conv_wino4p_launch(const ConvDesc& d, const ConvCfg& cfg, hipStream_t stream) {
  w4::Geo g;
  W4PLayout L;
  FlatGeo fg{};
  const bool flat = cfg.NI == 0;
  if (!w4p_geo(d, cfg, &g, &L, &fg) || !d.wfrag_wino4p) {
    poco_set_error("conv(winograd 4x4, specialised waves): needs ks = 3, stride 1, NT 1..3, WM = 2, WN = 4, R % 4 == 0, "
                   "NI*(R/4)*ceil(W/4) <= 32 tiles (or R = 4, NI = 0: flat items), a patch of <= 1024 slots that fits the LDS next "
                   "to the U ring, and the ALG 8 weight fragments");
    return POCO_ERR_ARG;
  }
  if (d.act == 3 || d.act == 2) { poco_set_error("conv(winograd 4x4): activation must be none or ReLU"); return POCO_ERR_ARG; }
  W4PParams p{};
  p.in = d.in + l16_chan_off(d.in_co, d.W);
  p.res = d.res ? d.res + l16_chan_off(d.res_co, d.W) : nullptr;
  p.out = d.out + l16_chan_off(d.out_co, d.W);
  p.ufrag = reinterpret_cast<const float4*>(d.wfrag_wino4p); p.bias = d.bias;
  p.B = d.B; p.H = d.H; p.W = d.W; p.nC4 = d.Cin / 4; p.nT16 = d.Cout / 16;
  p.in_rs = d.in_cs * d.W; p.in_ss = d.W * 16; p.res_rs = d.res_cs * d.W; p.out_rs = d.out_cs * d.W; p.out_ss = d.W * 16;
  p.R = g.R; p.NI = g.NI; p.S = g.S; p.nbands = g.nbands; p.TX = g.TX; p.PR = g.PR; p.PW = g.PW; p.npos = g.npos; p.rawF4 = g.rawF4;
  p.tiles_per_slab = g.tps;
  p.act = d.act; p.res_after_act = d.res_after_act;
  p.uoff = L.uoff; p.voff = L.voff; p.xoff = L.xoff;
  p.dPW = make_fastdiv(g.PW); p.dSlab = make_fastdiv(g.PR * g.PW); p.dBands = make_fastdiv(g.nbands);
  p.dTX = make_fastdiv(g.TX); p.dTslab = make_fastdiv(g.tps);
  p.nblocks_m = flat ? g.S : (g.S + g.NI - 1) / g.NI; p.nb_n = (p.nT16 + cfg.NT - 1) / cfg.NT;
  p.dNbn = make_fastdiv(p.nb_n);
  // measured (tools/ninner_solo.py, tools/conv_traffic.py): 480 -> 128 @ 56x56 -4 % and 715 -> 479 MB per launch; the short-K shapes
  // (14x14 192 -> 192, 28x28 96 -> 96) +1 ... +2 % - their patch is small next to the L2 and the extra division sits on the item start
  p.ninner = p.nb_n > 1 && d.Cin * p.nb_n >= 1024;
  if (flat) {
    p.TY = fg.TY; p.ntiles = fg.ntiles; p.fragW = fg.fragW;
    p.dTY = make_fastdiv(fg.TY); p.dFragW = make_fastdiv(fg.fragW);
    p.MS = fg.MS; p.Hp1 = d.H + 1; p.Wp1 = d.W + 1;
    p.dHp1 = make_fastdiv(d.H + 1); p.dWp1 = make_fastdiv(d.W + 1);
  }
  // balanced persistent grid: every block walks the same number of items (one block per CU)
  // cfg.MT = CU share divisor: the grid is sized for (CUs of the device) / MT.  A block needs a whole CU (LDS), all blocks of a launch run
  // their K loops (MFMA-bound, HBM nearly idle) and their store phases (HBM-write-bound, MFMA idle) in lockstep; two launches
  // of different lanes on half of the CUs each run out of phase and overlap one's stores with the other's MFMAs.
  int mt = std::max(1, cfg.MT);
#if W4P_EXP
  static const int mt_env = [] { const char* e = getenv("POCO_W4P_MT"); return e ? atoi(e) : 0; }();      // probe builds only
  if (mt_env > 0) mt = mt_env;
#endif
  const long cus = std::max(8, poco_num_cus() / mt);
  const long items = (long)p.nblocks_m * p.nb_n;
  const long rounds = (items + cus - 1) / cus;
  long g4 = (items + rounds - 1) / rounds;
  if (g4 > 8) g4 = std::min(cus, (g4 + 7) / 8 * 8);            // multiple of 8 for the XCD-aware walk
  const size_t lds = (size_t)L.totalF4 * sizeof(float4);
  const int mode = !flat ? 0 : fg.MS > 1 ? 2 : 1;
  void (*fn)(const W4PParams) =
      mode == 2 ? (cfg.NT == 3 ? conv_wino4p_kernel<3, 2> : cfg.NT == 2 ? conv_wino4p_kernel<2, 2> : conv_wino4p_kernel<1, 2>)
      : mode == 1 ? (cfg.NT == 3 ? conv_wino4p_kernel<3, 1> : cfg.NT == 2 ? conv_wino4p_kernel<2, 1> : conv_wino4p_kernel<1, 1>)
                  : (cfg.NT == 3 ? conv_wino4p_kernel<3, 0> : cfg.NT == 2 ? conv_wino4p_kernel<2, 0> : conv_wino4p_kernel<1, 0>);
  if (lds > 64 * 1024) {
    static thread_local bool configured[12] = {};
    if (!configured[cfg.NT + 4 * mode]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) { poco_set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return POCO_ERR_HIP; }
      configured[cfg.NT + 4 * mode] = true;
    }
  }
  hipLaunchKernelGGL(fn, dim3((unsigned)g4, 1), dim3(768), lds, stream, p);
  POCO_HIP_CHECK(hipGetLastError());
  return POCO_OK;
}

#if W4P_TRACE
extern "C" int poco_w4p_trace(unsigned long long* host_out, int n) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_w4p_trace), sizeof(unsigned long long) * (size_t)std::min(n, 96)) == hipSuccess ? 0 : 1;
}
#endif
