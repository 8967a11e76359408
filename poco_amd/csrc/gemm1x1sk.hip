// 1x1 stride-1 convolutions as a STREAM-K register-direct GEMM - ALG 14 (round 5).
//
// The 1x1 convs of ResNet-50's layer2-4 / the HRNet cls head (resnet.py:101-121, hrnet_cls.py:306-353) are GEMMs of 6.6 GFLOP whose
// output is a few thousand wave tiles: 14x14 1024->256 at 64 crops is 896 tiles of 112 x 32 for 1024 SIMDs, 28x28 128->512 3136 tiles
// of 64 x 64 (3.06 per SIMD), 7x7 2048->512 784 tiles.  ALG 6 / 9 give every wave whole tiles, so a launch lasts ceil(tiles / SIMDs)
// tile times: 12 ... 24 % of the chip idles in the last round ("wave-tile quantisation").  Here the unit of work is
// (tile, 16-channel K slice): the launch is a persistent grid of W waves, wave w owns the units [U w / W, U (w + 1) / W) of the
// tile-major list - at most one tile's tail, whole tiles, one tile's head - and every SIMD gets the same number of MFMAs.
//
//   * the K loop of a segment is ALG 6's (gemm1x1.hip): no LDS, no barrier, operands global -> VGPR with coalesced dwordx4, D - 1
//     slices in flight; waves are independent of their block;
//   * a segment that does not reach its tile's last slice stores its accumulators as a PARTIAL in the wave's scratch slot and raises
//     the wave's flag; the wave whose segment ends the tile adds the partials of the waves before it (fixed order: w - 1, w - 2, ...)
//     to its own and runs the epilogue (shift, residual, activation).  A wave finishes its whole tiles and writes its partial FIRST
//     and only then waits for others: nobody waits for a wave that is itself waiting, and the waited-for waves have lower block
//     indices (dispatched earlier).  The wait is bounded (error word as in mlp_chain.hip);
//   * partials and flags travel between XCDs with agent-scope accesses (global_load / store ... sc1), no cache maintenance;
//   * results are deterministic (the split depends only on the shape and the configuration), not bitwise those of ALG 6 (the
//     partial sums are added in a different order).
#include "conv_mfma_types.h"
#include <algorithm>
#include <string>

namespace {

struct SKParams {
  const float* in;
  const float* res;
  float* out;
  const float4* wfrag;   // [Cin/16][Cout16/16][64] float4
  const float* bias;
  int P, W;              // pixels, plane width
  int nC16, nT16, ntn, ntiles, nwaves, ngroups;
  int in_rs, in_ss, res_rs, out_rs, out_ss;
  int act, res_after_act, relu_from;
  FastDiv dW;
  float* part;           // [nwaves][MT * NT][64] float4
  unsigned* flag;        // [nwaves], zero between launches
  unsigned* err_host;    // pinned host word: a wait timed out
  unsigned max_spins;    // poll bound of a wait
};

__device__ __forceinline__ float4 sk_ld4(const float* p) {
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
  const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_float4(__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)),
                     __uint_as_float((unsigned)hi), __uint_as_float((unsigned)(hi >> 32)));
}
__device__ __forceinline__ void sk_st4(float* p, f32x4 v) {
  unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
  __hip_atomic_store(q, (unsigned long long)__float_as_uint(v[0]) | ((unsigned long long)__float_as_uint(v[1]) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(q + 1, (unsigned long long)__float_as_uint(v[2]) | ((unsigned long long)__float_as_uint(v[3]) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int MT, int NT, int D, int SCHED>
__global__ void __launch_bounds__(MT * NT > 16 ? 256 : 512)     // 7 x 4 tiles: 112 accumulators + two operand stages need the 512-register budget
gemm1x1sk_kernel(const SKParams p) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int idx = lane & 15, g = lane >> 4;
  const int wid = blockIdx.x * (blockDim.x >> 6) + wave;
  // The waves are dealt to the n-tile groups round-robin (wave w -> group w % ng, rank w / ng) and every group streams over ITS tiles
  // (all pixel tiles of one n-tile group): waves w ... w + ng - 1 of a block then read the same pixels at the same time, as the
  // wave columns of an ALG 6 block do.  With one tile-major list for all waves the ng reads of a pixel slice are far apart in time
  // and each comes from memory: 263 MB per launch instead of ~70 (PMC, ResNet-50's four ALG 14 shapes).
  const int ng = p.ngroups, group = wid % ng, rank = wid / ng, Wg = p.nwaves / ng;
  const long long U = (long long)(p.ntiles / ng) * p.nC16;
  const long long lo = U * rank / Wg, hi = U * (rank + 1) / Wg;
  if (lo >= hi) return;
  const int t_first = (int)(lo / p.nC16), c_first = (int)(lo - (long long)t_first * p.nC16);
  const int t_last = (int)((hi - 1) / p.nC16), c_last_end = (int)(hi - 1 - (long long)t_last * p.nC16) + 1;
  const bool finisher = c_first > 0 && (t_first < t_last || c_last_end == p.nC16);   // the first segment ends its tile

  int boff[MT], orow[MT], ox16[MT];
  f32x4 acc[MT][NT];
  const int wslice = p.nT16 * 64;                        // float4 per K slice
  // tile -> pixel offsets of this lane; K loop over slices [cb, ce) into acc
  auto segment = [&](int tl, int cb, int ce) {
    const int t = ng > 1 ? tl * p.ntn + group : tl;      // ng > 1: local tile tl = pixel tile tl of n-tile group `group`
    const int tm = t / p.ntn, tn = t - tm * p.ntn;
    const int mt0 = tm * MT, nt0 = tn * NT;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int pix = (mt0 + m) * 16 + idx;
      const uint32_t pc = (uint32_t)min(pix, p.P - 1);   // dead lanes re-read the last pixel
      const uint32_t row = fdiv(pc, p.dW);
      const uint32_t x = pc - row * p.W;
      boff[m] = (int)(row * (uint32_t)p.in_rs + x * 16u) + 4 * g;
      orow[m] = pix < p.P ? (int)row : -1;
      ox16[m] = (int)x * 16;
    }
    const float4* wl = p.wfrag + (size_t)nt0 * 64 + lane;
    int woff[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) woff[n] = (nt0 + n < p.nT16) ? n * 64 : 0;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 a[D][NT], b[D][MT];
    // one operand load of slice c into stage s: pieces 0..NT-1 = weight fragments, NT..NT+MT-1 = pixel sub-tiles
    auto load_piece = [&](int s, int c, int i) {
      if (i < NT) a[s][i] = wl[(size_t)c * wslice + woff[i]];
      else b[s][i - NT] = *reinterpret_cast<const float4*>(p.in + boff[i - NT] + (size_t)c * p.in_ss);
    };
    auto load = [&](int s, int c) {
#pragma unroll
      for (int i = 0; i < NT + MT; ++i) load_piece(s, c, i);
    };
    // SCHED as in gemm1x1.hip: 0 = hipcc's order; else one operand load of the slice D - 1 ahead per G = SCHED & 15 MFMAs, from the
    // start of the slice or (SCHED & 16) ending with it
    constexpr int G = SCHED & 15;
    constexpr int K0 = (SCHED & 16) ? 4 * MT * NT - G * (NT + MT) : 0;
    auto mma = [&](int s, bool ld, int sn, int cn) {
      int k = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const float wj = (j == 0) ? a[s][n].x : (j == 1) ? a[s][n].y : (j == 2) ? a[s][n].z : a[s][n].w;
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            if constexpr (SCHED > 0) {
              const int kk = k - K0;
              if (ld && kk >= 0 && kk % G == 0 && kk / G < NT + MT) {
                load_piece(sn, cn, kk / G);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
            const float bj = (j == 0) ? b[s][m].x : (j == 1) ? b[s][m].y : (j == 2) ? b[s][m].z : b[s][m].w;
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wj, bj, acc[m][n], 0, 0, 0);
            if constexpr (SCHED > 0) {
              if (ld && k >= K0 - 1 && k < K0 + G * (NT + MT)) __builtin_amdgcn_sched_barrier(0);
            }
            ++k;
          }
        }
    };
    const int last = ce - 1, n = ce - cb;
#pragma unroll
    for (int s = 0; s < D - 1; ++s) load(s, min(cb + s, last));
    const int nfull = n / D * D;
    for (int i0 = 0; i0 < nfull; i0 += D) {
#pragma unroll
      for (int u = 0; u < D; ++u) {
        if constexpr (SCHED == 0) load((u + D - 1) % D, min(cb + i0 + u + D - 1, last));
        mma(u, true, (u + D - 1) % D, min(cb + i0 + u + D - 1, last));
      }
    }
#pragma unroll
    for (int u = 0; u < D - 1; ++u)
      if (nfull + u < n) mma(u, false, 0, 0);
  };
  // shift (+ residual) (activation) -> L16 channel slice, as in gemm1x1.hip
  auto epilogue = [&](int tl) {
    const int t = ng > 1 ? tl * p.ntn + group : tl;
    const int tm = t / p.ntn, tn = t - tm * p.ntn;
    const int nt0 = tn * NT;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      if (nt0 + n >= p.nT16) continue;
      const int co = (nt0 + n) * 16 + g * 4;
      const float4 sh = *reinterpret_cast<const float4*>(p.bias + co);
      float4 r[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        r[m] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.res) r[m] = *reinterpret_cast<const float4*>(p.res + max(orow[m], 0) * p.res_rs + ox16[m] + g * 4 + (nt0 + n) * p.out_ss);
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        f32x4 v = acc[m][n];
        v[0] += sh.x; v[1] += sh.y; v[2] += sh.z; v[3] += sh.w;
        if (!p.res_after_act) { v[0] += r[m].x; v[1] += r[m].y; v[2] += r[m].z; v[3] += r[m].w; }
        if (p.act == 1 || (p.act == 3 && co >= p.relu_from)) {
          v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
        } else if (p.act == 2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = 1.f / (1.f + __expf(-v[e]));
        }
        if (p.res_after_act) { v[0] += r[m].x; v[1] += r[m].y; v[2] += r[m].z; v[3] += r[m].w; }
        if (orow[m] >= 0)
          *reinterpret_cast<float4*>(p.out + orow[m] * p.out_rs + ox16[m] + g * 4 + (nt0 + n) * p.out_ss) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  };

  // ---- pass 1: whole tiles and the head of the last tile (a partial) ----
  for (int t = finisher ? t_first + 1 : t_first; t <= t_last; ++t) {
    const int cb = t == t_first ? c_first : 0;
    const int ce = t == t_last ? c_last_end : p.nC16;
    segment(t, cb, ce);
    if (cb == 0 && ce == p.nC16) {
      epilogue(t);
    } else {                                             // does not end its tile: store the partial, raise the flag
      float* slot = reinterpret_cast<float*>(p.part) + ((size_t)wid * MT * NT * 64 + lane) * 4;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) sk_st4(slot + (size_t)(m * NT + n) * 256, acc[m][n]);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the (write-through) stores are acknowledged
      if (lane == 0) __hip_atomic_store(&p.flag[wid], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // ---- pass 2: the tail of the first tile: own slices + the partials of the waves before ----
  if (finisher) {
    segment(t_first, c_first, p.nC16);
    const long long tile_lo = (long long)t_first * p.nC16;
    bool ok = true;
    for (int vr = rank - 1; vr >= 0 && U * (vr + 1) / Wg > tile_lo; --vr) {
      if (U * vr / Wg >= U * (vr + 1) / Wg) continue;                  // more waves than units: that wave has no work and no partial
      const int v = vr * ng + group;                                    // global index of the wave with rank vr in this group
      unsigned spins = 0;
      while (__hip_atomic_load(&p.flag[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > p.max_spins) { ok = false; break; }
      }
      if (!ok) break;
      // ordering made explicit: the partial loads below are relaxed agent-scope accesses like the flag poll, so nothing in
      // the language keeps the compiler from hoisting them above the poll - this barrier does (the hardware issues them in program
      // order and the producer's s_waitcnt vmcnt(0) ordered its side; no cache maintenance is wanted: all of it bypasses the L2s)
      asm volatile("" ::: "memory");
      const float* slot = reinterpret_cast<const float*>(p.part) + ((size_t)v * MT * NT * 64 + lane) * 4;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const float4 q = sk_ld4(slot + (size_t)(m * NT + n) * 256);
          acc[m][n][0] += q.x; acc[m][n][1] += q.y; acc[m][n][2] += q.z; acc[m][n][3] += q.w;
        }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the partial is in registers before its flag is lowered
      if (lane == 0) __hip_atomic_store(&p.flag[v], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!ok) {
      if (lane == 0) __hip_atomic_store(p.err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
    epilogue(t_first);
  }
}

bool sk_tile_ok(int MT, int NT, int D) {
  return (MT == 7 && NT == 4 && D == 2) || (MT == 7 && NT == 2 && (D == 2 || D == 3)) || (MT == 4 && NT == 4 && (D == 2 || D == 3)) ||
         (MT == 4 && NT == 2 && D == 3) || (MT == 2 && NT == 4 && D == 3);
}

constexpr int SK_BLOCKS = 256;

}  // namespace

// scratch: [SK_MAX_WAVES flags][partials]; partial floats = waves * MT * NT * 256 <= SK_PART_FLOATS
size_t gemm1x1sk_scratch_floats() { return (size_t)SK_MAX_WAVES + SK_PART_FLOATS; }

// cfg: {MT, NT, WM = waves per block (1..8), WN = 1, R = prefetch depth D (2|3), NI = load schedule 1|3|6 (gemm1x1.hip), ALG = 14}
bool gemm1x1sk_cfg_valid(const ConvDesc& d, const ConvCfg& cfg) {
  const long P = (long)d.B * d.H * d.W;
  return d.ks == 1 && d.stride == 1 && d.Cin % 16 == 0 && d.Cout % 16 == 0 && sk_tile_ok(cfg.MT, cfg.NT, cfg.R) && cfg.WM >= 1 && cfg.WM <= (cfg.MT * cfg.NT > 16 ? 4 : 8) &&
         cfg.WN == 1 && (cfg.NI == 1 || cfg.NI == 3 || cfg.NI == 6) && 4 * (cfg.MT + cfg.NT) <= 4 * cfg.MT * cfg.NT && (size_t)SK_BLOCKS * cfg.WM * cfg.MT * cfg.NT * 256 <= SK_PART_FLOATS && SK_BLOCKS * cfg.WM <= SK_MAX_WAVES && P < (1L << 27) &&
         (long)d.B * d.H * d.in_cs * d.W < (1L << 31) && P * std::max(d.out_cs, d.res_cs) < (1L << 31);
}

int gemm1x1sk_launch(const ConvDesc& d, const ConvCfg& cfg, hipStream_t stream) {
  if (!gemm1x1sk_cfg_valid(d, cfg)) {
    poco_set_error("gemm1x1sk: ALG 14 needs ks = 1, stride 1, (MT,NT,R) in {(7,4,2),(7,2,2|3),(4,4,2|3),(4,2,3),(2,4,3)}, WM (waves per block) 1..8 (1..4 for 7x4 tiles), WN = 1");
    return POCO_ERR_ARG;
  }
  if (!d.sk_scratch || d.sk_scratch_floats < gemm1x1sk_scratch_floats() || !d.sk_err_host) {
    poco_set_error("gemm1x1sk: ALG 14 needs its scratch buffer (flags + partials) and the error word");
    return POCO_ERR_ARG;
  }
  if ((d.in_cs | d.in_co | d.out_cs | d.out_co | d.res_cs | d.res_co) & 3) {
    poco_set_error("conv: channel strides/offsets must be multiples of 4");
    return POCO_ERR_ARG;
  }
  SKParams p{};
  p.in = d.in + l16_chan_off(d.in_co, d.W);
  p.res = d.res ? d.res + l16_chan_off(d.res_co, d.W) : nullptr;
  p.out = d.out + l16_chan_off(d.out_co, d.W);
  p.wfrag = reinterpret_cast<const float4*>(d.wfrag); p.bias = d.bias;
  p.P = d.B * d.H * d.W; p.W = d.W;
  p.nC16 = d.Cin / 16; p.nT16 = d.Cout / 16;
  p.in_rs = d.in_cs * d.W; p.in_ss = d.W * 16;
  p.res_rs = d.res_cs * d.W; p.out_rs = d.out_cs * d.W; p.out_ss = d.W * 16;
  p.act = d.act; p.res_after_act = d.res_after_act; p.relu_from = d.relu_from;
  p.dW = make_fastdiv(d.W);
  const int mtiles = (p.P + 15) / 16;
  const int ntm = (mtiles + cfg.MT - 1) / cfg.MT;
  p.ntn = (p.nT16 + cfg.NT - 1) / cfg.NT;
  p.ntiles = ntm * p.ntn;
  const long long U = (long long)p.ntiles * p.nC16;
  int blocks = SK_BLOCKS;
  while (blocks > 1 && (long long)blocks * cfg.WM * 4 > U) blocks /= 2;      // tiny problems: at least 4 slices per wave
  p.nwaves = blocks * cfg.WM;
  p.ngroups = (p.ntn <= 32 && p.nwaves % p.ntn == 0 && cfg.WM % std::min(p.ntn, cfg.WM) == 0) ? p.ntn : 1;   // see the kernel
  p.flag = reinterpret_cast<unsigned*>(d.sk_scratch);
  p.part = d.sk_scratch + SK_MAX_WAVES;
  p.err_host = d.sk_err_host;
  p.max_spins = d.sk_max_spins ? d.sk_max_spins : (1u << 21);
  const dim3 grid(blocks), block(cfg.WM * 64);
  // cfg.NI: 1 = hipcc's order; 3 = a load per 4 MFMAs from the start of the slice; 6 = per 4 at its end (gemm1x1.hip g1_sched)
#define SK_CASE(mt, nt, dd) if (cfg.MT == mt && cfg.NT == nt && cfg.R == dd) { \
    if (cfg.NI == 3) hipLaunchKernelGGL((gemm1x1sk_kernel<mt, nt, dd, 4>), grid, block, 0, stream, p); \
    else if (cfg.NI == 6) hipLaunchKernelGGL((gemm1x1sk_kernel<mt, nt, dd, 16 + 4>), grid, block, 0, stream, p); \
    else hipLaunchKernelGGL((gemm1x1sk_kernel<mt, nt, dd, 0>), grid, block, 0, stream, p); \
    POCO_HIP_CHECK(hipGetLastError()); return POCO_OK; }
  SK_CASE(7, 4, 2) SK_CASE(7, 2, 2) SK_CASE(7, 2, 3) SK_CASE(4, 4, 2) SK_CASE(4, 4, 3) SK_CASE(4, 2, 3) SK_CASE(2, 4, 3)
#undef SK_CASE
  poco_set_error("gemm1x1sk: unsupported tile");
  return POCO_ERR_ARG;
}
