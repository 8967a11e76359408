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
#include "conv_mfma_types.h"

namespace {

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

int poco_num_cus() {
  static thread_local int dev_cached = -1, cus = 256;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return cus;
  if (dev != dev_cached) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    dev_cached = dev;
  }
  return cus;
}

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
  if (cfg.ALG == 6) return gemm1x1_cfg_valid(d, cfg) ? 16 : 0;      // no LDS; non-zero = "valid" for the callers
  if (cfg.ALG == 9) return gemm1x1t_lds_bytes(d, cfg);
  if (cfg.ALG == 14) return gemm1x1sk_cfg_valid(d, cfg) ? 16 : 0;   // no LDS; non-zero = "valid" for the callers
  if (cfg.ALG == 10) return gemm3x3_cfg_valid(d, cfg) ? 16 : 0;     // no LDS; non-zero = "valid" for the callers
  if (cfg.ALG == 11) return conv_wino4g_cfg_valid(d, cfg) ? 16 : 0;
#if POCO_EXPERIMENTS
  if (cfg.ALG == 12) return gemm1x1h_cfg_valid(d, cfg) ? 16 : 0;
#else
  if (cfg.ALG == 12) return 0;
#endif
  if (cfg.ALG == 7) return conv_wino4_lds_bytes(d, cfg);
  if (cfg.ALG == 8) return conv_wino4p_lds_bytes(d, cfg);
  if (cfg.ALG == 13) return conv_wino4w_lds_bytes(d, cfg);
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
  if (cfg.ALG == 6) return gemm1x1_launch(d, cfg, stream);
  if (cfg.ALG == 9) return gemm1x1t_launch(d, cfg, stream);
  if (cfg.ALG == 14) return gemm1x1sk_launch(d, cfg, stream);
  if (cfg.ALG == 10) return gemm3x3_launch(d, cfg, stream);
  if (cfg.ALG == 11) return conv_wino4g_launch(d, cfg, stream);
#if POCO_EXPERIMENTS
  if (cfg.ALG == 12) return gemm1x1h_launch(d, cfg, stream);
#else
  if (cfg.ALG == 12) { poco_set_error("conv: ALG 12 (split-fp16 experiment) is not part of this build (python -m poco_amd.build --experiments)"); return POCO_ERR_ARG; }
#endif
  if (cfg.ALG == 7) return conv_wino4_launch(d, cfg, stream);
  if (cfg.ALG == 8) return conv_wino4p_launch(d, cfg, stream);
  if (cfg.ALG == 13) return conv_wino4w_launch(d, cfg, stream);
  if (cfg.ALG == 3 || cfg.ALG == 4) {
    if (d.Cin % 16 || d.Cout % 16 || ((d.in_cs | d.in_co | d.out_cs | d.out_co | d.res_cs | d.res_co) & 3)) {
      poco_set_error("conv: channel counts/strides must be multiples of 16/4");
      return POCO_ERR_ARG;
    }
    if (d.act == 3 || d.act == 2) { poco_set_error("conv: the Winograd kernels support no activation or ReLU only"); return POCO_ERR_ARG; }
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
  kp.repeat = 1; kp.dbg = 0;
#if POCO_PROBES       // timing-probe builds only (tools/build_exp.sh conv_mfma.hip POCO_PROBES 1): results are then garbage
  {
    static const int rep = [] { const char* e = getenv("POCO_CONV_REPEAT"); return e ? atoi(e) : 1; }();
    kp.repeat = rep >= 0 ? rep : 1;
    static const int dbg = [] { const char* e = getenv("POCO_CONV_DBG"); return e ? atoi(e) : 0; }();
    kp.dbg = dbg;
  }
#endif
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
  if (d.ks == 1 && d.stride == 1) return conv_launch_k1s1(cfg.ALG, cfg.MT, cfg.NT, kp, grid, nthreads, lds, stream);
  if (d.ks == 1 && d.stride == 2) return conv_launch_k1s2(cfg.ALG, cfg.MT, cfg.NT, kp, grid, nthreads, lds, stream);
  if (d.ks == 3 && d.stride == 1) return conv_launch_k3s1(cfg.ALG, cfg.MT, cfg.NT, kp, grid, nthreads, lds, stream);
  return conv_launch_k3s2(cfg.ALG, cfg.MT, cfg.NT, kp, grid, nthreads, lds, stream);
}
