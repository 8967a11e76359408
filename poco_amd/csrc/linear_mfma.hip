// Small-M linear layers (H = W = 1 "convs": the CLIFF regressor fc1/fc2/dec and the poco_head MLPs,
// pocolib/models/head/cliff_head.py:104-118, head/poco_head.py:96-154) and small-plane 1x1 stride-1 convs
// (the HRNet up-path fuse convs, hrnet.py:196-207) — ALG 5.
//
// With 32..128 crops per forward these GEMMs have M = B rows only: 0.1 GFLOP against 4..9 MB of weights, so
// they are latency-bound, not MFMA-bound.  The implicit-GEMM conv kernel runs them as Cout/16 blocks that
// each walk the whole K dimension serially (60..230 us); here one block still owns 16 output features, but
// the K dimension is split over the block's waves, every wave has all of its loads in flight before the
// first MFMA, and the partial sums meet once in LDS:  ~2 dependent memory round trips per layer.
//
// Operand roles as in the conv kernels: weights = MFMA A operand (packed fragment order of
// conv_pack_weights, ks = 1), activation rows = B operand (lane (idx, g) reads x[row idx][16c + 4g .. +3]
// as one float4), so a lane ends up with 4 consecutive output features of one row -> 16-B stores.
#include "common.h"

namespace {

struct LinParams {
  const float* in;       // slice offsets are folded into the pointers
  const float* res;
  float* out;
  const float4* wfrag;   // [Cin/16][Cout16/16][64] float4
  const float* bias;
  int P, W, nC16, nT16;        // P = B*H*W rows ("pixels"), W = plane width (1 for Linear layers)
  int in_rs, in_ss, res_rs, out_rs, out_ss;   // L16 strides: image row / 16-channel slice of a row
  uint32_t w_magic;            // fast division by W
  int act, res_after_act, relu_from;
};

constexpr int LIN_MB = 4;      // 16-row sub-tiles per block (64 rows); grid.y walks larger batches
constexpr int LIN_UNROLL = 4;  // K slices whose loads are issued together

__global__ void __launch_bounds__(1024)
linear_mfma_kernel(const LinParams p) {
  extern __shared__ float4 red[];          // [wave][LIN_MB][64] partial accumulators
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwaves = blockDim.x >> 6;
  const int idx = lane & 15, g = lane >> 4;
  const int nt = blockIdx.x;
  const int row0 = blockIdx.y * (LIN_MB * 16);     // first pixel of this block

  // pixel -> (image row, x); L16: slice c of pixel (row, x) starts at row*in_rs + c*in_ss + x*16
  auto split = [&](int pix, int* row, int* x) {
    *row = p.W == 1 ? pix : (int)__umulhi((uint32_t)pix, p.w_magic);
    *x = pix - *row * p.W;
  };
  const float* xrow[LIN_MB];
#pragma unroll
  for (int m = 0; m < LIN_MB; ++m) {
    int row, x;
    split(min(row0 + m * 16 + idx, p.P - 1), &row, &x);             // dead rows re-read the last one
    xrow[m] = p.in + (size_t)row * p.in_rs + x * 16 + 4 * g;
  }
  f32x4 acc[LIN_MB];
#pragma unroll
  for (int m = 0; m < LIN_MB; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const float4* wl = p.wfrag + (size_t)nt * 64 + lane;
  const size_t wstride = (size_t)p.nT16 * 64;                        // float4 per K slice
  for (int c0 = wave; c0 < p.nC16; c0 += nwaves * LIN_UNROLL) {
    float4 a[LIN_UNROLL], b[LIN_UNROLL][LIN_MB];
#pragma unroll
    for (int u = 0; u < LIN_UNROLL; ++u) {
      const int c = min(c0 + u * nwaves, p.nC16 - 1);                // clamped duplicates are masked below
      a[u] = wl[(size_t)c * wstride];
#pragma unroll
      for (int m = 0; m < LIN_MB; ++m) b[u][m] = *reinterpret_cast<const float4*>(xrow[m] + (size_t)c * p.in_ss);
    }
#pragma unroll
    for (int u = 0; u < LIN_UNROLL; ++u) {
      if (c0 + u * nwaves < p.nC16) {
        const float av[4] = {a[u].x, a[u].y, a[u].z, a[u].w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int m = 0; m < LIN_MB; ++m) {
            const float bv = (j == 0) ? b[u][m].x : (j == 1) ? b[u][m].y : (j == 2) ? b[u][m].z : b[u][m].w;
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv, acc[m], 0, 0, 0);
          }
      }
    }
  }
#pragma unroll
  for (int m = 0; m < LIN_MB; ++m)
    red[(wave * LIN_MB + m) * 64 + lane] = make_float4(acc[m][0], acc[m][1], acc[m][2], acc[m][3]);
  __syncthreads();
  // wave m (m < LIN_MB) finishes sub-tile m: sum over the K-split waves in a fixed order, then the epilogue
  for (int m = wave; m < LIN_MB; m += nwaves) {
    float4 s = red[m * 64 + lane];
    for (int w = 1; w < nwaves; ++w) {
      const float4 t = red[(w * LIN_MB + m) * 64 + lane];
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    const int r = row0 + m * 16 + idx;
    if (r >= p.P) continue;
    int orow, ox;
    split(r, &orow, &ox);
    const int co = nt * 16 + g * 4;
    const size_t so = (size_t)nt * p.out_ss + ox * 16 + g * 4;       // slice + pixel offset inside an image row
    const float4 sh = *reinterpret_cast<const float4*>(p.bias + co);
    float v[4] = {s.x + sh.x, s.y + sh.y, s.z + sh.z, s.w + sh.w};
    float4 rr = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.res) rr = *reinterpret_cast<const float4*>(p.res + (size_t)orow * p.res_rs + so);
    if (!p.res_after_act) { v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w; }
    if (p.act == 1 || (p.act == 3 && co >= p.relu_from)) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    } else if (p.act == 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = 1.f / (1.f + __expf(-v[e]));
    }
    if (p.res_after_act) { v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w; }
    *reinterpret_cast<float4*>(p.out + (size_t)orow * p.out_rs + so) = make_float4(v[0], v[1], v[2], v[3]);
  }
}


// ---- the same scheme for 3x3 convs (stride 1 | 2, pad 1) on SMALL batches (round 4) ------------------------------------------------
// At 1-4 crops a 3x3 conv of the HRNet branches is a launch of 6-24 workgroups that walk K = 9 Cin / 16 slices one after the other
// (7x7 384->384, one crop: 24 blocks x 24 slices of the Winograd kernel = 30 us for 0.13 GFLOP; the stride-2 fuse convs: 6 blocks,
// 71 us) - the forward of one crop is a chain of such launches.  Here the conv is the direct GEMM D[co][pix] = sum_{tap, ci} over
// K = 9 Cin / 16 steps with the steps dealt round-robin to the 4-16 waves of a block (weights 5.3 MB instead of the 9.4 MB of F(2x2)
// fragments: the regime is bound by the weight stream), all loads of a wave's 4 steps in flight before its first MFMA, one LDS
// reduction in a fixed order, epilogue by the first waves.  Zero padding as in gemm3x3.hip: a lane whose tap falls outside the image
// loads its centre pixel and the value is replaced by zero.
// v where `ok`, zeros elsewhere - as a MULTIPLICATION by 1 / 0, which the optimiser cannot turn back into a predicated load.  A lane
// whose tap is padding loaded the CENTRE pixel of its own window: a finite value whenever the true output is finite (that pixel
// contributes to the same output through the centre tap), so 0 * v is an exact zero in every case that matters.
// CAVEAT: if that centre pixel is +-Inf or NaN, 0 * v is NaN where true zero padding contributes 0 - an output that
// the reference computes as +-Inf (an overflowed activation next to the border) comes out as NaN on this path only, i.e. results
// could then differ between batch sizes that use ALG 5 (1 ... 4 crops) and those that do not.  Not reachable with finite
// activations (every test and every trained network in range); a bitwise mask would be exact but `v & (ok ? ~0 : 0)` is folded back
// into select(ok, v, 0) by instcombine, i.e. into the predicated load this function exists to avoid.
__device__ __forceinline__ float4 select_or_zero(float4 v, bool ok) {
  const float k = ok ? 1.f : 0.f;
  return make_float4(v.x * k, v.y * k, v.z * k, v.w * k);
}

struct Lin3Params {
  LinParams l;
  int H, Wi, Ho, Wo, stride;   // input plane H x Wi, output plane Ho x Wo (l.W = Wo, l.P = B*Ho*Wo)
  int nK;                      // 9 * nC16 K steps, step c = tap * nC16 + slice (the order of conv_pack_weights(ks = 3))
  uint32_t ho_magic, nc_magic; // fast division by Ho and nC16
};

// <MB, UN>: 16-pixel sub-tiles per block x K steps whose loads a wave has in flight together.  <4, 4> (rounds 4-5): a block owns 64 pixels; <1, 8>
// (round 6, cfg.R = 2): 16 pixels per block = 4 x the blocks on the small planes and twice the steps in flight per wave - at one crop a
// 7x7 384->384 conv is 24 blocks whose waves walk 13.5 steps in four dependent rounds of loads (28 us for 5.3 MB of weights)
template <int MB, int UN>
__global__ void __launch_bounds__(1024)
conv3x3_splitk_kernel(const Lin3Params q) {
  constexpr int LIN_MB = MB, LIN_UNROLL = UN;
  const LinParams& p = q.l;
  extern __shared__ float4 red[];          // [wave][LIN_MB][64] partial accumulators
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwaves = blockDim.x >> 6;
  const int idx = lane & 15, g = lane >> 4;
  const int nt = blockIdx.x;
  const int row0 = blockIdx.y * (LIN_MB * 16);     // first output pixel of this block
  auto split = [&](int pix, int* row, int* x) {    // output pixel -> (b*Ho + oy, ox)
    *row = p.W == 1 ? pix : (int)__umulhi((uint32_t)pix, p.w_magic);
    *x = pix - *row * p.W;
  };
  const float* xc[LIN_MB];     // centre tap of this lane's pixel (slice 0, channel quad g)
  int vmask[LIN_MB];           // bit (3r + s): tap (r, s) lies inside the image
#pragma unroll
  for (int m = 0; m < LIN_MB; ++m) {
    int row, x;
    split(min(row0 + m * 16 + idx, p.P - 1), &row, &x);
    const int b = q.Ho == 1 ? row : (int)__umulhi((uint32_t)row, q.ho_magic);
    const int iy = (row - b * q.Ho) * q.stride, ix = x * q.stride;
    xc[m] = p.in + (size_t)(b * q.H + iy) * p.in_rs + ix * 16 + 4 * g;
    const int ym = (iy >= 1 ? 1 : 0) | 2 | (iy + 1 < q.H ? 4 : 0);
    const int xm = (ix >= 1 ? 1 : 0) | 2 | (ix + 1 < q.Wi ? 4 : 0);
    vmask[m] = ((ym & 1) ? xm : 0) | (xm << 3) | ((ym & 4) ? (xm << 6) : 0);
  }
  f32x4 acc[LIN_MB];
#pragma unroll
  for (int m = 0; m < LIN_MB; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float4* wl = p.wfrag + (size_t)nt * 64 + lane;
  const size_t wstride = (size_t)p.nT16 * 64;                        // float4 per K step
  for (int c0 = wave; c0 < q.nK; c0 += nwaves * LIN_UNROLL) {
    float4 a[LIN_UNROLL], b[LIN_UNROLL][LIN_MB];
#pragma unroll
    for (int u = 0; u < LIN_UNROLL; ++u) {
      const int c = min(c0 + u * nwaves, q.nK - 1);                  // wave-uniform; clamped duplicates are masked below
      const int tap = p.nC16 == 1 ? c : (int)__umulhi((uint32_t)c, q.nc_magic);
      const int c16 = c - tap * p.nC16;
      const int r = (tap * 11) >> 5;                                 // tap / 3 for tap < 9
      const int toff = (r - 1) * p.in_rs + (tap - 3 * r - 1) * 16;
      a[u] = wl[(size_t)c * wstride];
#pragma unroll
      for (int m = 0; m < LIN_MB; ++m) {
        // the value is zeroed where it is CONSUMED (opaque select below): with `ok ? v : 0` here hipcc predicates the load
        // (s_cbranch_execz around every gather), stops counting the loads in flight and waits for each group of a K step before it
        // requests the next one - one step in flight instead of four
        const bool ok = (vmask[m] >> tap) & 1;
        b[u][m] = *reinterpret_cast<const float4*>(xc[m] + (ok ? toff : 0) + (size_t)c16 * p.in_ss);
      }
    }
#pragma unroll
    for (int u = 0; u < LIN_UNROLL; ++u) {
      if (c0 + u * nwaves < q.nK) {
        {
          const int c = c0 + u * nwaves;
          const int tap = p.nC16 == 1 ? c : (int)__umulhi((uint32_t)c, q.nc_magic);
#pragma unroll
          for (int m = 0; m < LIN_MB; ++m) b[u][m] = select_or_zero(b[u][m], (vmask[m] >> tap) & 1);
        }
        const float av[4] = {a[u].x, a[u].y, a[u].z, a[u].w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int m = 0; m < LIN_MB; ++m) {
            const float bv = (j == 0) ? b[u][m].x : (j == 1) ? b[u][m].y : (j == 2) ? b[u][m].z : b[u][m].w;
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv, acc[m], 0, 0, 0);
          }
      }
    }
  }
#pragma unroll
  for (int m = 0; m < LIN_MB; ++m)
    red[(wave * LIN_MB + m) * 64 + lane] = make_float4(acc[m][0], acc[m][1], acc[m][2], acc[m][3]);
  __syncthreads();
  for (int m = wave; m < LIN_MB; m += nwaves) {
    float4 s = red[m * 64 + lane];
    for (int w = 1; w < nwaves; ++w) {
      const float4 t = red[(w * LIN_MB + m) * 64 + lane];
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    const int r = row0 + m * 16 + idx;
    if (r >= p.P) continue;
    int orow, ox;
    split(r, &orow, &ox);
    const int co = nt * 16 + g * 4;
    const size_t so = (size_t)nt * p.out_ss + ox * 16 + g * 4;
    const float4 sh = *reinterpret_cast<const float4*>(p.bias + co);
    float v[4] = {s.x + sh.x, s.y + sh.y, s.z + sh.z, s.w + sh.w};
    float4 rr = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.res) rr = *reinterpret_cast<const float4*>(p.res + (size_t)orow * p.res_rs + so);
    if (!p.res_after_act) { v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w; }
    if (p.act == 1 || (p.act == 3 && co >= p.relu_from)) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    } else if (p.act == 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = 1.f / (1.f + __expf(-v[e]));
    }
    if (p.res_after_act) { v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w; }
    *reinterpret_cast<float4*>(p.out + (size_t)orow * p.out_rs + so) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

}  // namespace

bool linear_cfg_valid(const ConvDesc& d, const ConvCfg& cfg) {
  if (d.ks == 3)      // the split-K direct 3x3 conv (small batches): stride 1 | 2, pad 1
    return (d.stride == 1 || d.stride == 2) && cfg.WM >= 1 && cfg.WM <= 16 && d.Cin % 16 == 0 && d.Cout % 16 == 0 && d.Cin <= 16384 &&
           (long)d.B * d.H * d.W < (1L << 24) && (long)d.B * d.H * d.W * std::max(std::max(d.in_cs, d.out_cs), d.res_cs) < (1L << 31);
  return d.ks == 1 && d.stride == 1 && cfg.WM >= 1 && cfg.WM <= 16 && d.Cin % 16 == 0 && d.Cout % 16 == 0 &&
         (long)d.B * d.H * d.W < (1L << 30);
}

int linear_launch(const ConvDesc& d, const ConvCfg& cfg, hipStream_t stream) {
  if (!linear_cfg_valid(d, cfg)) {
    poco_set_error("linear: ALG 5 needs ks = 1 stride 1 or ks = 3 stride 1|2, WM (K-split waves) in 1..16");
    return POCO_ERR_ARG;
  }
  if (d.ks == 3) {
    Lin3Params q{};
    LinParams& p = q.l;
    q.H = d.H; q.Wi = d.W; q.stride = d.stride;
    q.Ho = (d.H - 1) / d.stride + 1; q.Wo = (d.W - 1) / d.stride + 1;
    p.in = d.in + l16_chan_off(d.in_co, d.W);
    p.res = d.res ? d.res + l16_chan_off(d.res_co, q.Wo) : nullptr;
    p.out = d.out + l16_chan_off(d.out_co, q.Wo);
    p.wfrag = reinterpret_cast<const float4*>(d.wfrag); p.bias = d.bias;
    p.P = d.B * q.Ho * q.Wo; p.W = q.Wo; p.nC16 = d.Cin / 16; p.nT16 = d.Cout / 16;
    p.in_rs = d.in_cs * d.W; p.in_ss = d.W * 16;
    p.res_rs = d.res_cs * q.Wo; p.out_rs = d.out_cs * q.Wo; p.out_ss = q.Wo * 16;
    p.w_magic = q.Wo > 1 ? (uint32_t)(((1ull << 32) + q.Wo - 1) / q.Wo) : 0;
    q.ho_magic = q.Ho > 1 ? (uint32_t)(((1ull << 32) + q.Ho - 1) / q.Ho) : 0;
    q.nc_magic = p.nC16 > 1 ? (uint32_t)(((1ull << 32) + p.nC16 - 1) / p.nC16) : 0;
    q.nK = 9 * p.nC16;
    p.act = d.act; p.res_after_act = d.res_after_act; p.relu_from = d.relu_from;
    const int nwaves = cfg.WM;
    const int mb = cfg.R == 2 ? 1 : cfg.R == 3 ? 2 : 4;            // cfg.R: 1 (and anything else) = <4, 4>, 2 = <1, 8>, 3 = <2, 6>
    const size_t lds = (size_t)nwaves * mb * 64 * sizeof(float4);
    const dim3 grid(p.nT16, (p.P + mb * 16 - 1) / (mb * 16));
    if (mb == 1) hipLaunchKernelGGL((conv3x3_splitk_kernel<1, 8>), grid, dim3(nwaves * 64), lds, stream, q);
    else if (mb == 2) hipLaunchKernelGGL((conv3x3_splitk_kernel<2, 6>), grid, dim3(nwaves * 64), lds, stream, q);
    else hipLaunchKernelGGL((conv3x3_splitk_kernel<4, 4>), grid, dim3(nwaves * 64), lds, stream, q);
    POCO_HIP_CHECK(hipGetLastError());
    return POCO_OK;
  }
  LinParams p{};
  p.in = d.in + l16_chan_off(d.in_co, d.W);
  p.res = d.res ? d.res + l16_chan_off(d.res_co, d.W) : nullptr;
  p.out = d.out + l16_chan_off(d.out_co, d.W);
  p.wfrag = reinterpret_cast<const float4*>(d.wfrag); p.bias = d.bias;
  p.P = d.B * d.H * d.W; p.W = d.W; p.nC16 = d.Cin / 16; p.nT16 = d.Cout / 16;
  p.in_rs = d.in_cs * d.W; p.in_ss = d.W * 16;
  p.res_rs = d.res_cs * d.W; p.out_rs = d.out_cs * d.W; p.out_ss = d.W * 16;
  p.w_magic = d.W > 1 ? (uint32_t)(((1ull << 32) + d.W - 1) / d.W) : 0;   // exact for pix < 2^30 / W-sized planes
  p.act = d.act; p.res_after_act = d.res_after_act; p.relu_from = d.relu_from;
  const int nwaves = cfg.WM;
  const size_t lds = (size_t)nwaves * LIN_MB * 64 * sizeof(float4);
  const dim3 grid(p.nT16, (p.P + LIN_MB * 16 - 1) / (LIN_MB * 16));
  hipLaunchKernelGGL(linear_mfma_kernel, grid, dim3(nwaves * 64), lds, stream, p);
  POCO_HIP_CHECK(hipGetLastError());
  return POCO_OK;
}
