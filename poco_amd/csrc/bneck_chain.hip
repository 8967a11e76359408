// Tail of one Bottleneck chained with the head of the next (planes = 64: `layer1` of HRNet, hrnet.py:79-99, and of
// ResNet-50, resnet.py:101-121):
//     y = ReLU(bn3(conv3(t)) + x)        t [P,64] -> y [P,256]      (block k:   conv3 + residual + ReLU)
//     u = ReLU(bn1(conv1(y)))            y [P,256] -> u [P,64]      (block k+1: conv1)
// As two launches the 256-channel tensor y (205 MB at 64 crops) is written by the first and read back by the second.
// Here it never leaves the registers between the two GEMMs: in these kernels weights are the MFMA A operand and
// pixels the B operand, so the accumulator of output n-tile c (lane (idx, g): channels 16c + 4g .. +3 of pixel idx)
// IS the B operand of K slice c of the next 1x1 conv.  A wave owns one 16-pixel sub-tile and all 256 channels of it:
// 256 MFMAs into 16 accumulators, epilogue (y is stored: the next block needs it as its residual), 256 MFMAs into 4
// accumulators.  Both weight matrices (2 x 64 KiB of packed fragments) stay resident in LDS for the whole kernel; the
// blocks are persistent and their waves walk the pixel sub-tiles independently (no barrier after the weight load).
#include "conv_mfma_types.h"
#include "kernels.h"

namespace {

struct ChainParams {
  const float* t;
  const float* res;
  float* y;
  float* u;
  const float4* w3;      // packed fragments [4 slices][16 n-tiles][64]
  const float* b3;       // [256]
  const float4* w1;      // packed fragments [16 slices][4 n-tiles][64]
  const float* b1;       // [64]
  int P, W, ntiles;      // pixels, plane width, 16-pixel sub-tiles
  int t_rs, t_ss, res_rs, res_ss, y_rs, y_ss, u_rs, u_ss;   // L16 strides: image row / 16-channel slice of a row
  FastDiv dW;
};

constexpr int CH_W3 = 4 * 16 * 64, CH_W1 = 16 * 4 * 64;      // float4 per weight matrix
constexpr size_t CH_LDS = (size_t)(CH_W3 + CH_W1 + 64 + 16) * sizeof(float4);

__global__ void __launch_bounds__(512)
bneck_chain_kernel(const ChainParams p) {
  extern __shared__ float4 lds[];
  float4* w3s = lds;
  float4* w1s = lds + CH_W3;
  float4* b3s = w1s + CH_W1;     // [16 n-tiles][4 quads]
  float4* b1s = b3s + 64;        // [4 n-tiles][4 quads]
  for (int i = threadIdx.x; i < CH_W3; i += blockDim.x) w3s[i] = p.w3[i];
  for (int i = threadIdx.x; i < CH_W1; i += blockDim.x) w1s[i] = p.w1[i];
  if (threadIdx.x < 64) b3s[threadIdx.x] = reinterpret_cast<const float4*>(p.b3)[threadIdx.x];
  if (threadIdx.x < 16) b1s[threadIdx.x] = reinterpret_cast<const float4*>(p.b1)[threadIdx.x];
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwaves = blockDim.x >> 6;
  const int idx = lane & 15, g = lane >> 4;
  // Software pipeline over the wave's sub-tiles: the operands of tile k+1 (t: 4 float4, residual: 16 float4 per lane) are requested
  // right after the epilogue of tile k - `tb` / `rr` are dead by then - and travel under the 256 MFMAs of its second GEMM and the
  // first GEMM of tile k+1; before, every tile began with an exposed load of `t` and fetched its residual with only half a GEMM of
  // cover (148 us for a launch whose MFMA and HBM floors are both ~92 us).  Same arithmetic, same results.
  const int tstep = gridDim.x * nwaves;
  float4 tb[4], rr[16];
  auto tile_addr = [&](int mt, const float** tp, const float** rp, float** yp, float** up, bool* valid) {
    const int pix = mt * 16 + idx;
    *valid = pix < p.P;
    const uint32_t pc = (uint32_t)min(pix, p.P - 1);         // dead lanes recompute the last pixel
    const uint32_t row = fdiv(pc, p.dW);
    const int xo = (int)(pc - row * (uint32_t)p.W) * 16 + 4 * g;
    *tp = p.t + (size_t)row * p.t_rs + xo;
    *rp = p.res + (size_t)row * p.res_rs + xo;
    *yp = p.y + (size_t)row * p.y_rs + xo;
    *up = p.u + (size_t)row * p.u_rs + xo;
  };
  auto fetch = [&](int mt) {
    const float *tp, *rp; float *yq, *uq; bool v;
    tile_addr(min(mt, p.ntiles - 1), &tp, &rp, &yq, &uq, &v);   // past the end: a harmless re-read of the last tile (no predicated loads)
#pragma unroll
    for (int c = 0; c < 4; ++c) tb[c] = *reinterpret_cast<const float4*>(tp + c * p.t_ss);
#pragma unroll
    for (int n = 0; n < 16; ++n) rr[n] = *reinterpret_cast<const float4*>(rp + n * p.res_ss);
  };
  const int mt0 = blockIdx.x * nwaves + wave;
  if (mt0 < p.ntiles) fetch(mt0);
  for (int mt = mt0; mt < p.ntiles; mt += tstep) {
    const float *tp_, *rp_; float *yp, *up; bool valid;
    tile_addr(mt, &tp_, &rp_, &yp, &up, &valid);

    f32x4 acc[16];
#pragma unroll
    for (int n = 0; n < 16; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float bv[4] = {tb[c].x, tb[c].y, tb[c].z, tb[c].w};
      // groups of 4 n-tiles, k-step outermost inside a group: consecutive MFMAs never share an accumulator
#pragma unroll
      for (int n0 = 0; n0 < 16; n0 += 4) {
        float4 a[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) a[n] = w3s[(c * 16 + n0 + n) * 64 + lane];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int n = 0; n < 4; ++n) {
            const float aj = (j == 0) ? a[n].x : (j == 1) ? a[n].y : (j == 2) ? a[n].z : a[n].w;
            acc[n0 + n] = __builtin_amdgcn_mfma_f32_16x16x4f32(aj, bv[j], acc[n0 + n], 0, 0, 0);
          }
      }
    }
    // y = ReLU(acc + shift + residual): stored, and kept in place as the B operand of the second GEMM
#pragma unroll
    for (int n = 0; n < 16; ++n) {
      const float4 b = b3s[n * 4 + g];
      f32x4 v = acc[n];
      v[0] = fmaxf(v[0] + b.x + rr[n].x, 0.f); v[1] = fmaxf(v[1] + b.y + rr[n].y, 0.f);
      v[2] = fmaxf(v[2] + b.z + rr[n].z, 0.f); v[3] = fmaxf(v[3] + b.w + rr[n].w, 0.f);
      acc[n] = v;
      if (valid) *reinterpret_cast<float4*>(yp + n * p.y_ss) = make_float4(v[0], v[1], v[2], v[3]);
    }
    __builtin_amdgcn_sched_barrier(0);
    fetch(mt + tstep);                                   // operands of the wave's next tile: under GEMM 2 and the next GEMM 1
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc2[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) acc2[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      float4 a[4];
#pragma unroll
      for (int n = 0; n < 4; ++n) a[n] = w1s[(c * 4 + n) * 64 + lane];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          const float aj = (j == 0) ? a[n].x : (j == 1) ? a[n].y : (j == 2) ? a[n].z : a[n].w;
          acc2[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(aj, acc[c][j], acc2[n], 0, 0, 0);
        }
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const float4 b = b1s[n * 4 + g];
      if (valid)
        *reinterpret_cast<float4*>(up + n * p.u_ss) = make_float4(fmaxf(acc2[n][0] + b.x, 0.f), fmaxf(acc2[n][1] + b.y, 0.f),
                                                                  fmaxf(acc2[n][2] + b.z, 0.f), fmaxf(acc2[n][3] + b.w, 0.f));
    }
  }
}

}  // namespace

// t [B,H,W,64 of t_cs], res / y [B,H,W,256 of res_cs / y_cs], u [B,H,W,64 of u_cs] (all L16, pointers at channel offset 0 of
// their slice); w3 / w1: conv_pack_weights(ks = 1) of the BN-folded 64->256 / 256->64 weights, b3 / b1 the folded shifts.
int launch_bneck_chain(const float* t, int t_cs, const float* res, int res_cs, float* y, int y_cs, float* u, int u_cs,
                       const float* w3, const float* b3, const float* w1, const float* b1, int B, int H, int W,
                       hipStream_t s) {
  static thread_local bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bneck_chain_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    if (e != hipSuccess) { poco_set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return POCO_ERR_HIP; }
    configured = true;
  }
  ChainParams p{};
  p.t = t; p.res = res; p.y = y; p.u = u;
  p.w3 = reinterpret_cast<const float4*>(w3); p.b3 = b3; p.w1 = reinterpret_cast<const float4*>(w1); p.b1 = b1;
  p.P = B * H * W; p.W = W; p.ntiles = (p.P + 15) / 16;
  p.t_rs = t_cs * W; p.res_rs = res_cs * W; p.y_rs = y_cs * W; p.u_rs = u_cs * W;
  p.t_ss = p.res_ss = p.y_ss = p.u_ss = W * 16;
  p.dW = make_fastdiv(W);
  const int grid = std::min(256, (p.ntiles + 7) / 8);
  hipLaunchKernelGGL(bneck_chain_kernel, dim3(grid), dim3(512), CH_LDS, s, p);
  POCO_HIP_CHECK(hipGetLastError());
  return POCO_OK;
}
