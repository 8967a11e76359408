// Backbone side kernels: stem conv (Cin=3), max pool, bilinear x2, HRNet fuse-sum, global avg pool.
// All HBM-bound elementwise / small-reduction work on L16 (channel-slice-major NHWC, common.h) fp32 with 16-byte
// accesses; threads are ordered like the output memory.
#include "kernels.h"

namespace {

// ---- stem conv: NCHW [B,3,H,W] -> NHWC [B,Ho,Wo,64], stride 2, pad (KS-1)/2 ---------------------
// thread = (pixel, 16-channel group); weights [KS*KS*3][64] staged in LDS (broadcast reads).
template <int KS>
__global__ void __launch_bounds__(256)
stem_conv_kernel(const float* __restrict__ img, const float* __restrict__ w, const float* __restrict__ shift,
                 float* __restrict__ out, int B, int H, int W, int Ho, int Wo) {
  constexpr int K = KS * KS * 3;
  constexpr int PAD = (KS - 1) / 2;
  __shared__ float4 ws[K * 16];
  for (int i = threadIdx.x; i < K * 16; i += 256) ws[i] = reinterpret_cast<const float4*>(w)[i];
  __syncthreads();
  const int g = threadIdx.x & 3;
  const long pix = (long)blockIdx.x * 64 + (threadIdx.x >> 2);
  const long npix = (long)B * Ho * Wo;
  if (pix >= npix) return;
  const int xo = (int)(pix % Wo);
  const int yo = (int)((pix / Wo) % Ho);
  const int b = (int)(pix / ((long)Wo * Ho));
  float4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = reinterpret_cast<const float4*>(shift)[g * 4 + i];
  const float* ib = img + (size_t)b * 3 * H * W;
#pragma unroll 1
  for (int c = 0; c < 3; ++c) {
#pragma unroll 1
    for (int r = 0; r < KS; ++r) {
      const int iy = yo * 2 - PAD + r;
      if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int ix = xo * 2 - PAD + s;
        const float x = ((unsigned)ix < (unsigned)W) ? ib[((size_t)c * H + iy) * W + ix] : 0.f;
        const float4* wk = ws + ((r * KS + s) * 3 + c) * 16 + g * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 wv = wk[i];
          acc[i].x = fmaf(x, wv.x, acc[i].x); acc[i].y = fmaf(x, wv.y, acc[i].y);
          acc[i].z = fmaf(x, wv.z, acc[i].z); acc[i].w = fmaf(x, wv.w, acc[i].w);
        }
      }
    }
  }
  // L16: the 16 channels of group g are slice g of the 64-channel row
  float4* o = reinterpret_cast<float4*>(out) + L16_F4((size_t)b * Ho + yo, xo, g * 4, Wo, 4);
#pragma unroll
  for (int i = 0; i < 4; ++i)
    o[i] = make_float4(fmaxf(acc[i].x, 0.f), fmaxf(acc[i].y, 0.f), fmaxf(acc[i].z, 0.f), fmaxf(acc[i].w, 0.f));
}

// Register-tiled variant: a thread owns 4 horizontally adjacent output pixels x 16 channels, so one broadcast
// weight read (float4 from LDS) feeds 4 pixels and one image row segment (6 + KS floats) feeds all KS taps:
// 16 FMAs per LDS read instead of 4 (the 1-pixel kernel above is LDS-issue-bound).  Needs Wo % 4 == 0.
template <int KS>
__global__ void __launch_bounds__(256)
stem_conv4_kernel(const float* __restrict__ img, const float* __restrict__ w, const float* __restrict__ shift,
                  float* __restrict__ out, int B, int H, int W, int Ho, int Wo) {
  constexpr int K = KS * KS * 3;
  constexpr int PAD = (KS - 1) / 2;
  constexpr int NV = 6 + KS;               // input columns under 4 stride-2 outputs
  __shared__ float4 ws[K * 16];
  for (int i = threadIdx.x; i < K * 16; i += 256) ws[i] = reinterpret_cast<const float4*>(w)[i];
  __syncthreads();
  const int g = threadIdx.x & 3;
  const int Wq = Wo >> 2;
  const long quad = (long)blockIdx.x * 64 + (threadIdx.x >> 2);
  if (quad >= (long)B * Ho * Wq) return;
  const int xo0 = (int)(quad % Wq) * 4;
  const int yo = (int)((quad / Wq) % Ho);
  const int b = (int)(quad / ((long)Wq * Ho));
  float4 acc[4][4];
#pragma unroll
  for (int px = 0; px < 4; ++px)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[px][i] = reinterpret_cast<const float4*>(shift)[g * 4 + i];
  const float* ib = img + (size_t)b * 3 * H * W;
  const int ix0 = xo0 * 2 - PAD;
#pragma unroll 1
  for (int c = 0; c < 3; ++c) {
#pragma unroll 1
    for (int r = 0; r < KS; ++r) {
      const int iy = yo * 2 - PAD + r;
      if ((unsigned)iy >= (unsigned)H) continue;
      const float* row = ib + ((size_t)c * H + iy) * W;
      float v[NV];
#pragma unroll
      for (int j = 0; j < NV; ++j) v[j] = ((unsigned)(ix0 + j) < (unsigned)W) ? row[ix0 + j] : 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const float4* wk = ws + ((r * KS + s) * 3 + c) * 16 + g * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 wv = wk[i];
#pragma unroll
          for (int px = 0; px < 4; ++px) {
            const float x = v[2 * px + s];
            acc[px][i].x = fmaf(x, wv.x, acc[px][i].x); acc[px][i].y = fmaf(x, wv.y, acc[px][i].y);
            acc[px][i].z = fmaf(x, wv.z, acc[px][i].z); acc[px][i].w = fmaf(x, wv.w, acc[px][i].w);
          }
        }
      }
    }
  }
#pragma unroll
  for (int px = 0; px < 4; ++px) {
    float4* o = reinterpret_cast<float4*>(out) + L16_F4((size_t)b * Ho + yo, xo0 + px, g * 4, Wo, 4);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      o[i] = make_float4(fmaxf(acc[px][i].x, 0.f), fmaxf(acc[px][i].y, 0.f), fmaxf(acc[px][i].z, 0.f), fmaxf(acc[px][i].w, 0.f));
  }
}

__global__ void maxpool_kernel(const float4* __restrict__ in, float4* __restrict__ out, int B, int H, int W,
                               int Ho, int Wo, int C4, int outC16) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = (long)B * Ho * Wo * C4;
  if (i >= n) return;
  // thread order = L16 memory order of the output: quad, x, slice, y, b
  const int C16 = C4 >> 2;
  const int q = (int)(i & 3);
  long t = i >> 2;
  const int xo = (int)(t % Wo); t /= Wo;
  const int cb = (int)(t % C16); t /= C16;
  const int yo = (int)(t % Ho);
  const int b = (int)(t / Ho);
  const int c = cb * 4 + q;
  float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  for (int r = 0; r < 3; ++r) {
    const int iy = yo * 2 - 1 + r;
    if ((unsigned)iy >= (unsigned)H) continue;
    for (int s = 0; s < 3; ++s) {
      const int ix = xo * 2 - 1 + s;
      if ((unsigned)ix >= (unsigned)W) continue;
      const float4 v = in[L16_F4((size_t)b * H + iy, ix, c, W, C16)];
      m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
    }
  }
  out[L16_F4((size_t)b * Ho + yo, xo, c, Wo, outC16)] = m;     // outC16 > C16: a channel slice of a wider buffer
}

// torch F.interpolate(scale_factor=2, mode='bilinear', align_corners=True) semantics
__global__ void bilinear_up2x_kernel(const float4* __restrict__ in, float4* __restrict__ out, int B, int H, int W,
                                     int C4) {
  const int Ho = 2 * H, Wo = 2 * W;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = (long)B * Ho * Wo * C4;
  if (i >= n) return;
  const int C16 = C4 >> 2;
  const int q = (int)(i & 3);
  long t = i >> 2;
  const int xo = (int)(t % Wo); t /= Wo;
  const int cb = (int)(t % C16); t /= C16;
  const int yo = (int)(t % Ho);
  const int b = (int)(t / Ho);
  const int c = cb * 4 + q;
  const float sh = (Ho > 1) ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
  const float sw = (Wo > 1) ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
  const float fy = sh * yo, fx = sw * xo;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < H - 1), x1 = x0 + (x0 < W - 1);
  const float ly1 = fy - y0, lx1 = fx - x0;
  const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
  const size_t r0 = (size_t)b * H + y0, r1 = (size_t)b * H + y1;
  const float4 v00 = in[L16_F4(r0, x0, c, W, C16)], v01 = in[L16_F4(r0, x1, c, W, C16)];
  const float4 v10 = in[L16_F4(r1, x0, c, W, C16)], v11 = in[L16_F4(r1, x1, c, W, C16)];
  float4 o;
  o.x = ly0 * (lx0 * v00.x + lx1 * v01.x) + ly1 * (lx0 * v10.x + lx1 * v11.x);
  o.y = ly0 * (lx0 * v00.y + lx1 * v01.y) + ly1 * (lx0 * v10.y + lx1 * v11.y);
  o.z = ly0 * (lx0 * v00.z + lx1 * v01.z) + ly1 * (lx0 * v10.z + lx1 * v11.z);
  o.w = ly0 * (lx0 * v00.w + lx1 * v01.w) + ly1 * (lx0 * v10.w + lx1 * v11.w);
  out[i] = o;
}

__global__ void fuse_sum_kernel(FuseArgs a, float4* __restrict__ out, int B, int H, int W, int C4, int outC4,
                                int relu) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = (long)B * H * W * C4;
  if (i >= n) return;
  const int C16 = C4 >> 2;
  const int q = (int)(i & 3);
  long t = i >> 2;
  const int x = (int)(t % W); t /= W;
  const int cb = (int)(t % C16); t /= C16;
  const int y = (int)(t % H);
  const int b = (int)(t / H);
  const int c = cb * 4 + q;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (k < a.n) {
      const int sh = a.shift[k];
      const int hs = H >> sh, wsz = W >> sh;
      const float4 v = reinterpret_cast<const float4*>(a.src[k])[L16_F4((size_t)b * hs + (y >> sh), x >> sh, c, wsz, a.src_cs[k] >> 4)];
      if (k == 0) acc = v;
      else { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    }
  }
  if (relu) { acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f); }
  out[L16_F4((size_t)b * H + y, x, c, W, outC4 >> 2)] = acc;
}

__global__ void avgpool_kernel(const float4* __restrict__ in, float* __restrict__ dst, int B, int H, int W, int C4,
                               int dst_stride) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C4) return;
  const int c = i % C4, b = i / C4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int y = 0; y < H; ++y) {
    const float4* p = in + L16_F4((size_t)b * H + y, 0, c, W, C4 >> 2);
    for (int x = 0; x < W; ++x) {
      const float4 v = p[(size_t)x * 4];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  const float inv = 1.f / (float)(H * W);
  *reinterpret_cast<float4*>(dst + (size_t)b * dst_stride + c * 4) = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
}

inline int nblk(long n, int t) { return (int)((n + t - 1) / t); }

}  // namespace

void launch_stem_conv(const float* img, const float* w, const float* shift, float* out, int B, int H, int W,
                      int ks, hipStream_t s) {
  const int pad = (ks - 1) / 2;
  const int Ho = (H + 2 * pad - ks) / 2 + 1, Wo = (W + 2 * pad - ks) / 2 + 1;
  const long npix = (long)B * Ho * Wo;
  if (Wo % 4 == 0) {            // 4 pixels per thread
    const long nquad = npix / 4;
    if (ks == 3)
      hipLaunchKernelGGL(stem_conv4_kernel<3>, dim3(nblk(nquad, 64)), dim3(256), 0, s, img, w, shift, out, B, H, W, Ho, Wo);
    else
      hipLaunchKernelGGL(stem_conv4_kernel<7>, dim3(nblk(nquad, 64)), dim3(256), 0, s, img, w, shift, out, B, H, W, Ho, Wo);
  } else if (ks == 3)
    hipLaunchKernelGGL(stem_conv_kernel<3>, dim3(nblk(npix, 64)), dim3(256), 0, s, img, w, shift, out, B, H, W, Ho, Wo);
  else
    hipLaunchKernelGGL(stem_conv_kernel<7>, dim3(nblk(npix, 64)), dim3(256), 0, s, img, w, shift, out, B, H, W, Ho, Wo);
}

void launch_maxpool3x3s2(const float* in, float* out, int B, int H, int W, int C, int out_cs, hipStream_t s) {
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long n = (long)B * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(maxpool_kernel, dim3(nblk(n, 256)), dim3(256), 0, s, (const float4*)in, (float4*)out, B, H, W,
                     Ho, Wo, C / 4, out_cs / 16);
}

void launch_bilinear_up2x(const float* in, float* out, int B, int H, int W, int C, hipStream_t s) {
  const long n = (long)B * 4 * H * W * (C / 4);
  hipLaunchKernelGGL(bilinear_up2x_kernel, dim3(nblk(n, 256)), dim3(256), 0, s, (const float4*)in, (float4*)out, B,
                     H, W, C / 4);
}

void launch_fuse_sum(const FuseArgs& a, float* out, int B, int H, int W, int C, int out_cs, int relu, hipStream_t s) {
  const long n = (long)B * H * W * (C / 4);
  hipLaunchKernelGGL(fuse_sum_kernel, dim3(nblk(n, 256)), dim3(256), 0, s, a, (float4*)out, B, H, W, C / 4, out_cs / 4,
                     relu);
}

void launch_avgpool(const float* in, float* dst, int B, int H, int W, int C, int dst_stride, hipStream_t s) {
  const int n = B * (C / 4);
  hipLaunchKernelGGL(avgpool_kernel, dim3(nblk(n, 256)), dim3(256), 0, s, (const float4*)in, dst, B, H, W, C / 4,
                     dst_stride);
}

// ---- GPU-side crop + normalise ---------------------------------------------------
// Replaces the per-crop CPU loop  cv2.warpAffine(bilinear, BORDER_CONSTANT) -> ToTensor -> Normalize
// + per-crop H2D copy of pocolib/core/tester.py:182-203 / utils/vibe_image_utils.py:94-107,233-266,
// 343-351.  One thread per output pixel, frame read once from HBM (uint8 HWC RGB), output NCHW fp32.
namespace {
__global__ void crop_normalize_kernel(const unsigned char* __restrict__ frame, int H, int W,
                                      const float* __restrict__ boxes, float bbox_scale, float* __restrict__ out,
                                      int N, int res) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)N * res * res) return;
  const int x = (int)(i % res), y = (int)((i / res) % res), n = (int)(i / ((long)res * res));
  const float cx = boxes[n * 4], cy = boxes[n * 4 + 1], bw = boxes[n * 4 + 2], bh = boxes[n * 4 + 3];
  // gen_trans_from_patch_cv (rot = 0): src = centre + (dst - res/2) * (bbox * scale / res)
  const float sx = cx + ((float)x - 0.5f * res) * (bw * bbox_scale / res);
  const float sy = cy + ((float)y - 0.5f * res) * (bh * bbox_scale / res);
  const int x0 = (int)floorf(sx), y0 = (int)floorf(sy);
  const float fx = sx - x0, fy = sy - y0;
  const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int xx = x0 + (k & 1), yy = y0 + (k >> 1);
      v[k] = ((unsigned)xx < (unsigned)W && (unsigned)yy < (unsigned)H) ? (float)frame[((size_t)yy * W + xx) * 3 + c] : 0.f;
    }
    float p = (1.f - fy) * ((1.f - fx) * v[0] + fx * v[1]) + fy * ((1.f - fx) * v[2] + fx * v[3]);
    p = fminf(fmaxf(rintf(p), 0.f), 255.f);                 // cv2 stores the warped crop as uint8
    out[(((size_t)n * 3 + c) * res + y) * res + x] = (p / 255.f - mean[c]) / stdv[c];
  }
}
}  // namespace

void launch_crop_normalize(const unsigned char* frame, int H, int W, const float* boxes, float bbox_scale, float* out,
                           int N, int res, hipStream_t s) {
  const long n = (long)N * res * res;
  hipLaunchKernelGGL(crop_normalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, frame, H, W, boxes,
                     bbox_scale, out, N, res);
}
