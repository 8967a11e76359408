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
                      int ks, hipStream_t s, int use_mfma) {
  if (use_mfma && launch_stem_conv_mfma(img, w, shift, out, B, H, W, ks, s)) return;   // 224 x 224 crops: implicit GEMM on the MFMA (stem_mfma.hip)
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
// Replaces the per-crop CPU loop  cv2.getAffineTransform -> cv2.warpAffine(INTER_LINEAR, BORDER_CONSTANT) on uint8 ->
// ToTensor -> Normalize + per-crop H2D copy of pocolib/core/tester.py:182-203 / utils/vibe_image_utils.py:58-107,
// 233-266,343-351.  BYTE-exact with OpenCV 4.5.5's algorithm for that call (restated step by step in oracle/crop_np.py,
// which documents every constant): the 6x6 LU solve and the matrix inversion in IEEE double WITHOUT fma contraction,
// 10-bit fixed-point coordinates with round_delta 16, 5-bit sub-pixel fractions, int16 weights summing to 2^15 and a
// rounded 15-bit shift.  A block owns CROP_ROWS rows of one crop; its first wave solves the box's affine system once
// (~1 us), every thread then produces pixels x = t, t + 256, ... of those rows for the three channels (stores
// coalesced along x in NCHW; the uint8 frame is read through L2, 12 bytes per output pixel).
namespace {
constexpr int CROP_ROWS = 28;     // 8 blocks per 224-row crop: the ~3 us affine solve is paid once per 28 rows

// cv::hal::LU64f (LUImpl<double>, opencv/modules/core/src/matrix_decomp.cpp) for the 6x6 system of getAffineTransform.
__device__ __forceinline__ void lu_solve6(double (&A)[6][6], double (&b)[6]) {
#pragma clang fp contract(off)
  // fully unrolled with static indices (the pivot row is swapped in through selects) so that A lives in registers
  bool singular = false;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    int k = i;
    double best = fabs(A[i][i]);
#pragma unroll
    for (int j = i + 1; j < 6; ++j)
      if (fabs(A[j][i]) > best) { best = fabs(A[j][i]); k = j; }
    singular = singular || best < 2.220446049250313e-16 * 100;      // cv::solve returns false and leaves zeros
#pragma unroll
    for (int j = i + 1; j < 6; ++j) {
      const bool sw = k == j;
#pragma unroll
      for (int c = i; c < 6; ++c) {
        const double t = A[i][c];
        A[i][c] = sw ? A[j][c] : t;
        A[j][c] = sw ? t : A[j][c];
      }
      const double t = b[i];
      b[i] = sw ? b[j] : t;
      b[j] = sw ? t : b[j];
    }
    const double d = -1.0 / A[i][i];
#pragma unroll
    for (int j = i + 1; j < 6; ++j) {
      const double alpha = A[j][i] * d;
#pragma unroll
      for (int c = i + 1; c < 6; ++c) A[j][c] = A[j][c] + alpha * A[i][c];
      b[j] = b[j] + alpha * b[i];
    }
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double s = b[i];
#pragma unroll
    for (int c = i + 1; c < 6; ++c) s = s - A[i][c] * b[c];
    b[i] = s / A[i][i];
  }
  if (singular) {
#pragma unroll
    for (int j = 0; j < 6; ++j) b[j] = 0.0;
  }
}

// gen_trans_from_patch_cv(rot = 0) -> getAffineTransform(src, dst) -> the inversion cv::warpAffine applies: Mi[6].
__device__ __forceinline__ void crop_inverse_affine(double cx, double cy, double bw, double bh, double scale, int res, double (&Mi)[6]) {
#pragma clang fp contract(off)
  const float down = (float)(bh * scale * 0.5), right = (float)(bw * scale * 0.5);
  const float sx[3] = {(float)cx, (float)(cx + (double)0.0f), (float)(cx + (double)right)};
  const float sy[3] = {(float)cy, (float)(cy + (double)down), (float)(cy + (double)0.0f)};
  const float half = (float)(res * 0.5);
  const float dx[3] = {half, half, half + half}, dy[3] = {half, half + half, half};
  double A[6][6], b[6];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) A[i][j] = 0.0;
  for (int i = 0; i < 3; ++i) {
    A[2 * i][0] = A[2 * i + 1][3] = sx[i];
    A[2 * i][1] = A[2 * i + 1][4] = sy[i];
    A[2 * i][2] = A[2 * i + 1][5] = 1.0;
    b[2 * i] = dx[i];
    b[2 * i + 1] = dy[i];
  }
  lu_solve6(A, b);
  double D = b[0] * b[4] - b[1] * b[3];
  D = D != 0.0 ? 1.0 / D : 0.0;
  const double A11 = b[4] * D, A22 = b[0] * D;
  Mi[0] = A11;
  Mi[1] = b[1] * -D;
  Mi[3] = b[3] * -D;
  Mi[4] = A22;
  Mi[2] = -Mi[0] * b[2] - Mi[1] * b[5];
  Mi[5] = -Mi[3] * b[2] - Mi[4] * b[5];
}

__device__ __forceinline__ int cv_round(double v) { return __double2int_rn(v); }      // cvRound: round half to even

// frames != nullptr (video / streaming batches): crop n is cut from frames[frame_idx[n]] - one launch for the crops of many frames
template <typename BoxT>
__global__ void __launch_bounds__(256)
crop_normalize_kernel(const unsigned char* __restrict__ frame, const unsigned char* const* __restrict__ frames,
                      const int* __restrict__ frame_idx, int nframes, int H, int W, const BoxT* __restrict__ boxes,
                      double bbox_scale, float* __restrict__ out, int res) {
#pragma clang fp contract(off)
  __shared__ double sM[6];
  const int n = blockIdx.y, row0 = blockIdx.x * CROP_ROWS;
  // (an index outside the table is clamped into it: a caller's bad index reads the wrong frame, never a wild device pointer - ADVICE r4)
  if (frames) frame = frames[min(max(frame_idx[n], 0), nframes - 1)];
  if (threadIdx.x < 64) {                    // the whole first wave solves the same system (no divergence, no broadcast)
    double Mi[6];
    crop_inverse_affine((double)boxes[n * 4], (double)boxes[n * 4 + 1], (double)boxes[n * 4 + 2], (double)boxes[n * 4 + 3],
                        bbox_scale, res, Mi);
    if (threadIdx.x < 6) sM[threadIdx.x] = Mi[threadIdx.x];
  }
  __syncthreads();
  const double M0 = sM[0], M1 = sM[1], M2 = sM[2], M3 = sM[3], M4 = sM[4], M5 = sM[5];
  const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
  const int rows = min(CROP_ROWS, res - row0);
  for (int i = threadIdx.x; i < rows * res; i += 256) {
    const int x = i % res, y = row0 + i / res;
    const int adelta = cv_round(M0 * x * 1024.0), bdelta = cv_round(M3 * x * 1024.0);
    const int X0 = cv_round((M1 * y + M2) * 1024.0) + 16, Y0 = cv_round((M4 * y + M5) * 1024.0) + 16;
    const int X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
    const int sx = max(-32768, min(32767, X >> 5)), sy = max(-32768, min(32767, Y >> 5));
    const int fx = X & 31, fy = Y & 31;
    // BilinearTab_i: saturate_short((1-fy|fy)/32 * (1-fx|fx)/32 * 2^15) = exact integers; (0,0) saturates to 32767 and the
    // table's sum correction puts the missing 1 on the last weight (tests/test_host_cpu.py checks this formula == the table)
    int w00 = (32 - fy) * (32 - fx) * 32, w01 = (32 - fy) * fx * 32, w10 = fy * (32 - fx) * 32, w11 = fy * fx * 32;
    if ((fx | fy) == 0) { w00 = 32767; w11 = 1; }
    float* o = out + (((size_t)n * 3) * res + y) * res + x;
    const size_t cs = (size_t)res * res;
    if (sx >= 0 && sy >= 0 && sx <= W - 3 && sy + 1 < H) {
      // interior (the common case): the 2 x 2 x RGB neighbourhood is bytes 0..5 of two 8-byte windows (unaligned 64-bit
      // loads: 2 memory instructions per pixel instead of 12 byte loads; the window never leaves the frame since sx <= W - 3)
      typedef unsigned long long __attribute__((aligned(1))) u64u;
      const unsigned char* p00 = frame + ((size_t)sy * W + sx) * 3;
      const unsigned long long r0 = *reinterpret_cast<const u64u*>(p00);
      const unsigned long long r1 = *reinterpret_cast<const u64u*>(p00 + (size_t)W * 3);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int v00 = (int)((r0 >> (8 * c)) & 255), v01 = (int)((r0 >> (8 * (3 + c))) & 255);
        const int v10 = (int)((r1 >> (8 * c)) & 255), v11 = (int)((r1 >> (8 * (3 + c))) & 255);
        const int p = min(255, max(0, (w00 * v00 + w01 * v01 + w10 * v10 + w11 * v11 + (1 << 14)) >> 15));
        // ToTensor (uint8 -> float32 / 255) -> Normalize ((t - mean) / std), float32 like torchvision on the CPU
        o[c * cs] = ((float)p / 255.f - mean[c]) / stdv[c];
      }
    } else {
      const bool x0ok = (unsigned)sx < (unsigned)W, x1ok = (unsigned)(sx + 1) < (unsigned)W;
      const bool y0ok = (unsigned)sy < (unsigned)H, y1ok = (unsigned)(sy + 1) < (unsigned)H;
      const unsigned char* p00 = frame + ((long)sy * W + sx) * 3;      // only dereferenced where the *ok flags allow
      const unsigned char* p10 = p00 + (long)W * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int v00 = (x0ok && y0ok) ? p00[c] : 0, v01 = (x1ok && y0ok) ? p00[3 + c] : 0;
        const int v10 = (x0ok && y1ok) ? p10[c] : 0, v11 = (x1ok && y1ok) ? p10[3 + c] : 0;
        const int p = min(255, max(0, (w00 * v00 + w01 * v01 + w10 * v10 + w11 * v11 + (1 << 14)) >> 15));
        o[c * cs] = ((float)p / 255.f - mean[c]) / stdv[c];
      }
    }
  }
}
}  // namespace

template <typename BoxT>
static void launch_crop_t(const unsigned char* frame, int H, int W, const BoxT* boxes, double bbox_scale, float* out, int N,
                          int res, hipStream_t s) {
  for (int n0 = 0; n0 < N; n0 += 65535) {             // gridDim.y limit
    const int nn = N - n0 < 65535 ? N - n0 : 65535;
    hipLaunchKernelGGL(crop_normalize_kernel<BoxT>, dim3((res + CROP_ROWS - 1) / CROP_ROWS, nn), dim3(256), 0, s, frame,
                       (const unsigned char* const*)nullptr, (const int*)nullptr, 0, H, W,
                       boxes + (size_t)n0 * 4, bbox_scale, out + (size_t)n0 * 3 * res * res, res);
  }
}

void launch_crop_normalize_multi(const unsigned char* const* frames, int nframes, const int* frame_idx, int H, int W, const float* boxes,
                                 double bbox_scale, float* out, int N, int res, hipStream_t s) {
  for (int n0 = 0; n0 < N; n0 += 65535) {
    const int nn = N - n0 < 65535 ? N - n0 : 65535;
    hipLaunchKernelGGL(crop_normalize_kernel<float>, dim3((res + CROP_ROWS - 1) / CROP_ROWS, nn), dim3(256), 0, s,
                       (const unsigned char*)nullptr, frames, frame_idx + n0, nframes, H, W, boxes + (size_t)n0 * 4, bbox_scale,
                       out + (size_t)n0 * 3 * res * res, res);
  }
}

// ---- packed per-crop record (poco_outputs_t.record) -------------------------------------------------------------------------
// [rotmat 216 | betas 10 | cam 3 | var_pose 24 (raw) | confidence 1]; confidence = get_kinematic_uncert (poco_utils.py:21-25: children
// in the order 1..23 add their parent's accumulated value; get_smpl_skeleton() is the fixed SMPL tree below) -> get_global_uncert
// (poco_utils.py:50-60: root above the threshold -> every joint 1; cliff: root value, pare: mean over the 24 joints) -> clip to
// [0, 0.99] (tester.py:245).  One block per crop; thread 0 runs the 23 dependent adds in exactly that order.
namespace {
constexpr int SMPL_PARENT[24] = {-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21};

__global__ void __launch_bounds__(256)
pack_record_kernel(const float* __restrict__ rot, int rot_stride, const float* __restrict__ betas, int betas_stride,
                   const float* __restrict__ cam, int cam_stride, const float* __restrict__ var, int var_stride,
                   float* __restrict__ rec, int cliff, int kinematic, float thr) {
  const int b = blockIdx.x, t = threadIdx.x;
  float* r = rec + (size_t)b * 254;
  if (t < 216) r[t] = rot[(size_t)b * rot_stride + t];
  else if (t < 226) r[t] = betas[(size_t)b * betas_stride + (t - 216)];
  else if (t < 229) r[t] = cam[(size_t)b * cam_stride + (t - 226)];
  else if (t < 253) r[t] = var[(size_t)b * var_stride + (t - 229)];
  else if (t == 253) {
    float v[24];
#pragma unroll
    for (int j = 0; j < 24; ++j) v[j] = var[(size_t)b * var_stride + j];
    if (kinematic) {
#pragma unroll
      for (int j = 1; j < 24; ++j) v[j] += v[SMPL_PARENT[j]];
    }
    const bool hot = v[0] > (cliff ? 2.f * thr : thr);
    float g;
    if (cliff) g = hot ? 1.f : v[0];
    else {
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 24; ++j) sum += hot ? 1.f : v[j];
      g = sum / 24.f;
    }
    r[253] = fminf(fmaxf(g, 0.f), 0.99f);
  }
}
}  // namespace

void launch_pack_record(const float* rot, int rot_stride, const float* betas, int betas_stride, const float* cam, int cam_stride,
                        const float* var, int var_stride, float* rec, int cliff, int kinematic, float thr, int B, hipStream_t s) {
  hipLaunchKernelGGL(pack_record_kernel, dim3(B), dim3(256), 0, s, rot, rot_stride, betas, betas_stride, cam, cam_stride, var,
                     var_stride, rec, cliff, kinematic, thr);
}

void launch_crop_normalize(const unsigned char* frame, int H, int W, const float* boxes, double bbox_scale, float* out,
                           int N, int res, hipStream_t s) {
  launch_crop_t<float>(frame, H, W, boxes, bbox_scale, out, N, res, s);
}

void launch_crop_normalize_f64(const unsigned char* frame, int H, int W, const double* boxes, double bbox_scale, float* out,
                               int N, int res, hipStream_t s) {
  launch_crop_t<double>(frame, H, W, boxes, bbox_scale, out, N, res, s);
}
