// Head kernels: part-attention pooling (online-softmax weighted pool), per-joint pose linears,
// rot6d -> rotation matrices, small layout helpers.  Wavefront/LDS reductions, no MFMA: these are
// <1 % of the flops and HBM/latency bound.
#include "kernels.h"

namespace {

constexpr int NPART = 24;
constexpr int PTILE = 64;   // pixels per LDS weight tile

// Stage 1: one block per (crop b, pixel split s).  Unnormalised partial pool with a local max.
// scratch layout per (b,s): m[24] | l[24] | acc[C][24]
__global__ void __launch_bounds__(256)
attn_pool_partial_kernel(const float* __restrict__ heat, int heat_cs, const float* __restrict__ feat, int C,
                         float* __restrict__ scratch, int H, int W, int nsplit) {
  const int HW = H * W;
  // L16 float offset of (pixel p of crop b, channel ch) in a buffer with cs channels
  auto at = [&](int b, int p, int ch, int cs) {
    const int y = p / W, x = p - y * W;
    return (((size_t)b * H + y) * (cs >> 4) + (ch >> 4)) * (size_t)(W * 16) + (size_t)x * 16 + (ch & 15);
  };
  __shared__ float red[256 / 64][NPART];
  __shared__ float mloc[NPART];
  __shared__ __attribute__((aligned(16))) float wt[PTILE][NPART];
  const int b = blockIdx.x, s = blockIdx.y, tid = threadIdx.x;
  const int per = (HW + nsplit - 1) / nsplit;
  const int p0 = s * per, p1 = min(HW, p0 + per);
  // ---- local max per part ----
  float mx[NPART];
#pragma unroll
  for (int j = 0; j < NPART; ++j) mx[j] = -INFINITY;
  for (int p = p0 + tid; p < p1; p += 256) {
    // channel 0 = background (pare_head.py:794-796); parts 1..24 straddle the two 16-channel slices
    const float* h0 = heat + at(b, p, 0, heat_cs);
    const float* h1 = heat + at(b, p, 16, heat_cs);
#pragma unroll
    for (int j = 0; j < NPART; ++j) mx[j] = fmaxf(mx[j], (j + 1 < 16) ? h0[j + 1] : h1[j + 1 - 16]);
  }
#pragma unroll
  for (int j = 0; j < NPART; ++j) {
    float v = mx[j];
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    if ((tid & 63) == 0) red[tid >> 6][j] = v;
  }
  __syncthreads();
  if (tid < NPART) mloc[tid] = fmaxf(fmaxf(red[0][tid], red[1][tid]), fmaxf(red[2][tid], red[3][tid]));
  __syncthreads();
  // ---- weighted accumulation ----
  // L16 features: a wave reads 4 neighbouring pixels x one 16-channel slice = 256 contiguous bytes per step.
  // thread = (channel-in-slice tid&15, pixel lane (tid>>4)&3, wave = slice cb, cb+4 for C = 128)
  const int chl = tid & 15, pl = (tid >> 4) & 3, wv = tid >> 6;
  const int nkb = (C / 16 + 3) / 4;        // slices per wave (1 for C = 64, 2 for C = 128)
  float acc[2][NPART];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int j = 0; j < NPART; ++j) acc[k][j] = 0.f;
  float lsum = 0.f;                      // threads < 24 accumulate the softmax denominators
  for (int t0 = p0; t0 < p1; t0 += PTILE) {
    const int np = min(PTILE, p1 - t0);
    float hv[PTILE * NPART / 256];
#pragma unroll
    for (int u = 0; u < PTILE * NPART / 256; ++u) {          // all 6 loads in flight
      const int i = tid + u * 256, pp = i / NPART, j = i - pp * NPART;
      hv[u] = heat[at(b, t0 + min(pp, np - 1), 1 + j, heat_cs)];
    }
#pragma unroll
    for (int u = 0; u < PTILE * NPART / 256; ++u) {
      const int i = tid + u * 256, pp = i / NPART, j = i - pp * NPART;
      wt[pp][j] = (pp < np) ? __expf(hv[u] - mloc[j]) : 0.f;
    }
    __syncthreads();
    if (tid < NPART)
      for (int pp = 0; pp < np; ++pp) lsum += wt[pp][tid];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int cb = wv + k * 4;
      if (k < nkb && cb * 16 < C)
        for (int pp0 = pl; pp0 < np; pp0 += 16) {
          // 4 feature loads in flight per thread (the loop is latency-bound: ~1 wave per SIMD); rows >= np of wt are 0
          float f[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) f[u] = feat[at(b, t0 + min(pp0 + 4 * u, np - 1), cb * 16 + chl, C)];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float4* w4 = reinterpret_cast<const float4*>(&wt[min(pp0 + 4 * u, PTILE - 1)][0]);   // broadcast b128 reads
            const float m = (pp0 + 4 * u < np) ? f[u] : 0.f;
#pragma unroll
            for (int j4 = 0; j4 < NPART / 4; ++j4) {
              const float4 w = w4[j4];
              acc[k][j4 * 4] = fmaf(w.x, m, acc[k][j4 * 4]); acc[k][j4 * 4 + 1] = fmaf(w.y, m, acc[k][j4 * 4 + 1]);
              acc[k][j4 * 4 + 2] = fmaf(w.z, m, acc[k][j4 * 4 + 2]); acc[k][j4 * 4 + 3] = fmaf(w.w, m, acc[k][j4 * 4 + 3]);
            }
          }
        }
    }
    __syncthreads();
  }
  float* sc = scratch + ((size_t)b * nsplit + s) * (2 * NPART + (size_t)C * NPART);
  if (tid < NPART) { sc[tid] = mloc[tid]; sc[NPART + tid] = lsum; }
  float* accout = sc + 2 * NPART;
  // the 4 pixel lanes of a channel sit 16 lanes apart in the same wave
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int cb = wv + k * 4;
#pragma unroll
    for (int j = 0; j < NPART; ++j) {
      float v = acc[k][j];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (pl == 0 && k < nkb && cb * 16 < C) accout[(cb * 16 + chl) * NPART + j] = v;
    }
  }
}

// Stage 2: combine splits.  thread = (b, c, j)
__global__ void attn_pool_combine_kernel(const float* __restrict__ scratch, float* __restrict__ dst, int dst_stride,
                                         int B, int C, int nsplit) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C * NPART) return;
  const int j = i % NPART, c = (i / NPART) % C, b = i / (NPART * C);
  const size_t per = 2 * NPART + (size_t)C * NPART;
  const float* sb = scratch + (size_t)b * nsplit * per;
  float M = -INFINITY;
  for (int s = 0; s < nsplit; ++s) M = fmaxf(M, sb[s * per + j]);
  float num = 0.f, den = 0.f;
  for (int s = 0; s < nsplit; ++s) {
    const float e = __expf(sb[s * per + j] - M);
    den += sb[s * per + NPART + j] * e;
    num += sb[s * per + 2 * NPART + c * NPART + j] * e;
  }
  dst[(size_t)b * dst_stride + c * NPART + j] = num / den;
}

__global__ void lc2d_pose_kernel(const float* __restrict__ x, int x_stride, const float* __restrict__ w,
                                 float* __restrict__ pose6d, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * 144) return;
  const int o = i % 6, j = (i / 6) % 24, b = i / 144;
  const float* xb = x + (size_t)b * x_stride + j;
  const float* wo = w + (size_t)o * 128 * 24 + j;
  float acc = 0.f;
  for (int c = 0; c < 128; ++c) acc = fmaf(xb[c * 24], wo[c * 24], acc);
  pose6d[(size_t)b * 144 + j * 6 + o] = acc;
}

__global__ void rot6d_kernel(const float* __restrict__ in, int in_stride, float* dst0, int stride0, float* dst1,
                             int stride1, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * 24) return;
  const int j = i % 24, b = i / 24;
  const float* x = in + (size_t)b * in_stride + j * 6;
  const float a1[3] = {x[0], x[2], x[4]}, a2[3] = {x[1], x[3], x[5]};
  const float n1 = fmaxf(sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]), 1e-12f);
  const float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
  const float d = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
  float u[3] = {a2[0] - d * b1[0], a2[1] - d * b1[1], a2[2] - d * b1[2]};
  const float n2 = fmaxf(sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), 1e-12f);
  const float b2[3] = {u[0] / n2, u[1] / n2, u[2] / n2};
  const float b3[3] = {b1[1] * b2[2] - b1[2] * b2[1], b1[2] * b2[0] - b1[0] * b2[2], b1[0] * b2[1] - b1[1] * b2[0]};
  float R[9];
  for (int r = 0; r < 3; ++r) { R[r * 3 + 0] = b1[r]; R[r * 3 + 1] = b2[r]; R[r * 3 + 2] = b3[r]; }
  if (dst0) for (int k = 0; k < 9; ++k) dst0[(size_t)b * stride0 + j * 9 + k] = R[k];
  if (dst1) for (int k = 0; k < 9; ++k) dst1[(size_t)b * stride1 + j * 9 + k] = R[k];
}

__global__ void copy_rows_kernel(const float* __restrict__ src, int src_stride, float* __restrict__ dst,
                                 int dst_stride, int n, int B, int bcast) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * n) return;
  const int k = (int)(i % n), b = (int)(i / n);
  dst[(size_t)b * dst_stride + k] = src[(bcast ? 0 : (size_t)b * src_stride) + k];
}

__global__ void nhwc_to_nchw_kernel(const float* __restrict__ in, int cs, float* __restrict__ out, int B, int H, int W,
                                    int C) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int HW = H * W;
  if (i >= (long)B * C * HW) return;
  const int p = (int)(i % HW), c = (int)((i / HW) % C), b = (int)(i / ((long)HW * C));
  const int y = p / W, x = p - y * W;
  out[i] = in[(((size_t)b * H + y) * (cs >> 4) + (c >> 4)) * (size_t)(W * 16) + (size_t)x * 16 + (c & 15)];
}

inline int nblk(long n, int t) { return (int)((n + t - 1) / t); }
constexpr int ATTN_NSPLIT = 14;

}  // namespace

size_t part_attention_scratch_floats(int B, int C) {
  return (size_t)B * ATTN_NSPLIT * (2 * NPART + (size_t)C * NPART);
}

void launch_part_attention_pool_ws(const float* heat, int heat_cs, const float* feat, int C, float* dst,
                                   int dst_stride, int B, int H, int W, float* scratch, hipStream_t s) {
  hipLaunchKernelGGL(attn_pool_partial_kernel, dim3(B, ATTN_NSPLIT), dim3(256), 0, s, heat, heat_cs, feat, C,
                     scratch, H, W, ATTN_NSPLIT);
  hipLaunchKernelGGL(attn_pool_combine_kernel, dim3(nblk((long)B * C * NPART, 256)), dim3(256), 0, s, scratch, dst,
                     dst_stride, B, C, ATTN_NSPLIT);
}

void launch_lc2d_pose(const float* x, int x_stride, const float* w, float* pose6d, int B, hipStream_t s) {
  hipLaunchKernelGGL(lc2d_pose_kernel, dim3(nblk((long)B * 144, 128)), dim3(128), 0, s, x, x_stride, w, pose6d, B);
}

void launch_rot6d(const float* in, int in_stride, float* dst0, int stride0, float* dst1, int stride1, int B,
                  hipStream_t s) {
  hipLaunchKernelGGL(rot6d_kernel, dim3(nblk((long)B * 24, 128)), dim3(128), 0, s, in, in_stride, dst0, stride0,
                     dst1, stride1, B);
}

void launch_copy_rows(const float* src, int src_stride, float* dst, int dst_stride, int n, int B, hipStream_t s) {
  hipLaunchKernelGGL(copy_rows_kernel, dim3(nblk((long)B * n, 256)), dim3(256), 0, s, src, src_stride, dst,
                     dst_stride, n, B, 0);
}

void launch_broadcast_rows(const float* src, float* dst, int dst_stride, int n, int B, hipStream_t s) {
  hipLaunchKernelGGL(copy_rows_kernel, dim3(nblk((long)B * n, 256)), dim3(256), 0, s, src, 0, dst, dst_stride, n, B,
                     1);
}

void launch_nhwc_to_nchw(const float* in, int cs, float* out, int B, int H, int W, int C, hipStream_t s) {
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(nblk((long)B * C * H * W, 256)), dim3(256), 0, s, in, cs, out, B, H, W, C);
}
