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

// Stage 1 on the fp32 MFMA (round 3; C a multiple of 64).  The pool IS a GEMM over the pixels:
//   acc[part j][channel c] = sum_p exp(heat[p][1 + j] - m_j) * feat[p][c]
// weights (parts x pixels) = A operand, features (pixels x channels) = B operand.  A wave owns every 4th group of 4 pixels of
// the block's pixel range: per step a lane loads ONE heat value per 16-part tile (lane (m, k): part 16 mt + m of pixel 4 s + k),
// takes its exp, and ONE float4 of features per 64-channel group (lane (n, k): channels 4 n .. 4 n + 3 of that pixel - K permuted
// as in the conv kernels: MFMA i of the quad feeds the n-tile of channels {4 n + i}), 8 MFMAs per float4.  The softmax
// denominators ride along in VALU (a lane sums its own weights; the four pixel lanes and the four waves are merged at the end).
// The LDS-broadcast FMA loop of the VALU kernel above (6 ds_read_b128 per feature float) ran at 1.0 TB/s of algorithmic bytes.
__global__ void __launch_bounds__(256)
attn_pool_partial_mfma_kernel(const float* __restrict__ heat, int heat_cs, const float* __restrict__ feat, int C,
                              float* __restrict__ scratch, int H, int W, int nsplit) {
  const int HW = H * W;
  __shared__ float red[4][32];
  __shared__ float mloc[32];
  extern __shared__ float accs[];            // [3 waves][(C/64) * 8 tiles][64 lanes][4]: partial accumulators of waves 1..3
  const int b = blockIdx.x, s = blockIdx.y, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int per = ((HW + nsplit - 1) / nsplit + 3) & ~3;        // pixels per split, a multiple of 4
  const int p0 = s * per, p1 = min(HW, p0 + per);
  const size_t img_heat = (size_t)b * H * (heat_cs >> 4) * (size_t)(W * 16);
  const size_t img_feat = (size_t)b * H * (C >> 4) * (size_t)(W * 16);
  // float offset of (pixel p, 16-channel slice sl) inside an image of a buffer with cs channels
  auto pix = [&](int p, int sl, int cs) {
    const int y = p / W, x = p - y * W;
    return ((size_t)y * (cs >> 4) + sl) * (size_t)(W * 16) + (size_t)x * 16;
  };
  // ---- local max per part: thread = (pixel lane, channel 0..31 of the heat map) ----
  {
    const int ch = tid & 31, pl = tid >> 5;
    float mx = -INFINITY;
    for (int p = p0 + pl; p < p1; p += 8) mx = fmaxf(mx, heat[img_heat + pix(p, ch >> 4, heat_cs) + (ch & 15)]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));                              // the two pixel lanes of a wave
    if (lane < 32) red[wave][lane] = mx;
  }
  __syncthreads();
  if (tid < 32) mloc[tid] = fmaxf(fmaxf(red[0][tid], red[1][tid]), fmaxf(red[2][tid], red[3][tid]));   // channel tid = part tid - 1
  __syncthreads();
  const int m = lane & 15, k = lane >> 4;
  // A operand: m-tile 0 = parts 0..15 (heat channels 1..16), m-tile 1 = parts 16..23 (channels 17..24; rows 8..15 are padding)
  const int hc0 = 1 + m, hc1 = 17 + m;
  const bool a1ok = m < 8;
  const float mx0 = mloc[hc0], mx1 = a1ok ? mloc[hc1] : 0.f;
  const int NG = C >> 6;                                                  // 64-channel groups
  f32x4 acc[2][2][4];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[g][mt][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float l0 = 0.f, l1 = 0.f;
  const int n = m;                                                        // B operand: lane (n, k)
  // 4 steps (16 pixels of this wave) per trip: all 8 heat + 8 feature loads are issued before the first exp / MFMA
  for (int q0 = p0 + 4 * wave; q0 < p1; q0 += 64) {
    float h0[4], h1[4];
    float4 f[4][2];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = q0 + 16 * u + k;
      ok[u] = p < p1;
      const int pc = ok[u] ? p : p1 - 1;
      h0[u] = heat[img_heat + pix(pc, hc0 >> 4, heat_cs) + (hc0 & 15)];
      h1[u] = a1ok ? heat[img_heat + pix(pc, 1, heat_cs) + (hc1 & 15)] : 0.f;
#pragma unroll
      for (int g = 0; g < 2; ++g)
        f[u][g] = (g < NG) ? *reinterpret_cast<const float4*>(feat + img_feat + pix(pc, g * 4 + (n >> 2), C) + (n & 3) * 4)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float w0 = ok[u] ? __expf(h0[u] - mx0) : 0.f;
      const float w1 = (ok[u] && a1ok) ? __expf(h1[u] - mx1) : 0.f;
      l0 += w0; l1 += w1;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        if (g < NG) {
          const float fv[4] = {f[u][g].x, f[u][g].y, f[u][g].z, f[u][g].w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            acc[g][0][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0, fv[i], acc[g][0][i], 0, 0, 0);
            acc[g][1][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1, fv[i], acc[g][1][i], 0, 0, 0);
          }
        }
      }
    }
  }
  // ---- merge: denominators over the 4 pixel lanes and the 4 waves; accumulators of waves 1..3 through LDS ----
  l0 += __shfl_xor(l0, 16); l0 += __shfl_xor(l0, 32);
  l1 += __shfl_xor(l1, 16); l1 += __shfl_xor(l1, 32);
  __syncthreads();                                                        // red[] is free again
  if (lane < 16) { red[wave][lane] = l0; red[wave][16 + lane] = l1; }
  const int ntile = NG * 8;
  if (wave > 0) {
    float4* dstw = reinterpret_cast<float4*>(accs) + (size_t)(wave - 1) * ntile * 64 + lane;
#pragma unroll
    for (int g = 0; g < 2; ++g)
      if (g < NG) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const f32x4 v = acc[g][mt][i];
            dstw[((g * 2 + mt) * 4 + i) * 64] = make_float4(v[0], v[1], v[2], v[3]);
          }
      }
  }
  __syncthreads();
  float* sc = scratch + ((size_t)b * nsplit + s) * (2 * NPART + (size_t)C * NPART);
  if (tid < NPART) {
    sc[tid] = mloc[1 + tid];
    sc[NPART + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
  }
  if (wave == 0) {
    float* accout = sc + 2 * NPART;
    const float4* src = reinterpret_cast<const float4*>(accs) + lane;
    // D[row 4 k + e][column n] of tile (g, mt, i): part 16 mt + 4 k + e, channel 64 g + 4 n + i
#pragma unroll
    for (int g = 0; g < 2; ++g)
      if (g < NG) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            f32x4 v = acc[g][mt][i];
            const int t = (g * 2 + mt) * 4 + i;
#pragma unroll
            for (int w = 0; w < 3; ++w) {
              const float4 o = src[((size_t)w * ntile + t) * 64];
              v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w;
            }
            const int ch = 64 * g + 4 * n + i;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int part = 16 * mt + 4 * k + e;
              if (part < NPART) accout[ch * NPART + part] = v[e];
            }
          }
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
  if ((C == 64 || C == 128) && heat_cs >= 32) {   // the PARE head's pools: fp32-MFMA formulation (two 64-channel accumulator groups at most)
    const size_t lds = (size_t)3 * (C / 64) * 8 * 64 * sizeof(float4);
    hipLaunchKernelGGL(attn_pool_partial_mfma_kernel, dim3(B, ATTN_NSPLIT), dim3(256), lds, s, heat, heat_cs, feat, C,
                       scratch, H, W, ATTN_NSPLIT);
  } else {
    hipLaunchKernelGGL(attn_pool_partial_kernel, dim3(B, ATTN_NSPLIT), dim3(256), 0, s, heat, heat_cs, feat, C,
                       scratch, H, W, ATTN_NSPLIT);
  }
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
