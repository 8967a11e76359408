// Stem conv (Cin = 3, stride 2, 7x7 pad 3: resnet.py:203-205; 3x3 pad 1: hrnet.py:467-469) + BN + ReLU on the fp32 MFMA (round 5).
//
// The VALU kernels of kernels_misc.hip do the 7x7 stem of 64 crops (15.1 GFLOP) in 214 us: v_pk_fma_f32 issues at 8 clk per wave
// on this chip, so 4 pixels x 16 channels per thread is instruction-issue bound at ~70 TFLOP/s, and the 3x3 stem (K = 27) spends
// its 114 us on addressing.  Here the conv is the implicit GEMM D[co][pix] += W[co][k] X[k][pix], k = (r, s, c), K = 147 | 27 padded
// to a multiple of 4, on v_mfma_f32_16x16x4_f32 with the operand roles of the other conv kernels (weights = A, pixels = B: a lane
// ends up with 4 consecutive channels of one pixel -> 16-byte stores in the L16 layout).
//
//   * a wave owns one output ROW: MT = Wo / 16 pixel sub-tiles x all 64 channels = 4 MT accumulators (112 VGPRs at Wo = 112);
//     a block = 4 waves = a CHUNK of 4 consecutive rows, and walks `nchunk` chunks of one image.  At 64 crops that is 7 chunks =
//     a quarter image per block and 256 blocks: one wave per SIMD, exactly 49 sub-tile rows of work for each of the 1024 SIMDs.
//   * the input rows under a chunk (2 * 4 + KS - 2 rows x 3 channels, zero padding written out: no masks in the K loop) live in
//     LDS, double-buffered: a thread owns one patch column, requests the next chunk's values before the K loop and writes them to
//     the other buffer after it.
//   * per K step (4 taps): the B operand of sub-tile m is ONE ds_read_b32 at lane base + ktab[k] + 128 m bytes (the stride-2
//     gather: lane (idx, g) reads column 2 idx + s of row r, channel c of its tap k = 4 step + g; ktab = byte offset of a tap
//     inside the patch, held in registers), the A operand of n-tile n one conflict-free ds_read_b32: MT + 4 LDS reads for 4 MT
//     MFMAs, requested one K step ahead.
#include "kernels.h"

namespace {

constexpr int STEM_RC = 4;      // rows per chunk = waves per block

template <int KS, int MT>
__global__ void __launch_bounds__(STEM_RC * 64)
stem_mfma_kernel(const float* __restrict__ img, const float* __restrict__ w, const float* __restrict__ shift,
                 float* __restrict__ out, int H, int W, int Ho, int Wo, int nchunk, int blocks_per_img) {
  constexpr int PAD = (KS - 1) / 2;
  constexpr int K = KS * KS * 3;
  constexpr int NSTEP = (K + 3) / 4;
  constexpr int IR = 2 * STEM_RC + KS - 2;          // input rows under a chunk
  constexpr int NQ = 3 * IR;                        // patch rows (channel-major)
  extern __shared__ float lds[];
  const int pitch = W + 2 * PAD;                    // floats per patch row
  float* wl = lds;                                  // A fragments [NSTEP][4 n][64 lanes]
  int* ktab = reinterpret_cast<int*>(wl + NSTEP * 256);     // [NSTEP * 4] byte offset of tap k inside the patch
  float* patch = wl + NSTEP * 256 + ((NSTEP * 4 + 63) & ~63);   // [2][NQ][pitch]
  const int patch_sz = NQ * pitch;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int idx = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / blocks_per_img;
  const int row_base = (blockIdx.x - b * blocks_per_img) * (STEM_RC * nchunk);
  const float* ib = img + (size_t)b * 3 * H * W;

  {   // A fragments: thread (n, l) of the 256 owns element (n, l) of every step - NSTEP independent loads in flight (a loop with
      // `k < K ? w[..] : 0` is NSTEP dependent L2 round trips at the start of every block)
    static_assert(STEM_RC * 64 == 256, "one fragment element per thread and step");
    const int n = (tid >> 6) & 3, l = tid & 63;
    float wv[NSTEP];
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) wv[st] = w[min(4 * st + (l >> 4), K - 1) * 64 + 16 * n + (l & 15)];
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) wl[st * 256 + tid] = (4 * st + (l >> 4) < K) ? wv[st] : 0.f;
  }
  for (int k = tid; k < NSTEP * 4; k += STEM_RC * 64) {
    int off = 0;                                    // padded taps: weight 0, any valid address
    if (k < K) {
      const int rs = k / 3, c = k - rs * 3, r = rs / KS, s = rs - r * KS;
      off = ((c * IR + r) * pitch + s) * 4;
    }
    ktab[k] = off;
  }
  // one patch column per thread (pitch <= 256): values of chunk ci, zero outside the image
  float stage[NQ];
  auto fetch = [&](int ci) {
    const int iy0 = 2 * (row_base + STEM_RC * ci) - PAD;
    const int ix = tid - PAD;
    const bool colok = tid < pitch && (unsigned)ix < (unsigned)W;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int c = q / IR, iy = iy0 + (q - c * IR);
      stage[q] = (colok && (unsigned)iy < (unsigned)H) ? ib[((size_t)c * H + iy) * W + ix] : 0.f;
    }
  };
  auto commit = [&](int buf) {
    if (tid < pitch) {
      float* dst = patch + buf * patch_sz + tid;
#pragma unroll
      for (int q = 0; q < NQ; ++q) dst[q * pitch] = stage[q];
    }
  };
  fetch(0);
  commit(0);
  __syncthreads();

  float4 sh[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) sh[n] = *reinterpret_cast<const float4*>(shift + 16 * n + 4 * g);
  const char* wlb = reinterpret_cast<const char*>(wl) + lane * 4;
  // this lane's tap offsets (k = 4 step + g) stay in registers for the whole kernel: one wave per SIMD owns 512 registers, the
  // accumulators live in the AGPR half
  int ko[NSTEP];
#pragma unroll
  for (int st = 0; st < NSTEP; ++st) ko[st] = ktab[st * 4 + g];

  for (int ci = 0; ci < nchunk; ++ci) {
    const int buf = ci & 1;
    const bool more = ci + 1 < nchunk;
    if (more) fetch(ci + 1);                        // travels under this chunk's K loop
    const char* pb = reinterpret_cast<const char*>(patch + buf * patch_sz) + ((2 * wave) * pitch + 2 * idx) * 4;
    f32x4 acc[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // operands one K step ahead (fenced: hipcc otherwise sinks every ds_read to just before its first MFMA and the wave - alone
    // on its SIMD - waits out the LDS latency once per 8 MFMAs: 183 us for the 7x7 stem)
    float a[2][4], bv[2][MT];
    auto ldops = [&](int st, float (&aa)[4], float (&bb)[MT]) {
#pragma unroll
      for (int n = 0; n < 4; ++n) aa[n] = *reinterpret_cast<const float*>(wlb + (st * 4 + n) * 256);
      const char* q = pb + ko[st];
#pragma unroll
      for (int m = 0; m < MT; ++m) bb[m] = *reinterpret_cast<const float*>(q + 128 * m);
    };
    ldops(0, a[0], bv[0]);
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
      if (st + 1 < NSTEP) ldops(st + 1, a[(st + 1) & 1], bv[(st + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[st & 1][n], bv[st & 1][m], acc[m][n], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    const int r = row_base + STEM_RC * ci + wave;
    if (r < Ho) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        float4* o = reinterpret_cast<float4*>(out) + L16_F4((size_t)b * Ho + r, 16 * m + idx, g, Wo, 4);
#pragma unroll
        for (int n = 0; n < 4; ++n) {               // channels 16 n + 4 g .. + 3 = quad g of slice n: slices are Wo * 4 float4 apart
          const f32x4 v = acc[m][n];
          o[(size_t)n * Wo * 4] = make_float4(fmaxf(v[0] + sh[n].x, 0.f), fmaxf(v[1] + sh[n].y, 0.f),
                                              fmaxf(v[2] + sh[n].z, 0.f), fmaxf(v[3] + sh[n].w, 0.f));
        }
      }
    }
    if (more) commit(buf ^ 1);                      // the other buffer was last read in chunk ci - 1 (barrier below)
    __syncthreads();
  }
}

template <int KS>
size_t stem_lds_bytes(int W) {
  constexpr int K = KS * KS * 3, NSTEP = (K + 3) / 4, IR = 2 * STEM_RC + KS - 2;
  return ((size_t)NSTEP * 256 + ((NSTEP * 4 + 63) & ~63) + 2 * (size_t)3 * IR * (W + KS - 1)) * sizeof(float);
}

// chunks per block: few enough that the grid covers the chip, as many as that allows (fewer partial rounds, weights staged once)
int stem_nchunk(int B, int Ho) {
  const int per_img = (Ho + STEM_RC - 1) / STEM_RC;
  int best = 1;
  double best_cost = 1e30;
  for (int nc = 1; nc <= per_img; ++nc) {
    const int blocks = B * ((per_img + nc - 1) / nc);
    const double rounds = (double)((blocks + 255) / 256);
    const double cost = rounds * (nc + 0.35);       // 0.35 chunk-times of per-block start-up
    if (cost < best_cost - 1e-9) { best_cost = cost; best = nc; }
  }
  return best;
}

template <int KS>
bool launch_stem_mfma_t(const float* img, const float* w, const float* shift, float* out, int B, int H, int W, int Ho, int Wo,
                        hipStream_t s) {
  const size_t lds = stem_lds_bytes<KS>(W);
  if (Wo != 112 || W + KS - 1 > STEM_RC * 64 || lds > 160 * 1024) return false;
  static thread_local bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(stem_mfma_kernel<KS, 7>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess) return false;
    configured = true;
  }
  const int nchunk = stem_nchunk(B, Ho);
  const int per_img = ((Ho + STEM_RC - 1) / STEM_RC + nchunk - 1) / nchunk;
  hipLaunchKernelGGL((stem_mfma_kernel<KS, 7>), dim3(B * per_img), dim3(STEM_RC * 64), lds, s, img, w, shift, out, H, W, Ho, Wo,
                     nchunk, per_img);
  return true;
}

}  // namespace

// true = launched (224 x 224 crops: Wo = 112); false = shape not covered, the caller falls back to the VALU kernels
bool launch_stem_conv_mfma(const float* img, const float* w, const float* shift, float* out, int B, int H, int W, int ks,
                           hipStream_t s) {
  const int pad = (ks - 1) / 2;
  const int Ho = (H + 2 * pad - ks) / 2 + 1, Wo = (W + 2 * pad - ks) / 2 + 1;
  if (ks == 7) return launch_stem_mfma_t<7>(img, w, shift, out, B, H, W, Ho, Wo, s);
  if (ks == 3) return launch_stem_mfma_t<3>(img, w, shift, out, B, H, W, Ho, Wo, s);
  return false;
}
