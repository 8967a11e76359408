// Conditional RealNVP coupling layers (pocolib/models/layers/real_nvp.py:25-65 with the s/t MLPs of
// pocolib/models/head/nf_head.py:13-17) as a wavefront kernel.
//
// One 64-lane wave owns ROWS rows; lane h is hidden unit h of the 64-wide MLPs, so a weight column
// [i][0..63] is one coalesced 256 B read shared by all ROWS rows (weights are pre-transposed to
// input-major at load time).  The 9-dim state, the context rows and the hidden activations live in
// LDS; nothing but x/ctx is read from and log_p (or x) written to HBM.  Latency/L2-bound, <1 % of
// the model's flops; not on the demo path (nf_head.py:129-136 returns log_phi=None at inference).
#include "kernels.h"

namespace {

constexpr int ROWS = 8;
constexpr int D = 9;
constexpr int HID = 64;

__device__ __forceinline__ float leaky(float v) { return v > 0.f ? v : 0.01f * v; }

__global__ void __launch_bounds__(64)
realnvp_kernel(FlowDev f, const float* __restrict__ x, const float* __restrict__ ctx, float* __restrict__ out,
               int N, int forward) {
  extern __shared__ float lds[];
  float* cs = lds;                       // [ROWS][ctx]
  float* z = cs + ROWS * f.ctx;          // [ROWS][D]
  float* zm = z + ROWS * D;              // [ROWS][D] masked state
  float* h = zm + ROWS * D;              // [ROWS][HID]
  float* st = h + ROWS * HID;            // [2][ROWS][D]  s and t outputs
  float* ld = st + 2 * ROWS * D;         // [ROWS] log-det
  const int lane = threadIdx.x;
  const int r0 = blockIdx.x * ROWS;
  const int nr = min(ROWS, N - r0);
  for (int i = lane; i < ROWS * f.ctx; i += 64) {
    const int r = i / f.ctx, k = i % f.ctx;
    cs[i] = (r < nr) ? ctx[(size_t)(r0 + r) * f.ctx + k] : 0.f;
  }
  for (int i = lane; i < ROWS * D; i += 64) z[i] = (i / D < nr) ? x[(size_t)r0 * D + i] : 0.f;
  if (lane < ROWS) ld[lane] = 0.f;
  __syncthreads();
  const int K0 = D + f.ctx;
  for (int step = 0; step < f.L; ++step) {
    const int li = forward ? step : f.L - 1 - step;
    const float* mask = f.mask + li * D;
    for (int i = lane; i < ROWS * D; i += 64) zm[i] = z[i] * mask[i % D];
    __syncthreads();
    for (int net = 0; net < 2; ++net) {       // 0 = s (tanh), 1 = t
      const float* w0 = f.w0t[net] + (size_t)li * K0 * HID;
      const float* w1 = f.w1t[net] + (size_t)li * HID * HID;
      const float* w2 = f.w2[net] + (size_t)li * D * HID;
      float a[ROWS];
      const float b0 = f.b0[net][li * HID + lane];
#pragma unroll
      for (int r = 0; r < ROWS; ++r) a[r] = b0;
      for (int i = 0; i < D; ++i) {
        const float w = w0[i * HID + lane];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) a[r] = fmaf(w, zm[r * D + i], a[r]);
      }
#pragma unroll 4
      for (int i = 0; i < f.ctx; ++i) {
        const float w = w0[(D + i) * HID + lane];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) a[r] = fmaf(w, cs[r * f.ctx + i], a[r]);
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < ROWS; ++r) h[r * HID + lane] = leaky(a[r]);
      __syncthreads();
      const float b1 = f.b1[net][li * HID + lane];
#pragma unroll
      for (int r = 0; r < ROWS; ++r) a[r] = b1;
#pragma unroll 4
      for (int k = 0; k < HID; ++k) {
        const float w = w1[k * HID + lane];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) a[r] = fmaf(w, h[r * HID + k], a[r]);
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < ROWS; ++r) h[r * HID + lane] = leaky(a[r]);
      __syncthreads();
      for (int i = lane; i < ROWS * D; i += 64) {
        const int r = i / D, d = i % D;
        float acc = f.b2[net][li * D + d];
        for (int k = 0; k < HID; ++k) acc = fmaf(w2[d * HID + k], h[r * HID + k], acc);
        if (net == 0) acc = tanhf(acc);
        st[net * ROWS * D + i] = acc * (1.f - mask[d]);
      }
      __syncthreads();
    }
    for (int i = lane; i < ROWS * D; i += 64) {
      const float m = mask[i % D];
      const float s = st[i], t = st[ROWS * D + i];
      z[i] = forward ? zm[i] + (1.f - m) * (z[i] * expf(s) + t)
                     : (1.f - m) * (z[i] - t) * expf(-s) + zm[i];
    }
    if (lane < ROWS) {
      float sum = 0.f;
      for (int d = 0; d < D; ++d) sum += st[lane * D + d];
      ld[lane] -= sum;
    }
    __syncthreads();
  }
  if (forward) {
    for (int i = lane; i < nr * D; i += 64) out[(size_t)r0 * D + i] = z[i];
  } else if (lane < nr) {
    float q = 0.f;
    for (int d = 0; d < D; ++d) q += z[lane * D + d] * z[lane * D + d];
    // MultivariateNormal(0, I_9).log_prob(z) + log_det   (real_nvp.py:64-65)
    out[r0 + lane] = -0.5f * q - 0.5f * D * 1.8378770664093453f + ld[lane];
  }
}

}  // namespace

void launch_realnvp(const FlowDev& f, const float* x, const float* ctx, float* out, int N, int forward,
                    hipStream_t s) {
  const size_t lds = sizeof(float) * (ROWS * f.ctx + 2 * ROWS * D + ROWS * HID + 2 * ROWS * D + ROWS);
  hipLaunchKernelGGL(realnvp_kernel, dim3((N + ROWS - 1) / ROWS), dim3(64), lds, s, f, x, ctx, out, N, forward);
}
