// SMPL linear blend skinning + joints + camera projection for gfx950.
//
// Restates smplx.lbs.lbs as invoked by the reference (pocolib/models/head/smpl_head.py:22-34,53-58,
// smplcam_head.py:48-53; algorithm: SURVEY.md 3.5) as three batched small-matrix kernels:
//   1. smpl_chain_kernel   : J(betas), 24-joint kinematic chain of 3x4 affines, A_i = G_i * [I|-J_i]
//   2. smpl_skin_kernel    : per vertex: shape blend (10), pose blend (207), skinning (24x12), apply.
//                            A block owns SKIN_T vertices x CB crops so posedirs (17 MB, the only large
//                            operand) is streamed once per CB crops with 12-byte coalesced reads.
//   3. smpl_joints_kernel  : 9 regressed extra joints (block reduction over 6890 vertices), 21 vertex
//                            picks, 49-joint gather (smpl_head.py:25-27).
// HBM bound: 82.7 KB of vertices written per crop; algorithmic reads ~17.8 MB of model per CB crops.
#include "kernels.h"

namespace {

constexpr int CB = 8;    // crops per skinning block (posedirs is re-read once per CB crops).  Round 3 re-measured the neighbours at 64
                         // crops (all three kernels): CB = 4 / 256 threads 80 us, CB = 8 / 256 threads 67 us, CB = 16 / 128 threads 117 us
constexpr int SKIN_T = 256;  // vertices (threads) per skinning block

// One wave per crop.  The chain is serial over the 24 joints (each needs its parent), but the 12 entries of a
// joint's 3x4 affine are independent: lanes 0..11 compute one entry each from LDS, one barrier per joint.
__global__ void __launch_bounds__(64)
smpl_chain_kernel(SmplDev m, SmplIO io, int B) {
  __shared__ float J[24][3];
  __shared__ float R[24][9];
  __shared__ float G[24][12];   // row-major 3x4 world transforms
  const int b = blockIdx.x, t = threadIdx.x;
  const float* betas = io.betas + (size_t)b * io.betas_stride;
  for (int i = t; i < 72; i += 64) {
    float acc = m.J_template[i];
    for (int l = 0; l < 10; ++l) acc = fmaf(m.J_shapedirs[i * 10 + l], betas[l], acc);
    J[i / 3][i % 3] = acc;
  }
  for (int i = t; i < 216; i += 64) R[i / 9][i % 9] = io.rotmat[(size_t)b * io.rot_stride + i];
  __syncthreads();
  const int r = t >> 2, c = t & 3;          // entry (r, c) of the 3x4 affine, lanes 0..11
  for (int i = 0; i < 24; ++i) {
    const int p = m.parents[i];
    if (t < 12) {
      // local transform T_i = [R_i | J_i - J_parent]
      auto T = [&](int rr, int cc) { return cc < 3 ? R[i][rr * 3 + cc] : (i == 0 ? J[0][rr] : J[i][rr] - J[p][rr]); };
      float v;
      if (i == 0) v = T(r, c);
      else {
        v = G[p][r * 4 + 0] * T(0, c) + G[p][r * 4 + 1] * T(1, c) + G[p][r * 4 + 2] * T(2, c);
        if (c == 3) v += G[p][r * 4 + 3];
      }
      G[i][t] = v;
    }
    __syncthreads();
  }
  float* A = io.A + (size_t)b * 288;
  float* j24 = io.joints24 + (size_t)b * 72;
  for (int k = t; k < 288; k += 64) {
    const int i = k / 12, e = k % 12, rr = e >> 2, cc = e & 3;
    float v = G[i][e];
    if (cc == 3) {
      v -= G[i][rr * 4 + 0] * J[i][0] + G[i][rr * 4 + 1] * J[i][1] + G[i][rr * 4 + 2] * J[i][2];
      j24[i * 3 + rr] = G[i][e];
    }
    A[k] = v;
  }
}

__global__ void __launch_bounds__(SKIN_T)
smpl_skin_kernel(SmplDev m, SmplIO io, int B) {
  __shared__ float pf[CB][208];
  __shared__ float bt[CB][12];
  __shared__ __attribute__((aligned(16))) float As[CB][288];
  const int tid = threadIdx.x;
  const int v = blockIdx.x * SKIN_T + tid;
  const int b0 = blockIdx.y * CB;
  const int nb = min(CB, B - b0);
  for (int i = tid; i < CB * 207; i += SKIN_T) {
    const int cb = i / 207, k = i % 207;
    float val = 0.f;
    if (cb < nb) {
      const int e = 9 + k;   // skip the root joint's 9 entries
      val = io.rotmat[(size_t)(b0 + cb) * io.rot_stride + e] - (((e % 9) % 4 == 0) ? 1.f : 0.f);
    }
    pf[cb][k] = val;
  }
  for (int i = tid; i < CB * 10; i += SKIN_T) {
    const int cb = i / 10, l = i % 10;
    bt[cb][l] = (cb < nb) ? io.betas[(size_t)(b0 + cb) * io.betas_stride + l] : 0.f;
  }
  for (int i = tid; i < CB * 288; i += SKIN_T) {
    const int cb = i / 288, k = i % 288;
    As[cb][k] = (cb < nb) ? io.A[(size_t)(b0 + cb) * 288 + k] : 0.f;
  }
  __syncthreads();
  if (v >= m.V) return;
  const int V3 = m.V * 3;
  float vp[CB][3];
  {
    const float t0 = m.v_template[v * 3], t1 = m.v_template[v * 3 + 1], t2 = m.v_template[v * 3 + 2];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) { vp[cb][0] = t0; vp[cb][1] = t1; vp[cb][2] = t2; }
  }
  for (int l = 0; l < 10; ++l) {
    const float* sd = m.shapedirs + (size_t)l * V3 + v * 3;
    const float s0 = sd[0], s1 = sd[1], s2 = sd[2];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      const float be = bt[cb][l];
      vp[cb][0] = fmaf(s0, be, vp[cb][0]); vp[cb][1] = fmaf(s1, be, vp[cb][1]); vp[cb][2] = fmaf(s2, be, vp[cb][2]);
    }
  }
  // pose blend shapes: accumulate the offset separately, then add (matches v_shaped + offsets)
  float po[CB][3];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) po[cb][0] = po[cb][1] = po[cb][2] = 0.f;
  // 207 = 9 x 23: the 23 row loads of a chunk are issued together (the loop is latency-bound: < 1 wave per SIMD)
#pragma unroll 1
  for (int k0 = 0; k0 < 207; k0 += 23) {
    float d[23][3];
#pragma unroll
    for (int u = 0; u < 23; ++u) {
      const float* pd = m.posedirs + (size_t)(k0 + u) * V3 + v * 3;
      d[u][0] = pd[0]; d[u][1] = pd[1]; d[u][2] = pd[2];
    }
#pragma unroll
    for (int u = 0; u < 23; ++u) {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        const float f = pf[cb][k0 + u];
        po[cb][0] = fmaf(f, d[u][0], po[cb][0]); po[cb][1] = fmaf(f, d[u][1], po[cb][1]); po[cb][2] = fmaf(f, d[u][2], po[cb][2]);
      }
    }
  }
  float w[24];
  {
    const float4* wr = reinterpret_cast<const float4*>(m.lbs_weights + (size_t)v * 24);
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const float4 t = wr[q];
      w[q * 4] = t.x; w[q * 4 + 1] = t.y; w[q * 4 + 2] = t.z; w[q * 4 + 3] = t.w;
    }
  }
#pragma unroll     // fully unrolled (vp/po stay in registers); the tail of a ragged last block is predicated
  for (int cb = 0; cb < CB; ++cb) {
    if (cb < nb) {
    float T[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) T[k] = 0.f;
#pragma unroll
    for (int j = 0; j < 24; ++j) {
      const float wj = w[j];
      const float4* a4 = reinterpret_cast<const float4*>(&As[cb][j * 12]);
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const float4 a = a4[q];
        T[q * 4] = fmaf(wj, a.x, T[q * 4]); T[q * 4 + 1] = fmaf(wj, a.y, T[q * 4 + 1]);
        T[q * 4 + 2] = fmaf(wj, a.z, T[q * 4 + 2]); T[q * 4 + 3] = fmaf(wj, a.w, T[q * 4 + 3]);
      }
    }
    const float x = vp[cb][0] + po[cb][0], y = vp[cb][1] + po[cb][1], z = vp[cb][2] + po[cb][2];
    float* o = io.verts + ((size_t)(b0 + cb) * m.V + v) * 3;
    o[0] = T[0] * x + T[1] * y + T[2] * z + T[3];
    o[1] = T[4] * x + T[5] * y + T[6] * z + T[7];
    o[2] = T[8] * x + T[9] * y + T[10] * z + T[11];
    }
  }
}

__global__ void __launch_bounds__(1024)
smpl_joints_kernel(SmplDev m, SmplIO io, int B) {
  __shared__ float red[16][27];
  __shared__ float extra[27];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* vb = io.verts + (size_t)b * m.V * 3;
  float acc[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) acc[k] = 0.f;
  for (int v = tid; v < m.V; v += 1024) {
    const float x = vb[v * 3], y = vb[v * 3 + 1], z = vb[v * 3 + 2];
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      const float wj = m.J_regressor_extra[(size_t)j * m.V + v];
      acc[j * 3] = fmaf(wj, x, acc[j * 3]); acc[j * 3 + 1] = fmaf(wj, y, acc[j * 3 + 1]);
      acc[j * 3 + 2] = fmaf(wj, z, acc[j * 3 + 2]);
    }
  }
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    float v = acc[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((tid & 63) == 0) red[tid >> 6][k] = v;
  }
  __syncthreads();
  if (tid < 27) {
    float s = 0.f;
    for (int w = 0; w < 16; ++w) s += red[w][tid];
    extra[tid] = s;
  }
  __syncthreads();
  if (tid < 49 * 3) {
    const int t = tid / 3, k = tid % 3;
    const int idx = m.joint_map[t];
    float val;
    if (idx < 24) val = io.joints24[(size_t)b * 72 + idx * 3 + k];
    else if (idx < 45) val = vb[(size_t)m.extra_vertex_ids[idx - 24] * 3 + k];
    else val = extra[(idx - 45) * 3 + k];
    io.joints49[(size_t)b * 147 + tid] = val;
  }
}

__global__ void camera_kernel(CamArgs a, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * 49) return;
  const int t = i % 49, b = i / 49;
  const float* cam = a.cam + (size_t)b * a.cam_stride;
  const float s = cam[0], tx = cam[1], ty = cam[2];
  // convert_weak_perspective_to_perspective, utils/geometry.py:447-463
  const float ctz = 2.f * 5000.f / (224.f * s + 1e-9f);
  float px, py, pz, f, cx, cy, post;
  if (a.cliff) {
    // convert_pare_to_full_img_cam, smplcam_head.py:123-139
    const float focal = a.focal[b];
    const float bh = a.scale[b] * 200.f;
    const float img_h = a.orig_shape[b * 2], img_w = a.orig_shape[b * 2 + 1];
    const float r = bh / 224.f;
    const float tz = 2.f * focal / (r * 224.f * s);
    const float ox = 2.f * (a.center[b * 2] - img_w / 2.f) / (s * bh);
    const float oy = 2.f * (a.center[b * 2 + 1] - img_h / 2.f) / (s * bh);
    px = tx + ox; py = ty + oy; pz = tz;
    f = focal; cx = img_w / 2.f; cy = img_h / 2.f; post = 1.f;
    if (t == 0 && a.fullimg_cam_t) {
      a.fullimg_cam_t[b * 3] = px; a.fullimg_cam_t[b * 3 + 1] = py; a.fullimg_cam_t[b * 3 + 2] = pz;
    }
  } else {
    px = tx; py = ty; pz = ctz; f = 5000.f; cx = 0.f; cy = 0.f; post = 1.f / 112.f;
  }
  if (t == 0) { a.cam_t[b * 3] = tx; a.cam_t[b * 3 + 1] = ty; a.cam_t[b * 3 + 2] = ctz; }
  const float* J = a.joints49 + (size_t)i * 3;
  const float X = J[0] + px, Y = J[1] + py, Z = J[2] + pz;
  const float u = f * (X / Z) + cx, v = f * (Y / Z) + cy;
  a.joints2d[(size_t)i * 2] = a.cliff ? u : u / 112.f;
  a.joints2d[(size_t)i * 2 + 1] = a.cliff ? v : v / 112.f;
  (void)post;
}

}  // namespace

void launch_smpl_lbs(const SmplDev& m, const SmplIO& io, int B, hipStream_t s) {
  hipLaunchKernelGGL(smpl_chain_kernel, dim3(B), dim3(64), 0, s, m, io, B);
  hipLaunchKernelGGL(smpl_skin_kernel, dim3((m.V + SKIN_T - 1) / SKIN_T, (B + CB - 1) / CB), dim3(SKIN_T), 0, s, m, io, B);
  hipLaunchKernelGGL(smpl_joints_kernel, dim3(B), dim3(1024), 0, s, m, io, B);
}

void launch_camera(const CamArgs& a, int B, hipStream_t s) {
  hipLaunchKernelGGL(camera_kernel, dim3((B * 49 + 127) / 128), dim3(128), 0, s, a, B);
}
