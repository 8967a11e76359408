// SMPL linear blend skinning + joints + camera projection for gfx950.
//
// Restates smplx.lbs.lbs as invoked by the reference (pocolib/models/head/smpl_head.py:22-34,53-58,
// smplcam_head.py:48-53; algorithm: SURVEY.md 3.5) as three batched small-matrix kernels:
//   1. smpl_chain_kernel   : J(betas), 24-joint kinematic chain of 3x4 affines, A_i = G_i * [I|-J_i]
//   2. smpl_skin_kernel    : the blend shapes as ONE GEMM on the fp32 MFMA (round 5),
//                                v_posed[crop][v, c] = sum_k coef[crop][k] * blend[k][c][v],   k = 207 pose | 10 shape | template,
//                            coef = [R - I (23 joints x 9) | betas | 1] written by the chain kernel, blend = posedirs / shapedirs /
//                            v_template stacked coordinate-major on the host; then skinning (24 x 12 per vertex and crop) + apply on
//                            the VALU.  A block owns 32 vertices x up to 64 crops: the 18 MB of blend rows are streamed once per 64
//                            crops in 64-byte runs.  (Rounds 1-4: a VALU kernel, one thread per vertex x 8 crops, 207 x 3
//                            12-byte loads per thread: 42 us for 64 crops; this one: see DESIGN.md.)
//   3. smpl_joints_kernel  : 9 regressed extra joints (block reduction over 6890 vertices), 21 vertex
//                            picks, 49-joint gather (smpl_head.py:25-27).
// HBM bound: 82.7 KB of vertices written per crop; algorithmic reads ~17.8 MB of model per CB crops.
#include "kernels.h"

namespace {

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
  // coefficient row of the blend GEMM: pose feature (rotmat - I of joints 1..23, smplx lbs.py `pose_feature`), betas, 1 (template)
  float* cf = io.coef + (size_t)b * SMPL_KB;
  for (int k = t; k < SMPL_KB; k += 64) {
    float v = 0.f;
    if (k < 207) { const int e = 9 + k; v = R[e / 9][e % 9] - (((e % 9) % 4 == 0) ? 1.f : 0.f); }
    else if (k < 217) v = betas[k - 207];
    else if (k == 217) v = 1.f;
    cf[k] = v;
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

// grid (VP / 32, ceil(B / 64)), 8 waves: wave w owns crops 64 by + 16 (w & 3) ... + 15 (MFMA rows) x 16 of the block's 32 vertices
// (tile w >> 2) x 3 coordinates: three accumulators.  D[i = crop][j = vertex]: lane (j = lane % 16, q = lane / 16) ends up with
// x, y, z of vertex j for crops 4 q ... 4 q + 3 of the wave's 16 - one vertex's 24 skinning weights per lane, the crops' 3 x 4
// affines A (chain kernel) broadcast from LDS.  The kernel is a chain of memory round trips, not of MFMAs (55 K steps x 3 per wave):
// the operands of 28 K steps are requested together, i.e. two round trips per wave.
constexpr int SKIN_CH = 28;
__global__ void __launch_bounds__(512)
smpl_skin_kernel(SmplDev m, SmplIO io, int B) {
  // 64 crops x 24 affines of 12 floats = 73.7 KB of the CU's 160 KB (this file is built for gfx950 only: poco_amd/build.py ARCH)
  static_assert(sizeof(float) * 64 * 288 <= 160 * 1024, "smpl_skin_kernel: the 64-crop affine tile must fit the LDS of a gfx950 CU");
  __shared__ __attribute__((aligned(16))) float As[64][288];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, q = lane >> 4;
  const int b0 = blockIdx.y * 64;
  const int nb = min(64, B - b0);
  const int c0 = 16 * (wave & 3);                     // first crop of this wave inside the block
  const int v = blockIdx.x * 32 + 16 * (wave >> 2) + j;
  const bool active = c0 < nb;                        // wave-uniform
  const float* cf = io.coef + (size_t)(b0 + min(c0 + j, nb - 1)) * SMPL_KB + q;      // A operand: row = crop, k = 4 step + q
  const float* bl = m.blend_cm + (size_t)q * 3 * m.VP + v;                             // B operand: k = 4 step + q, column = vertex
  const size_t kstride = (size_t)4 * 3 * m.VP;                                        // floats per K step
  f32x4 acc[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int NS = SMPL_KB / 4;
  float a[SKIN_CH], bv[SKIN_CH][3];
  auto request = [&](int s0) {
#pragma unroll
    for (int u = 0; u < SKIN_CH; ++u) {
      const int st = min(s0 + u, NS - 1);             // past the end: a re-read, not used
      a[u] = cf[st * 4];
      const float* bp = bl + (size_t)st * kstride;
#pragma unroll
      for (int c = 0; c < 3; ++c) bv[u][c] = bp[(size_t)c * m.VP];
    }
  };
  if (active) request(0);                             // first round trip under the staging of the affines
  for (int i = tid; i < nb * 72; i += 512)            // float4 copies of the block's affines
    reinterpret_cast<float4*>(&As[0][0])[i] = reinterpret_cast<const float4*>(io.A + (size_t)b0 * 288)[i];
  __syncthreads();
  if (!active) return;
#pragma unroll 1
  for (int s0 = 0; s0 < NS; s0 += SKIN_CH) {
#pragma unroll
    for (int u = 0; u < SKIN_CH; ++u)
      if (s0 + u < NS) {
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], bv[u][c], acc[c], 0, 0, 0);
      }
    if (s0 + SKIN_CH < NS) request(s0 + SKIN_CH);
  }
  if (v >= m.V) return;
  float w[24];
  const float4* wr = reinterpret_cast<const float4*>(m.lbs_weights + (size_t)v * 24);
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const float4 t = wr[k];
    w[k * 4] = t.x; w[k * 4 + 1] = t.y; w[k * 4 + 2] = t.z; w[k * 4 + 3] = t.w;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int cb = c0 + 4 * q + e;
    if (cb >= nb) continue;
    float T[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) T[k] = 0.f;
#pragma unroll
    for (int jj = 0; jj < 24; ++jj) {
      const float wj = w[jj];
      const float4* a4 = reinterpret_cast<const float4*>(&As[cb][jj * 12]);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float4 aa = a4[k];
        T[k * 4] = fmaf(wj, aa.x, T[k * 4]); T[k * 4 + 1] = fmaf(wj, aa.y, T[k * 4 + 1]);
        T[k * 4 + 2] = fmaf(wj, aa.z, T[k * 4 + 2]); T[k * 4 + 3] = fmaf(wj, aa.w, T[k * 4 + 3]);
      }
    }
    const float x = acc[0][e], y = acc[1][e], z = acc[2][e];
    float* o = io.verts + ((size_t)(b0 + cb) * m.V + v) * 3;
    o[0] = T[0] * x + T[1] * y + T[2] * z + T[3];
    o[1] = T[4] * x + T[5] * y + T[6] * z + T[7];
    o[2] = T[8] * x + T[9] * y + T[10] * z + T[11];
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
  // vertices tid, tid + 1024, ...: all 12 loads of an iteration are independent of the previous one's - unrolled so that they are
  // requested together (a rolled loop is 7 dependent memory round trips per crop); past-the-end lanes re-read vertex V - 1 with weight 0
#pragma unroll
  for (int it = 0; it < SMPL_JOINTS_ITERS; ++it) {     // (V <= SMPL_MAX_V is checked where the body model is loaded)
    const int vv = tid + it * 1024;
    const int v = min(vv, m.V - 1);
    const float keep = vv < m.V ? 1.f : 0.f;
    const float x = vb[v * 3], y = vb[v * 3 + 1], z = vb[v * 3 + 2];
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      const float wj = m.J_regressor_extra[(size_t)j * m.V + v] * keep;
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
    if (io.joints49_out) io.joints49_out[(size_t)b * 147 + tid] = val;
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
  hipLaunchKernelGGL(smpl_skin_kernel, dim3(m.VP / 32, (B + 63) / 64), dim3(512), 0, s, m, io, B);
  hipLaunchKernelGGL(smpl_joints_kernel, dim3(B), dim3(1024), 0, s, m, io, B);
}

void launch_camera(const CamArgs& a, int B, hipStream_t s) {
  hipLaunchKernelGGL(camera_kernel, dim3((B * 49 + 127) / 128), dim3(128), 0, s, a, B);
}
