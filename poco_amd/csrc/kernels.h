// Launchers of the non-GEMM kernels (all enqueue on `stream`, no allocation, no sync).
#pragma once
// float4 index of (row = b*H + y, x, channel quad c4) in the L16 activation layout (csrc/common.h): a buffer of
// C channels has C/16 slice rows of W*16 floats per image row.  Usable from host and device code.
#define L16_F4(row, x, c4, W, C16) ((((size_t)(row) * (C16)) + ((c4) >> 2)) * (size_t)(W) * 4 + (size_t)(x) * 4 + ((c4) & 3))
#include "common.h"

// ---- backbone side kernels (kernels_misc.hip) ---------------------------------------------------
// Direct small-Cin stem conv (Cin = 3): NCHW image -> NHWC features, + shift, ReLU.
//   ks=3,stride 2,pad 1 : hrnet.py:467-469 ;  ks=7,stride 2,pad 3 : resnet.py:203-205
// w: [ks*ks*3][Cout] (tap-major, scale folded), shift [Cout]; Cout must be 64.
void launch_stem_conv(const float* img_nchw, const float* w, const float* shift, float* out_nhwc, int B,
                      int H, int W, int ks, hipStream_t s, int use_mfma = 1);
// the same as an implicit GEMM on the fp32 MFMA (stem_mfma.hip); false = shape not covered (Wo != 112), nothing launched
bool launch_stem_conv_mfma(const float* img_nchw, const float* w, const float* shift, float* out_nhwc, int B, int H, int W, int ks,
                           hipStream_t s);
// 3x3 stride-2 pad-1 max pool, NHWC (resnet.py:206).
// conv3 + residual + ReLU of one Bottleneck chained with conv1 + ReLU of the next (bneck_chain.hip)
int launch_bneck_chain(const float* t, int t_cs, const float* res, int res_cs, float* y, int y_cs, float* u, int u_cs,
                       const float* w3, const float* b3, const float* w1, const float* b1, int B, int H, int W,
                       hipStream_t s);
// K-concatenated projection shortcut with a strided second source (gemm1x1.hip)
int launch_gemm1x1_dual(const float* a, int a_cs, int Ca, const float* b, int b_cs, int Cb, int H2, int W2, int stride2,
                        const float* wfrag, const float* bias, float* out, int out_cs, int Cout, int B, int Ho, int Wo, int act,
                        hipStream_t stream, int wave_layout = 0);   // wave_layout = 100 NI + 10 WM + WN (0 = one wave per block, hipcc's load order)
void launch_maxpool3x3s2(const float* in, float* out, int B, int H, int W, int C, int out_cs, hipStream_t s);
// x2 bilinear upsample, align_corners=True, NHWC (hrnet.py:440).
void launch_bilinear_up2x(const float* in, float* out, int B, int H, int W, int C, hipStream_t s);
// out = ReLU( sum_k addend_k[b, y >> shift_k, x >> shift_k, :] )  (HRNet fuse, hrnet.py:257-264).
struct FuseArgs {
  const float* src[4];   // first channel of each term
  int shift[4];          // nearest-neighbour upsampling: the term's plane is (H >> shift) x (W >> shift)
  int src_cs[4];         // channels per pixel of each term's buffer (a term may be a channel slice)
  int n;
};
// `out` points at the first output channel; out_cs = channels per pixel of the destination buffer.
void launch_fuse_sum(const FuseArgs& a, float* out, int B, int H, int W, int C, int out_cs, int relu, hipStream_t s);
// Global average pool NHWC [B,HW,C] -> dst[b*dst_stride + c] (hrnet_cls.py:482, cliff_head.py:96).
void launch_avgpool(const float* in, float* dst, int B, int H, int W, int C, int dst_stride, hipStream_t s);

// ---- head kernels (kernels_head.hip) ---------------------------------------------------------------
// Part attention pooling (KeypointAttention, layers/keypoint_attention.py:34-48):
//   out[b, c, j] = sum_p softmax_p(heat[b,p,1+j]) * feat[b,p,c]      (24 parts; heat channel 0 = bg)
// heat: NHWC [B,HW,heat_cs] ; feat: NHWC [B,HW,C] ; out: dst[b*dst_stride + c*24 + j] (channel-major).
// scratch: part_attention_scratch_floats(B, C) floats.  C <= 128.
size_t part_attention_scratch_floats(int B, int C);
void launch_part_attention_pool_ws(const float* heat, int heat_cs, const float* feat, int C, float* dst,
                                   int dst_stride, int B, int H, int W, float* scratch, hipStream_t s);
// LocallyConnected2d 128->6 per joint (layers/locallyconnected2d.py:27-37):
//   pose6d[b, j, o] = sum_c x[b*x_stride + c*24 + j] * w[o][c][j]
void launch_lc2d_pose(const float* x, int x_stride, const float* w /*[6][128][24]*/, float* pose6d /*[B,144]*/,
                      int B, hipStream_t s);
// rot6d -> rotmat (utils/geometry.py:247-261). in: [B, in_stride] holding 144 floats (24 x (3x2));
// writes rotmat (24*9 per crop) to up to two destinations (stride in floats, nullable).
void launch_rot6d(const float* in, int in_stride, float* dst0, int stride0, float* dst1, int stride1, int B,
                  hipStream_t s);
// Strided row copy: dst[b*dst_stride + i] = src[b*src_stride + i], i < n.
void launch_copy_rows(const float* src, int src_stride, float* dst, int dst_stride, int n, int B,
                      hipStream_t s);
// dst[b*dst_stride + i] = src[i] (broadcast an init vector into every crop's row).
void launch_broadcast_rows(const float* src, float* dst, int dst_stride, int n, int B, hipStream_t s);
// NHWC [B,HW,cs] (first C channels) -> NCHW [B,C,HW]
void launch_nhwc_to_nchw(const float* in, int cs, float* out, int B, int H, int W, int C, hipStream_t s);

// ---- the CLIFF regressor as one persistent launch (mlp_chain.hip) -----------------------------------------
// A program of stages; the jobs of a stage are independent, consecutive stages are separated by a grid barrier.
struct MlpLayer {            // out[r][n] = act(bias[n] + sum_k W[n][k] in[r][k] (+ res[r][n])) for rows r < B (row-major vectors)
  const float* in;           // [B][in_rs], K = 16 nC16 consecutive floats from `in`
  const float* res;          // nullable, [B][res_rs]
  float* out;                // [B][out_rs]
  const float4* wfrag;       // conv_pack_weights(ks = 1): [nC16][nT16][64] float4
  const float* bias;         // [16 nT16]
  int nC16, nT16, in_rs, res_rs, out_rs, act;   // act: 0 none, 1 ReLU, 2 sigmoid
};
enum { MLP_ROW_COPY = 0, MLP_ROW_BCAST = 1, MLP_ROW_ROT6D = 2 };
struct MlpRowJob {           // COPY: dst[r][i] = src[r][i], i < n; BCAST: dst[r][i] = src[i]; ROT6D: 24 x 6 -> 24 x 9 into dst and dst2 (nullable)
  const float* src;
  float* dst;
  float* dst2;
  int kind, src_rs, dst_rs, dst2_rs, n;
};
struct MlpStage { int layer0, nlayers, row0, nrows; };
constexpr int MLP_MAX_LAYERS = 16, MLP_MAX_ROWS = 12, MLP_MAX_STAGES = 14;
struct MlpProgram {
  MlpLayer layer[MLP_MAX_LAYERS];
  MlpRowJob row[MLP_MAX_ROWS];
  MlpStage stage[MLP_MAX_STAGES];
  int nstages, B;
  unsigned* sync;            // device, 1 KiB: arrival counters (mlp_chain.hip); zero before the first launch, re-armed by the kernel
  unsigned* err_host;        // pinned host word (device-visible): set when a grid barrier timed out
  unsigned max_spins;        // bound of a barrier's poll loop (0 = the default, 2^21 polls ~ 3 s); engine option debug_wait_spins
  int debug_skip_arrival;    // test hook (engine option debug_mlp_timeouts): block 0 does not arrive at the first barrier, i.e. every block times out
  long long* trace;          // nullable (tools/probe/mlp_probe.hip): block 0 writes wall_clock64() at the start, after each stage's jobs and after each barrier
};
int mlp_chain_grid(const MlpProgram& p, int max_blocks);
int launch_mlp_chain(const MlpProgram& p, int max_blocks, hipStream_t s);
// blocks of mlp_chain_kernel the device can hold at once (CUs x blocks per CU, queried once): the hand-rolled grid barrier needs
// every block of the grid resident
int mlp_chain_resident_blocks();

// ---- SMPL (kernels_smpl.hip) -------------------------------------------------------------------------
constexpr int SMPL_JOINTS_ITERS = 7;                     // smpl_joints_kernel: 1024 threads x this many hard-unrolled vertex rounds ...
constexpr int SMPL_MAX_V = SMPL_JOINTS_ITERS * 1024;     // ... i.e. body models of up to 7168 vertices (SMPL: 6890); larger ones are refused at load time
constexpr int SMPL_KB = 220;   // K of the blend GEMM: 207 pose + 10 shape + 1 template, padded to a multiple of 4
struct SmplDev {
  int V;                       // 6890
  const float* v_template;     // [V,3]
  const float* shapedirs;      // [10][V*3]   (transposed for coalescing)
  const float* posedirs;       // [207][V*3]
  const float* blend_cm;       // [SMPL_KB][3][VP]: rows 0..206 posedirs, 207..216 shapedirs, 217 v_template, coordinate-major (kernels_smpl.hip)
  int VP;                      // V padded to a multiple of 32
  const float* lbs_weights;    // [V,24]
  const float* J_template;     // [24,3]      J_regressor . v_template
  const float* J_shapedirs;    // [24,3,10]   J_regressor . shapedirs
  const float* J_regressor_extra;  // [9,V]
  const int* parents;          // [24]
  const int* extra_vertex_ids; // [21]
  const int* joint_map;        // [49]
};
struct SmplIO {
  const float* betas;  int betas_stride;    // [B,10]
  const float* rotmat; int rot_stride;      // [B,216]
  float* A;                                 // scratch [B,24,12]
  float* coef;                              // scratch [B,SMPL_KB]: coefficient rows of the blend GEMM (chain kernel -> skin kernel)
  float* joints24;                          // scratch [B,24,3]
  float* verts;                             // [B,V,3] output
  float* joints49;                          // [B,49,3] output
  float* joints49_out;                      // nullable: the same, written a second time (the caller's smpl_joints3d)
};
// smplx.lbs.lbs restated + the 49-joint wrapper of smpl_head.py:22-34.
void launch_smpl_lbs(const SmplDev& m, const SmplIO& io, int B, hipStream_t s);

struct CamArgs {
  const float* cam; int cam_stride;   // [B,3] (s,tx,ty)
  const float* joints49;              // [B,49,3]
  // cliff only (nullable for pare):
  const float* focal; const float* scale; const float* center; const float* orig_shape;
  float* cam_t;          // [B,3]
  float* fullimg_cam_t;  // [B,3] (cliff) nullable
  float* joints2d;       // [B,49,2]
  int cliff;
};
// smpl_head.py:63-78 (pare) / smplcam_head.py:65-90 (cliff) camera conversion + projection.
void launch_camera(const CamArgs& a, int B, hipStream_t s);

// ---- RealNVP (kernels_flow.hip) -----------------------------------------------------------------------
struct FlowDev {
  int L;               // number of coupling layers (rows of mask; even: nf_head.py:20-21 builds mask pairs)
  int ctx;             // context dim (512)
  const float* mask16; // [L][16]: mask rows zero-padded from 9 to 16
  // per (layer, net s|t): 24 x 64 float4 MFMA A-operand fragments = conv_pack_weights(ks = 1) of
  //   W0[:, :9] (K padded to 16) [4 quads] | W1 [16 quads: k slice major] | W2 (rows padded to 16) [4 quads]
  const float4* wpack;
  const float* b1;     // [L][2][64]
  const float* b2;     // [L][2][16] (zero padded)
  // step A (context GEMM): W0[:, 9:] of all 2L MLPs stacked to [L*2*64][512] in conv_pack_weights(ks = 1) order, bias = b0
  const float* wctx_frag;
  const float* bctx;   // [L*2*64]
};
// log_prob (backward_p + N(0,I) prior, layers/real_nvp.py:40-65; forward = 0) or forward_p (:25-38; forward = 1) of N rows.
// ctx: [ceil(N/rep)][512], row r uses context row r / rep (rep = 1: one context per row; rep = 24: nf_head.py:105-110's
// repeat_interleave without materialising it).  scratch: realnvp_scratch_floats(f, ceil(N/rep)) floats.
size_t realnvp_scratch_floats(const FlowDev& f, int ctx_rows);
int launch_realnvp(const FlowDev& f, const float* x, const float* ctx, int rep, float* out, int N, int forward, float* scratch,
                   hipStream_t s);

// ---- preprocessing (kernels_misc.hip) --------------------------------------------------------------------
// frame: uint8 [H,W,3] RGB (device); boxes: [N,4] = (cx, cy, w, h) in pixels (float32 or float64); out: [N,3,res,res] fp32
// NCHW = ToTensor + Normalize of the uint8 crop cv2.warpAffine(getAffineTransform(box * scale -> res x res)) makes, byte-exact.
void launch_crop_normalize(const unsigned char* frame, int H, int W, const float* boxes, double bbox_scale, float* out,
                           int N, int res, hipStream_t s);
void launch_crop_normalize_f64(const unsigned char* frame, int H, int W, const double* boxes, double bbox_scale, float* out,
                               int N, int res, hipStream_t s);
// crops of several same-sized frames in one launch: frames = device array of frame pointers, frame_idx[n] = frame of crop n
void launch_crop_normalize_multi(const unsigned char* const* frames, int nframes, const int* frame_idx, int H, int W, const float* boxes,
                                 double bbox_scale, float* out, int N, int res, hipStream_t s);
// poco_outputs_t.record: [rotmat 216 | betas 10 | cam 3 | var 24 | post-processed confidence 1] per crop (254 floats)
void launch_pack_record(const float* rot, int rot_stride, const float* betas, int betas_stride, const float* cam, int cam_stride,
                        const float* var, int var_stride, float* rec, int cliff, int kinematic, float thr, int B, hipStream_t s);
