/* poco_hip.h — C ABI of libpoco_hip.so, the MI355X (gfx950) implementation of POCO's per-crop
 * SMPL regressor hot path.
 *
 * The reference (saidwivedi/POCO) has no FFI/plugin interface: its seam is the Python call
 * `output = self.model(batch)` on an nn.Module (pocolib/core/tester.py:213,408;
 * pocolib/models/poco.py:99-129).  This header is therefore the boundary a binding for that call
 * site uses; every entry point names the reference code it replaces.  Plain C types only: device
 * and host pointers, sizes, a hipStream_t passed as void*.  No torch types.
 *
 * Conventions
 *   - all functions return 0 (POCO_OK) or a positive error code; poco_last_error() returns the
 *     message of the last failure on the calling thread.  Nothing throws across the ABI.
 *   - "d_" pointers are device (HBM) pointers owned by the caller, "h_" pointers are host memory.
 *   - activations handed to the stand-alone conv operators are fp32 in the library's internal layout "L16"
 *     (channel-slice-major NHWC): [B][H][C/16][W][16], i.e. element (b,y,x,c) at
 *     ((b*H + y)*(C/16) + c/16)*W*16 + x*16 + c%16; for H = W = 1 this is plain [B][C].
 *     poco_forward itself takes the reference's NCHW image batch and returns the reference's tensors.
 *   - `stream` is a hipStream_t (NULL = default stream).  Operators enqueue on it; the
 *     poco_op_* test entry points additionally synchronise it before returning.
 */
#ifndef POCO_HIP_H
#define POCO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Message of the last error on this thread ("" if none). */
const char* poco_last_error(void);

/* Version of this ABI: bumped whenever a struct layout or the meaning / type of an argument of an existing entry point changes
 * (2: poco_crop_normalize takes bbox_scale as double; 3: poco_outputs_t gained `record`, poco_create_ex, poco_crop_normalize_multi,
 * RealNVP scratch planned at finalize; 4: poco_inputs_t / poco_outputs_t start with `struct_size`, see below).  A binding
 * compiled against another header must refuse to run:
 *     if (poco_abi_version() != POCO_ABI_VERSION) fail;                                                                          */
#define POCO_ABI_VERSION 4
int poco_abi_version(void);

/* ---- the engine: POCO(backbone=..., pretrained=ckpt) + model(batch) ---------------------------
 * replaces pocolib/models/poco.py:13-154 (ctor :13-97, forward :99-129, load_pretrained :131-154)
 * as called from pocolib/core/tester.py:75-98 (build) and :213,:408 (output = self.model(batch)). */

typedef struct poco_engine* poco_handle_t;

/* Both I/O structs are SELF-DESCRIBING (ABI 4): the first member is the size in bytes of the struct the CALLER compiled or
 * declared (`x.struct_size = sizeof x;`).  The library reads exactly that many bytes:
 *   - members beyond `struct_size` (added by a later header) read as NULL = "not wanted / not given";
 *   - a `struct_size` that is 0, not a multiple of 8, smaller than the size word plus one pointer, or larger than 4096 is
 *     POCO_ERR_ARG (that is what a struct written from an older header - whose first member is a pointer - or an
 *     uninitialised one looks like), and so is a non-NULL member beyond what THIS library knows;
 * so a binding whose field list is shorter than this header's can no longer make the engine read past its struct. */

/* batch dict of pocolib/core/tester.py:205-212 (device pointers, fp32).  bbox_info/focal_length/
 * scale/center/orig_shape are only read by the *-cliff variants (poco.py:102-111). */
typedef struct {
  uint64_t struct_size;      /* = sizeof(poco_inputs_t) as the caller sees it           */
  const float* img;          /* [B,3,224,224] NCHW, ImageNet-normalised crop            */
  const float* bbox_info;    /* [B,3]   image_utils.py:171-183                           */
  const float* focal_length; /* [B]     image_utils.py:185-187                           */
  const float* scale;        /* [B]     bbox size / 200                                  */
  const float* center;       /* [B,2]   bbox centre (x,y) in full-image pixels           */
  const float* orig_shape;   /* [B,2]   (img_h, img_w)                                   */
} poco_inputs_t;

/* output dict of POCO.forward.  Any pointer may be NULL = not wanted
 * (pred_cam_t / smpl_joints2d / pred_fullimg_cam_t are computed into scratch then). */
typedef struct {
  uint64_t struct_size;      /* = sizeof(poco_outputs_t) as the caller sees it          */
  float* pred_pose;          /* [B,24,3,3] rotation matrices                             */
  float* pred_pose6d;        /* [B,144]    ('pred_pose6d' pare / 'pred_pose_6d' cliff)   */
  float* pred_shape;         /* [B,10]                                                   */
  float* pred_cam;           /* [B,3]  weak-perspective (s,tx,ty)                        */
  float* pred_cam_t;         /* [B,3]  crop camera translation                           */
  float* pred_fullimg_cam_t; /* [B,3]  cliff only                                        */
  float* smpl_vertices;      /* [B,6890,3]                                               */
  float* smpl_joints3d;      /* [B,49,3]                                                 */
  float* smpl_joints2d;      /* [B,49,2]  pare: crop-normalised, cliff: full-image px    */
  float* var_pose;           /* [B,24]  per-joint uncertainty (poco_head.py:144-148)     */
  float* uncert_feat;        /* [B,3072] pare / [B,2048] cliff                           */
  float* pred_segm_mask;     /* [B,25,56,56] NCHW, pare only                             */
  float* body_feat2;         /* [B,1024] cliff only                                      */
  float* backbone_feat;      /* [B,480,56,56] NCHW, hrnet_w32 only: the backbone's output map  */
                             /* (hrnet.py:515-519), for parity checks; NULL in production      */
  float* record;             /* [B,254] packed per-crop record = the payload of the multi-GPU all-gather and of the streaming
                              * D2H copy: [pred_pose 216 | pred_shape 10 | pred_cam 3 | var_pose 24 (raw) | confidence 1], written
                              * by one kernel inside the forward (hipGraph-capturable).  confidence = the reference's
                              * post-processed scalar: kinematic accumulation along the SMPL tree (poco_utils.py:21-25), rows whose
                              * root exceeds the threshold set to 1, cliff: root value / pare: mean over joints (poco_utils.py:50-60),
                              * clipped to [0, 0.99] (tester.py:245).  Options record_kinematic / record_thr of poco_create_ex.  */
} poco_outputs_t;

/* variant = "<backbone>-<head>" exactly as POCO.BACKBONE in the yaml (poco.py:41):
 * "hrnet_w32-pare", "hrnet_w48_cls-cliff", "resnet50-cliff".  num_flow_layers = POCO.NUM_FLOW_LAYERS.
 * Host only (no GPU needed): builds the list of tensors the variant expects. */
int poco_create(const char* variant, int max_batch, int num_flow_layers, poco_handle_t* out);
/* The same with build options: "key=value,key=value" (NULL or "" = defaults = poco_create).  The library reads NO environment
 * variable; every alternative form is chosen here, computes the same model and is compared with the default in tests/:
 *   kcat, kmerge, chain, dual, wg_fuse = 0   the separate-launch form of a fused op group (csrc/engine.hip EngineOpts)
 *   dual_layout = <100 NI + 10 WM + WN>      ResNet-50's two-source shortcut GEMM: load schedule / waves per block (default 41; 0 = one-wave blocks)
 *   stem_mfma = 0                            the stem conv on the packed-FMA kernels instead of the MFMA implicit GEMM (csrc/stem_mfma.hip)
 *   mlp_fuse = 0, mlp_blocks = <1..256>      cliff head: the regressor chain as separate launches instead of the one persistent launch of
 *                                            csrc/mlp_chain.hip / the number of blocks of that launch (default 256)
 *   xdep, tail_lanes, up_lanes = 0           the more conservative lane schedules;  seq_phases = <bit mask>, branch_lanes = "0123"
 *   split_f16 = 1                            EXPERIMENT, only in libraries built with `python -m poco_amd.build --experiments` (an unknown
 *                                            key in the shipped one): plain 1x1 convs in split fp16
 *   w4_min_plane = <7..14>                   smallest plane the whole-position F(4x4) kernel (ALG 13) is prepared for (default 7; 14 = no fragments for the 7x7 planes)
 *   debug_wait_spins = <n>, debug_mlp_timeouts = <n>   TEST HOOKS for poco_status: poll bound of the in-kernel waits (default 2^21 polls,
 *                                            ~3 s) / the first n launches of the fused regressor time out on purpose
 *   flow_ctx_rows = <n>                      context rows poco_realnvp*'s scratch is planned for at finalize (default max_batch)
 *   record_kinematic = 0|1, record_thr = <f> post-processing of poco_outputs_t.record's confidence (defaults 1, 0.40)
 * Unknown keys are an error. */
int poco_create_ex(const char* variant, int max_batch, int num_flow_layers, const char* options, poco_handle_t* out);
void poco_destroy(poco_handle_t h);

/* Expected tensors: names are the reference state_dict keys after the prefix stripping of
 * pocolib/utils/train_utils.py:69-90 ("backbone.", "head.", "uncert_head.", "flow_head.") plus
 * "smpl." for the body model (smpl_head.py:40).  required=0: tolerated but unused by forward. */
int poco_num_tensors(poco_handle_t h);
int poco_tensor_info(poco_handle_t h, int i, char* name, size_t name_cap, int64_t* shape6, int* rank,
                     int* required);
/* Copy one tensor (host fp32) into the engine.  Strict: unknown names and wrong element counts are
 * errors (the reference's silent strict->non-strict fallback, train_utils.py:118-124, is not kept). */
int poco_load_tensor(poco_handle_t h, const char* name, const float* host_data, const int64_t* shape,
                     int rank);
/* BN folding, MFMA weight packing, upload, workspace planning + allocation.  Needs the GPU. */
int poco_finalize(poco_handle_t h);
/* Enqueue one forward pass for B <= max_batch crops on `stream`.  No allocation, no sync. */
int poco_forward(poco_handle_t h, int B, const poco_inputs_t* in, const poco_outputs_t* out, void* stream);

/* Deferred failures of the forwards enqueued since the last call.  poco_forward only ENQUEUES; two of its kernels contain bounded waits
 * (the grid barrier of the fused CLIFF regressor, csrc/mlp_chain.hip, and the partial hand-off of the stream-K 1x1 GEMM, ALG 14): one
 * that runs out ends its launch early - the queue stays alive, the outputs of that forward are invalid - and raises a word in pinned
 * host memory.  Call this AFTER synchronising the stream (or the event) the forward, or the hipGraph replay of it, was enqueued on and
 * BEFORE trusting its outputs:
 *   POCO_OK       nothing timed out;
 *   POCO_ERR_HIP  a wait timed out (poco_last_error says so).  The word is CLEARED and the device-side state re-armed: the next
 *                 poco_forward / graph replay runs normally.
 * Without this call the failure is still not silent for ever: the next poco_forward refuses to enqueue (POCO_ERR_HIP) until
 * poco_status has been asked - but the forward that failed has already returned POCO_OK, and a hipGraph replay never re-enters
 * poco_forward, so bindings must ask here (poco_amd/model.py POCO.check_status; tester.py and stream.py do before results are
 * written).  Cost on the good path: one read of host memory.  Replaces nothing in the reference (PyTorch raises from the
 * synchronising call); SURVEY.md 8(b): "int status codes, never throw". */
int poco_status(poco_handle_t h);

/* Independent branches of the network (HRNet branches, fuse terms, head branches) are enqueued on up
 * to 4 HIP streams forked from / joined to `stream` (default 4; 1 = everything on `stream`). */
int poco_set_num_lanes(poco_handle_t h, int n);

/* Introspection / tuning. */
int poco_num_ops(poco_handle_t h);
int poco_op_info(poco_handle_t h, int i, char* name, size_t name_cap, double* flops_per_crop, int* type);
/* Schedule of op i (available after poco_create, no GPU needed): sched[0] = phase (parallel region; regions are separated by joins of
 * all lanes), [1] = lane (HIP stream inside the region), [2] = bit mask of the lanes of the region this op additionally waits for
 * through events (everything enqueued on them before it), [3] = number of reads, then (activation id, first channel, end channel)
 * per read, then the number of writes and the same triples.  At most `cap` ints are written (cap >= 40 always suffices).  Lets a
 * test check, without a GPU, that no op reads channels another lane writes in the same region without waiting for it. */
int poco_op_sched(poco_handle_t h, int i, int* sched, int cap);
int poco_profile_ops(poco_handle_t h, int B, const poco_inputs_t* in, const poco_outputs_t* out, int iters,
                     float* ms_per_op, int cap, void* stream);
size_t poco_workspace_bytes(poco_handle_t h);
int poco_uncert_feat_dim(poco_handle_t h);
int poco_set_conv_cfg(poco_handle_t h, int op_index, int B, const int* cfg7);
int poco_get_conv_desc(poco_handle_t h, int op_index, int* desc8);
/* the tile configuration (7 ints, csrc/common.h) conv op `op_index` runs with at batch size B: the value set by
 * poco_set_conv_cfg or, without one, the built-in heuristic's choice. */
int poco_get_conv_cfg(poco_handle_t h, int op_index, int B, int* cfg7);

/* SMPL linear blend skinning with the engine's loaded body model: replaces
 * smplx.SMPL.forward(pose2rot=False) as wrapped by pocolib/models/head/smpl_head.py:22-34.
 * d_betas [B,10], d_rotmat [B,24,3,3] -> d_verts [B,6890,3], d_joints49 [B,49,3]. */
int poco_smpl_lbs(poco_handle_t h, int B, const float* d_betas, const float* d_rotmat, float* d_verts,
                  float* d_joints49, void* stream);
/* RealNVP with the engine's flow_head.flow.* weights (pocolib/models/layers/real_nvp.py):
 * forward=0: log_prob(x[N,9] | ctx[N,512]) -> out[N]   (:55-65)
 * forward=1: forward_p(z[N,9], ctx)        -> out[N,9] (:25-38)
 * Two launches: one GEMM for the context part of the first Linear of all 2L s/t MLPs, one fp32-MFMA kernel for the coupling
 * recursion (csrc/kernels_flow.hip).  The GEMM's scratch is planned at poco_finalize for `flow_ctx_rows` context rows (option of
 * poco_create_ex; default max_batch = one context per crop): no allocation, no synchronisation here; more context rows than
 * planned is POCO_ERR_ARG. */
int poco_realnvp(poco_handle_t h, int N, const float* d_x, const float* d_ctx, float* d_out, int forward,
                 void* stream);
/* The same with the context as the reference's flow_head actually has it (pocolib/models/head/nf_head.py:93-110): one
 * 512-vector per CROP, used by `rep` consecutive rows (rep = 24 joints) - torch.repeat_interleave(context, rep) is never
 * materialised.  d_ctx [ceil(N/rep), 512]; row r uses d_ctx[r / rep].  rep = 1 is poco_realnvp. */
int poco_realnvp_rep(poco_handle_t h, int N, const float* d_x, const float* d_ctx, int rep, float* d_out, int forward,
                     void* stream);

/* ---- stand-alone operators (parity tests, tuner, micro-benchmarks) -------------------------- */

/* conv(ks x ks, stride, pad=(ks-1)/2, no groups/dilation) * scale[co] + shift[co] (+ residual) (ReLU)
 * == nn.Conv2d -> nn.BatchNorm2d(eval) [-> "+= residual"] [-> ReLU] of
 * pocolib/models/backbone/hrnet.py:42-58,79-99 / hrnet_cls.py / resnet.py:101-121.
 * d_in  [B,H,Cin/16,W,16] (L16), h_weight [Cout,Cin,ks,ks] (host, torch OIHW order),
 * h_scale/h_shift [Cout] (host, nullable), d_res / d_out [B,Ho,Cout/16,Wo,16] (L16; d_res nullable).
 * Cin and Cout must be multiples of 16.  cfg = 7 ints {MT,NT,WM,WN,R,NI,ALG} (csrc/common.h) or NULL for the
 * heuristic. */
int poco_op_conv2d(const float* d_in, int B, int H, int W, int Cin, const float* h_weight,
                   const float* h_scale, const float* h_shift, int Cout, int ks, int stride,
                   const float* d_res, int relu, float* d_out, const int* cfg7, void* stream);

/* Time `iters` launches of the same conv (+ReLU) with hipEvents; ms_out = mean ms per launch.
 * cfg_used7 (nullable) receives the tile configuration that ran: SEVEN ints {MT,NT,WM,WN,R,NI,ALG} (csrc/common.h
 * CONV_CFG_INTS) - the caller's array must hold 7. */
int poco_bench_conv2d(const float* d_in, int B, int H, int W, int Cin, const float* h_weight, int Cout,
                      int ks, int stride, float* d_out, const int* cfg7, int iters, float* ms_out,
                      int* cfg_used7, void* stream);

/* Part-attention pooling == pocolib/models/layers/keypoint_attention.py:34-48 (KeypointAttention.forward with
 * use_conv = False, softmax over the pixels) as pare_head.py:794-796 calls it (heat-map channel 0 = background is skipped):
 *   out[b, c, j] = sum_p softmax_p(heat[b, p, 1 + j]) * feat[b, p, c],  j = 0..23.
 * d_heat [B,H,heat_cs/16,W,16] (L16, heat_cs >= 32: channels 1..24 are the parts), d_feat [B,H,C/16,W,16] (L16, C a multiple of
 * 16, <= 128), d_out [B, C, 24]. */
int poco_op_part_attention(const float* d_heat, int heat_cs, const float* d_feat, int C, int B, int H, int W,
                           float* d_out, void* stream);
/* Timing of the same pool: `iters` back-to-back launches between two HIP events on `stream`; ms_out = mean ms per launch
 * (both kernels of the split-pixel softmax pool).  bench.py's `side_kernels` line. */
int poco_bench_part_attention(const float* d_heat, int heat_cs, const float* d_feat, int C, int B, int H, int W,
                              float* d_out, int iters, float* ms_out, void* stream);
/* LocallyConnected2d(128 -> 6, output_size [24,1], kernel 1, no bias) == pocolib/models/layers/locallyconnected2d.py:27-37
 * as pare_head.py builds its pose_mlp: pose6d[b, j, o] = sum_c x[b, c, j] * w[o, c, j].
 * d_x [B,128,24], d_w [6,128,24], d_pose6d [B,24,6]. */
int poco_op_lc2d_pose(const float* d_x, const float* d_w, float* d_pose6d, int B, void* stream);
/* rot6d_to_rotmat == pocolib/utils/geometry.py:247-261: d_in [B,24,3,2] (144 floats per crop) -> d_rotmat [B,24,3,3]. */
int poco_op_rot6d(const float* d_in, float* d_rotmat, int B, void* stream);

/* GPU-side crop + normalise: replaces the per-detection CPU loop cv2.getAffineTransform -> cv2.warpAffine(INTER_LINEAR,
 * BORDER_CONSTANT) -> ToTensor -> Normalize + per-crop H2D copy of pocolib/core/tester.py:182-203 and
 * pocolib/utils/vibe_image_utils.py:58-107,233-266,343-351.  The uint8 crop underneath is BYTE-exact with OpenCV 4.5.5's
 * fixed-point algorithm for that call (requirements.txt:4; restated in oracle/crop_np.py): 6x6 LU solve + inversion in
 * double, 10-bit fixed-point coordinates, 1/32-px fractions, int16 weights, rounded 15-bit shift.
 * d_frame uint8 [H,W,3] RGB, d_boxes [N,4] (cx,cy,w,h) px (float32, or float64 for the _f64 entry: the reference's
 * detections / smoothed tracks may be either), bbox_scale = the Python float `scale` of get_single_image_crop_demo,
 * d_out [N,3,res,res] fp32 NCHW. */
int poco_crop_normalize(const unsigned char* d_frame, int H, int W, const float* d_boxes, int N, double bbox_scale,
                        int res, float* d_out, void* stream);
int poco_crop_normalize_f64(const unsigned char* d_frame, int H, int W, const double* d_boxes, int N, double bbox_scale,
                            int res, float* d_out, void* stream);
/* The crops of a whole batch in ONE launch when they come from several frames of the same size (video / streaming mode:
 * tester.py:399-408 crops per frame): d_frames = device array of `nframes` device pointers to uint8 [H,W,3] frames,
 * d_frame_idx [N] int32 = which of them crop n is cut from (device data: cannot be validated on the host; the kernel clamps an
 * index into 0 .. nframes - 1, so a bad one reads the wrong frame, never a wild pointer).  Same arithmetic as poco_crop_normalize
 * (float32 boxes). */
int poco_crop_normalize_multi(const unsigned char* const* d_frames, int nframes, const int* d_frame_idx, int H, int W,
                              const float* d_boxes, int N, double bbox_scale, int res, float* d_out, void* stream);

/* Time `ncfg` tile configurations (cfgs7 = ncfg x SEVEN ints {MT,NT,WM,WN,R,NI,ALG} each, csrc/common.h CONV_CFG_INTS;
 * MT<=0 = heuristic) for one conv shape on random data; ms_out[i] < 0 = configuration invalid for this shape.  NULL cfgs7 /
 * ms_out, ncfg < 1 or channel counts that are not multiples of 16 are POCO_ERR_ARG.  iters < 0: |iters| launches of the RESIDUAL form (the
 * input doubles as the residual: conv2 of a BasicBlock; needs Cin == Cout, stride 1).  Used by poco_amd/tune.py. */
int poco_tune_conv(int B, int H, int W, int Cin, int Cout, int ks, int stride, const int* cfgs7, int ncfg,
                   int iters, float* ms_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* POCO_HIP_H */
