// The POCO engine: builds the per-crop regressor (backbone -> head -> SMPL -> confidence MLP) as a
// linear program of HIP kernel launches over a statically planned HBM workspace.
//
// Replaces  POCO.__init__/forward/load_pretrained  (pocolib/models/poco.py:13-154).  Tensors are
// addressed by the reference's own state_dict keys (after the prefix stripping of
// pocolib/utils/train_utils.py:69-90):  backbone.*, head.*, uncert_head.*, flow_head.*  plus the
// SMPL body model as smpl.* (the reference loads that from data/smpl, smpl_head.py:40).
//
// Life cycle (include/poco_hip.h): create (host only; declares every tensor the variant needs) ->
// load_tensor* (host copies) -> finalize (strict key check, BN folding, MFMA weight packing,
// upload, workspace planning/allocation) -> forward* (kernel launches only: no allocation, no sync,
// everything on the caller's stream, hipGraph-capturable) -> destroy.
#include "../../include/poco_hip.h"
#include "kernels.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <set>
#include <vector>

namespace {

constexpr float BN_EPS = 1e-5f;

struct HostParam {
  std::vector<int64_t> shape;
  std::vector<float> data;
};
struct ParamDecl {
  std::string name;
  std::vector<int64_t> shape;
  int required;   // 1 = used by forward; 0 = tolerated (present in reference checkpoints, unused)
};

enum OpType {
  OP_STEM, OP_CONV, OP_MAXPOOL, OP_BILINEAR, OP_FUSE, OP_AVGPOOL, OP_ATTN, OP_LC2D, OP_ROT6D, OP_COPY,
  OP_BCAST, OP_SMPL, OP_CAMERA, OP_NCHW_OUT, OP_CHAIN, OP_DUAL1X1, OP_RECORD, OP_MLP
};

// external buffer slots (inputs / outputs of poco_forward)
enum Ext {
  X_NONE = 0, X_IMG, X_BBOX, X_FOCAL, X_SCALE, X_CENTER, X_ORIG,
  Y_POSE, Y_POSE6D, Y_SHAPE, Y_CAM, Y_CAM_T, Y_FULL_CAM_T, Y_VERTS, Y_J3D, Y_J2D, Y_VAR, Y_UFEAT, Y_SEGM, Y_BBFEAT,
  Y_BODY2, Y_RECORD
};

struct Act {
  int C = 0, H = 1, W = 1;
  size_t off = 0;       // floats from workspace base
  int first = 1 << 30, last = -1;
  bool persistent = false;
  size_t per_crop() const { return (size_t)C * H * W; }
};

struct Ref {            // a (possibly strided) view: activation + channel offset, or external slot
  int act = -1;
  int co = 0;
  int ext = X_NONE;
};

struct Op {
  int type = 0;
  int phase = 0, lane = 0;
  unsigned wait_mask = 0;   // lanes of its phase this op waits for (everything enqueued on them so far): cross-lane dependencies
  int dep_ev = -1;          // first of the events that carry them (one per set bit)
  std::string name;
  double flops = 0;     // per crop
  Ref in, in2, res, out, out2;
  // conv
  int Cin = 0, Cout = 0, ks = 1, stride = 1, actfn = 0, res_after = 0, relu_from = 0;
  float* wdev = nullptr;
  float* bdev = nullptr;
  float* wdev_wino = nullptr;   // 3x3 stride-1 convs: Winograd-transformed weights (ALG 3)
  float* wdev_wino4p = nullptr; // the same in the LDS order of ALG 8
  float* wdev_wino4w = nullptr; // ... and in the quad order of ALG 13 (conv_wino4w.hip)
  float* wdev_wino4g = nullptr; // planes <= 8x8: per-position GEMM fragments of ALG 11 (conv_wino4g.hip)
  float* wdev_h = nullptr;      // POCO_SPLIT_F16=1 only: hi / lo fp16 halves of a plain 1x1 conv's weights (ALG 12 experiment)
  float* wdev_wino4 = nullptr;  // 3x3 stride-1 convs on planes >= 28x28: F(4x4,3x3) fragments (ALG 7)
  float* wdev2 = nullptr;       // OP_CHAIN: the second 1x1 conv (next block's conv1)
  float* bdev2 = nullptr;
  // fuse
  Ref fsrc[4];        // terms may be channel slices of wider (merged-conv) buffers
  int fshift[4] = {0, 0, 0, 0};
  int fn = 0, frelu = 1;
  // misc ints
  int n = 0, C = 0;
  std::map<int, ConvCfg> cfg;   // per batch size
  // OP_MLP (mlp_chain.hip): the Linear layers (OP_CONV), row copies / broadcasts (OP_COPY / OP_BCAST) and rot6d (OP_ROT6D) of the
  // regressor chain as sub-ops in stage order; `stage` = the sub-op's stage
  std::vector<Op> sub;
  int stage = 0;
};

// Build options (poco_create_ex): the A/B forms of the schedule and of the fused ops.  Every form computes the same model (tests
// compare them); they are chosen explicitly through the C ABI - the library reads no environment variable.
struct EngineOpts {
  bool kcat = true;        // layer1.0: bn3(conv3(t)) + bn_d(conv_d(x)) as one 1x1 conv over [t ; x]
  bool kmerge = true;      // HR module: the lowest-resolution branch's fuse sum as one K-concatenated stride-2 conv
  bool chain = true;       // layer1: conv3 + residual + ReLU of block k chained with conv1 of block k+1 (bneck_chain.hip)
  bool dual = true;        // ResNet-50 layer2-4.0: conv3 + stride-2 projection shortcut as one two-source GEMM
  bool xdep = true;        // one join per HR module, open stage boundaries, inferred cross-lane events
  bool tail_lanes = true;  // SMPL-LBS + camera | output copies + confidence MLP on two lanes (PARE: the two head branches too)
  bool up_lanes = true;    // PARE: the three upsample chains continue on their branches' lanes
  bool wg_fuse = true;     // ALG 11: consecutive convs of a lane chained through wg_mid_kernel
  // two-source GEMM (layer2-4 .0 of ResNet-50): 100 NI + 10 WM + WN, 0 = one wave per block.  One-wave blocks were the fastest layout in the
  // round-2 probe (204 us for layer4.0) but their time depends on how the dispatcher spreads 896 single waves over the 1024 SIMDs - hidden state
  // left by the kernels before: with other configurations of the 14x14 convs five launches earlier the same launch takes 343 us (round 5,
  // DESIGN.md 8).  Four-wave blocks (one wave per SIMD by construction) cost ~3 us per launch and do not have that cliff.
  int dual_layout = 41;
  bool stem_mfma = true;   // stem conv as an implicit GEMM on the MFMA (stem_mfma.hip) instead of the packed-FMA kernels of kernels_misc.hip
  bool mlp_fuse = true;    // CLIFF regressor (fc1 / fc2 / decoders x 3 iterations, state scatter, rot6d) as one persistent launch (mlp_chain.hip)
  int mlp_blocks = 256;    // ... on at most this many blocks (one grid barrier per stage: fewer blocks = cheaper barriers, more = more jobs at once)
  bool split_f16 = false;  // EXPERIMENT (never the default): plain 1x1 convs on the split-fp16 GEMM (gemm1x1h.hip)
  int seq_mask = 0;        // bit mask: run the tagged kind of parallel region on one lane
  std::string branch_lanes = "0123";   // HR branch i runs on lane branch_lanes[i]
  int w4_min_plane = 7;    // ALG 13 (F(4x4), blocks own tiles x all positions) is offered for planes from this size up; ALG 8 from 14x14.  Round 6: the 7x7 planes of
                           // HRNet-W48 run on ALG 13 at >= 64 crops (rectangular items of 8 whole images: time +-0 against ALG 11, which stages V and M through memory -
                           // 128 MB per conv); 14 = round-5 behaviour (no ALG 13 fragments for the 7x7 convs: 21 MB each at 384->384)
  int wg_max_plane = 8;    // ALG 11 (F(4x4) as 36 position GEMMs, V / M staged in memory) is offered for planes up to this size (<= 16)
  int flow_ctx_rows = 0;   // context rows the RealNVP scratch is planned for at finalize (0 = max_batch: one context per crop)
  int debug_wait_spins = 0;    // TEST HOOKS for poco_status (tests/test_model_gpu.py): poll bound of the in-kernel waits (0 = 2^21 polls ~ 3 s) ...
  int debug_mlp_timeouts = 0;  // ... and: the first n launches of the fused regressor leave one block out of their first grid barrier, i.e. time out
  int rec_kinematic = 1;   // poco_outputs_t.record: kinematic accumulation of the per-joint uncertainty (KINEMATIC_UNCERT)
  float rec_thr = 0.40f;   // ... and the sensitivity threshold of get_global_uncert (poco_utils.py:50)
};

static bool parse_opts(const char* str, EngineOpts* o, std::string* err) {
  if (!str) return true;
  std::string s(str);
  size_t i = 0;
  while (i < s.size()) {
    size_t j = s.find(',', i);
    if (j == std::string::npos) j = s.size();
    std::string kv = s.substr(i, j - i);
    i = j + 1;
    if (kv.empty()) continue;
    const size_t eq = kv.find('=');
    const std::string k = kv.substr(0, eq), v = eq == std::string::npos ? "1" : kv.substr(eq + 1);
    const bool on = v != "0";
    if (k == "kcat") o->kcat = on;
    else if (k == "kmerge") o->kmerge = on;
    else if (k == "chain") o->chain = on;
    else if (k == "dual") o->dual = on;
    else if (k == "xdep") o->xdep = on;
    else if (k == "tail_lanes") o->tail_lanes = on;
    else if (k == "up_lanes") o->up_lanes = on;
    else if (k == "wg_fuse") o->wg_fuse = on;
    else if (k == "mlp_fuse") o->mlp_fuse = on;
    else if (k == "stem_mfma") o->stem_mfma = on;
    else if (k == "dual_layout") o->dual_layout = atoi(v.c_str());
    else if (k == "mlp_blocks") o->mlp_blocks = std::min(256, std::max(1, atoi(v.c_str())));
#if POCO_EXPERIMENTS
    else if (k == "split_f16") o->split_f16 = on;
#endif
    else if (k == "seq_phases") o->seq_mask = atoi(v.c_str());
    else if (k == "branch_lanes") o->branch_lanes = v;
    else if (k == "flow_ctx_rows") o->flow_ctx_rows = atoi(v.c_str());
    else if (k == "w4_min_plane") o->w4_min_plane = std::min(14, std::max(4, atoi(v.c_str())));
    else if (k == "wg_max_plane") o->wg_max_plane = std::min(16, std::max(1, atoi(v.c_str())));
    else if (k == "debug_wait_spins") o->debug_wait_spins = std::max(0, atoi(v.c_str()));
    else if (k == "debug_mlp_timeouts") o->debug_mlp_timeouts = std::max(0, atoi(v.c_str()));
    else if (k == "record_kinematic") o->rec_kinematic = on;
    else if (k == "record_thr") o->rec_thr = (float)atof(v.c_str());
    else { *err = "unknown engine option '" + k + "'"; return false; }
  }
  for (char c : o->branch_lanes)
    if (c < '0' || c > '3') { *err = "branch_lanes must be digits 0..3"; return false; }
  return true;
}

struct Engine {
  EngineOpts opts;
  std::string backbone, head;
  int max_batch = 0;
  int flow_layers = 0;
  bool finalized = false;
  std::vector<ParamDecl> decls;
  std::map<std::string, size_t> decl_index;
  std::map<std::string, HostParam> params;
  // graph
  std::vector<Act> acts;
  std::vector<Op> ops;
  float* ws = nullptr;
  size_t ws_floats = 0;
  std::vector<void*> dev_allocs;
  int num_lanes = 4;      // 1 = run everything on the caller's stream
  hipStream_t lane_stream[4] = {};   // side lanes 1..3 (lane 0 is the caller's stream)
  hipEvent_t ev_fork = nullptr, ev_join[4] = {};
  std::vector<hipEvent_t> ev_dep;   // one per op with a cross-lane dependency (Op::wait_lane)
  // SMPL / flow device models
  SmplDev smpl{};
  int a_coef = -1, a_A = -1, a_j24 = -1, a_verts = -1, a_j49 = -1, a_attn_scratch = -1, a_camt = -1, a_fullt = -1, a_j2d = -1;
  Ref smpl_betas, smpl_rot, cam_ref;
  FlowDev flow{};
  bool has_flow = false;
  float* wino4g_scratch[4] = {};          // ALG 11 (V + M staging): one buffer per lane, sized at finalize for max_batch
  int wg_ready_act[4] = {-1, -1, -1, -1}; // ALG 11 chaining: activation whose V the previous conv of the lane left in scratch ...
  int wg_ready_vsel[4] = {0, 0, 0, 0};    // ... and in which half
  std::vector<int> act_uses;              // how many op inputs / residuals / fuse terms read each activation (built at finalize)
  size_t wino4g_scratch_need = 0;
  float* flow_scratch = nullptr;          // step A of the flow (context GEMM): planned at finalize for opts.flow_ctx_rows context rows
  size_t flow_scratch_floats = 0;
  int uncert_feat_dim = 0;
  float* sk_scratch[4] = {};          // ALG 14 (stream-K 1x1 GEMM): flags + partials, one buffer per lane
  unsigned* mlp_sync = nullptr;       // OP_MLP: grid-barrier counters (device) ...
  unsigned* mlp_err_host = nullptr;   // ... and the time-out word (pinned host memory the kernels can write; shared with ALG 14): poco_status reads and clears it
  int mlp_timeouts_left = 0;          // test hook (option debug_mlp_timeouts)
  std::string err;

  ~Engine() {
    for (int k = 1; k < 4; ++k) {
      if (lane_stream[k]) (void)hipStreamDestroy(lane_stream[k]);
      if (ev_join[k]) (void)hipEventDestroy(ev_join[k]);
    }
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    for (hipEvent_t ev : ev_dep) (void)hipEventDestroy(ev);
    for (void* p : dev_allocs) (void)hipFree(p);
    if (ws) (void)hipFree(ws);
    if (flow_scratch) (void)hipFree(flow_scratch);
    if (mlp_sync) (void)hipFree(mlp_sync);
    for (float* q : sk_scratch) if (q) (void)hipFree(q);
    if (mlp_err_host) (void)hipHostFree(mlp_err_host);
    for (float* q : wino4g_scratch) if (q) (void)hipFree(q);
  }
};

// ------------------------------------------------------------------------------------------------
// Builder: the same code path declares the tensors (declare=true, host only) and builds the
// launch program (declare=false, uploads packed weights).
// ------------------------------------------------------------------------------------------------
// What an op reads and writes, as channel ranges of activations (concat buffers are written and read by slices): the one
// description the builder's dependency inference (Builder::push) and the schedule introspection (poco_op_sched) share.
struct OpAccess { int act, lo, hi; };
static void op_accesses(const Engine& e, const Op& op, std::vector<OpAccess>* rd, std::vector<OpAccess>* wr) {
  constexpr int ALL = 1 << 30;
  auto add = [](std::vector<OpAccess>* to, const Ref& r, int n) {
    if (r.act >= 0) to->push_back({r.act, n == ALL ? 0 : r.co, n == ALL ? ALL : r.co + n});
  };
  const bool conv = op.type == OP_CONV;
  add(rd, op.in, conv ? op.Cin : ALL);
  add(rd, op.in2, ALL);
  add(rd, op.res, conv ? op.Cout : ALL);
  for (int k = 0; k < op.fn && k < 4; ++k) add(rd, op.fsrc[k], op.type == OP_FUSE ? op.C : ALL);
  if (op.type == OP_SMPL || op.type == OP_CAMERA || op.type == OP_RECORD) { add(rd, e.smpl_betas, ALL); add(rd, e.smpl_rot, ALL); add(rd, e.cam_ref, ALL); }   // implicit operands
  add(wr, op.out, conv ? op.Cout : (op.type == OP_FUSE ? op.C : ALL));
  add(wr, op.out2, ALL);
  for (const Op& su : op.sub) op_accesses(e, su, rd, wr);
}

struct Builder {
  Engine& e;
  bool declare;
  bool ok = true;

  explicit Builder(Engine& eng, bool decl) : e(eng), declare(decl) {}

  const HostParam* P(const std::string& name, std::vector<int64_t> shape, int required = 1) {
    if (declare) {
      if (!e.decl_index.count(name)) {
        e.decl_index[name] = e.decls.size();
        e.decls.push_back({name, shape, required});
      }
      return nullptr;
    }
    auto it = e.params.find(name);
    if (it == e.params.end()) {
      if (required) { ok = false; e.err += "missing tensor " + name + "; "; }
      return nullptr;
    }
    return &it->second;
  }

  float* upload(const std::vector<float>& h) {
    float* d = nullptr;
    if (hipMalloc(&d, std::max<size_t>(h.size(), 4) * sizeof(float)) != hipSuccess) { ok = false; e.err += "hipMalloc failed; "; return nullptr; }
    if (hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { ok = false; e.err += "H2D failed; "; }
    e.dev_allocs.push_back(d);
    return d;
  }
  int* upload_i(const std::vector<int>& h) {
    int* d = nullptr;
    if (hipMalloc(&d, std::max<size_t>(h.size(), 4) * sizeof(int)) != hipSuccess) { ok = false; return nullptr; }
    if (hipMemcpy(d, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) ok = false;
    e.dev_allocs.push_back(d);
    return d;
  }

  int new_act(int C, int H, int W, bool persistent = false) {
    Act a; a.C = C; a.H = H; a.W = W; a.persistent = persistent;
    e.acts.push_back(a);
    return (int)e.acts.size() - 1;
  }
  static Ref R(int act, int co = 0) { Ref r; r.act = act; r.co = co; return r; }
  static Ref X(int ext) { Ref r; r.ext = ext; return r; }

  // Scheduling model: the program is a sequence of PHASES separated by full joins; inside a phase
  // ops are split into LANES (HIP streams) that run concurrently, each lane in program order.
  // Outside a parallel region every op is its own single-lane phase.
  int cur_phase = -1, cur_lane = 0;
  bool in_parallel = false;
  // A/B forms (EngineOpts, poco_create_ex): seq_mask runs the tagged kind of parallel region on one lane; kcat / kmerge / chain /
  // dual select the fused or the separate form of an op group; xdep = one join per HR module with the stage boundaries inside
  // ONE parallel region (the transition conv waits for the K-merged conv's lane through an event instead of a join of all lanes)
  int seq_mask = e.opts.seq_mask;
  bool region_seq = false;
  bool kcat = e.opts.kcat, kmerge = e.opts.kmerge, chain = e.opts.chain, dual = e.opts.dual, xdep = e.opts.xdep;
  bool tail_lanes = e.opts.tail_lanes;
  // Cross-lane dependencies are INFERRED, not declared: inside a region the builder remembers which lanes wrote (a slice of) every
  // activation and up to which op a lane has already synchronised with every other lane; an op that reads an activation written
  // on another lane after that point gets that lane in its wait mask.  (Activations are written once per region - concat
  // buffers by slices - and the planner recycles memory only across regions, so read-after-write is the only hazard.)
  struct SliceW { int lo, hi, lane, idx; };   // channels [lo, hi) of an activation written by op idx on that lane
  std::vector<std::vector<SliceW>> act_w;     // [act]: what this region has written so far (concat buffers: several slices)
  int lane_last[4] = {-1, -1, -1, -1};        // last op pushed on each lane in this region
  int lane_seen[4][4];                        // [a][b]: lane a has waited for lane b up to this op index
  void region_reset() {
    act_w.clear();
    for (int a = 0; a < 4; ++a) { lane_last[a] = -1; for (int b2 = 0; b2 < 4; ++b2) lane_seen[a][b2] = -1; }
  }
  void begin_parallel(int kind = 0) {
    ++cur_phase; cur_lane = 0; in_parallel = true; region_seq = (seq_mask >> kind) & 1;
    region_reset();
  }
  void end_parallel() { in_parallel = false; }
  void lane(int k) { cur_lane = region_seq ? 0 : k % 4; }
  std::vector<Op>* capture = nullptr;   // OP_MLP under construction: ops are collected as its sub-ops instead of entering the program
  int cur_stage = 0;
  void push(Op&& op) {
    if (capture) { op.stage = cur_stage; capture->push_back(std::move(op)); return; }
    if (!in_parallel) { ++cur_phase; cur_lane = 0; }
    op.phase = cur_phase;
    op.lane = cur_lane;
    if (in_parallel && !region_seq) {
      const int idx = (int)e.ops.size(), a = cur_lane;
      unsigned mask = 0;
      std::vector<OpAccess> rd, wr;
      op_accesses(e, op, &rd, &wr);
      for (const OpAccess& r : rd) {
        if (r.act >= (int)act_w.size()) continue;
        for (const SliceW& w : act_w[r.act])
          if (w.lane != a && w.lo < r.hi && r.lo < w.hi && w.idx > lane_seen[a][w.lane]) mask |= 1u << w.lane;
      }
      for (int b2 = 0; b2 < 4; ++b2)
        if (mask & (1u << b2)) lane_seen[a][b2] = lane_last[b2];
      op.wait_mask = mask;
      for (const OpAccess& w : wr) {
        if ((int)act_w.size() <= w.act) act_w.resize(w.act + 1);
        act_w[w.act].push_back({w.lo, w.hi, a, idx});
      }
      lane_last[a] = idx;
    }
    e.ops.push_back(std::move(op));
  }

  // BN(eval) folded into per-channel scale/shift:  y = conv*scale + shift
  void bn_fold(const std::string& bn, const HostParam* conv_bias, int Cout, std::vector<float>& scale,
               std::vector<float>& shift) {
    scale.assign(Cout, 1.f);
    shift.assign(Cout, 0.f);
    if (!bn.empty()) {
      const HostParam* g = P(bn + ".weight", {Cout});
      const HostParam* b = P(bn + ".bias", {Cout});
      const HostParam* m = P(bn + ".running_mean", {Cout});
      const HostParam* v = P(bn + ".running_var", {Cout});
      P(bn + ".num_batches_tracked", {}, 0);
      if (declare || !g || !b || !m || !v) return;
      for (int c = 0; c < Cout; ++c) {
        const double s = (double)g->data[c] / std::sqrt((double)v->data[c] + (double)BN_EPS);
        scale[c] = (float)s;
        shift[c] = (float)((double)b->data[c] - (double)m->data[c] * s);
        if (conv_bias) shift[c] = (float)((double)shift[c] + (double)conv_bias->data[c] * s);
      }
    } else if (conv_bias && !declare) {
      for (int c = 0; c < Cout; ++c) shift[c] = conv_bias->data[c];
    }
  }

  // conv (+bias) + BN + act (+ residual).  `in` may be a channel slice; returns the output act
  // (or writes into `into` at channel offset).  kperm: optional K-column permutation for Linear
  // layers whose input vector is laid out differently from the reference's torch.cat.
  int conv(const std::string& name, const std::string& convp, const std::string& bnp, Ref in, int Cin_real,
           int Cout, int ks, int stride, int actfn, bool has_bias, Ref res = Ref(), int res_after = 0,
           Ref into = Ref(), const std::vector<int>* kperm = nullptr, int Cin_padded = 0, bool is_linear = false,
           bool direct = false, const HostParam* w_direct = nullptr, const HostParam* b_direct = nullptr) {
    const Act ain = e.acts[in.act];
    const int Cin = Cin_padded ? Cin_padded : Cin_real;
    const int pad = (ks - 1) / 2;
    const int Ho = (ain.H + 2 * pad - ks) / stride + 1, Wo = (ain.W + 2 * pad - ks) / stride + 1;
    const int Cout16 = (Cout + 15) / 16 * 16;
    std::vector<int64_t> wshape = is_linear ? std::vector<int64_t>{Cout, Cin_real}
                                            : std::vector<int64_t>{Cout, Cin_real, ks, ks};
    const HostParam* w = direct ? w_direct : P(convp + ".weight", wshape);
    const HostParam* cb = direct ? b_direct : (has_bias ? P(convp + ".bias", {Cout}) : nullptr);
    std::vector<float> scale, shift;
    bn_fold(bnp, cb, Cout, scale, shift);
    Op op;
    op.type = OP_CONV; op.name = name;
    op.in = in; op.res = res; op.res_after = res_after;
    op.Cin = Cin; op.Cout = Cout16; op.ks = ks; op.stride = stride; op.actfn = actfn;
    op.flops = 2.0 * Ho * Wo * (double)Cout * Cin_real * ks * ks;
    int out_act;
    if (into.act >= 0) { out_act = into.act; op.out = into; }
    else { out_act = new_act(Cout16, Ho, Wo); op.out = R(out_act); }
    if (!declare && w) {
      std::vector<float> wsrc;
      const float* wp = w->data.data();
      if (kperm || Cin != Cin_real) {        // re-lay the K columns (Linear layers only: ks == 1)
        wsrc.assign((size_t)Cout * Cin, 0.f);
        for (int o = 0; o < Cout; ++o)
          for (int k = 0; k < Cin_real; ++k) {
            const int dst = kperm ? (*kperm)[k] : k;
            if (dst < 0) continue;      // column handled by another (split) GEMM
            wsrc[(size_t)o * Cin + dst] = wp[(size_t)o * Cin_real + k];
          }
        wp = wsrc.data();
      }
      std::vector<float> packed(conv_packed_weight_floats(Cin, Cout16, ks));
      conv_pack_weights(wp, scale.data(), Cout, Cin, ks, Cout16, packed.data());
      std::vector<float> sh(Cout16, 0.f);
      std::copy(shift.begin(), shift.end(), sh.begin());
      op.wdev = upload(packed);
      op.bdev = upload(sh);
#if POCO_EXPERIMENTS
      if (e.opts.split_f16 && ks == 1 && !is_linear && Cin % 32 == 0 && ain.H * ain.W >= 16 && !kperm) {
        std::vector<float> ph(gemm1x1h_packed_floats(Cin, Cout16));
        gemm1x1h_pack_weights(wp, scale.data(), Cout, Cin, Cout16, ph.data());
        op.wdev_h = upload(ph);
      }
#endif
      if (ks == 3 && stride == 1 && !is_linear) {
        std::vector<float> wt, pu(conv_packed_weight_floats(Cin, Cout16, 4));
        conv_wino_transform_weights(wp, Cout, Cin, &wt);
        conv_pack_weights(wt.data(), scale.data(), Cout, Cin, 4, Cout16, pu.data());
        op.wdev_wino = upload(pu);
        if (ain.H >= e.opts.w4_min_plane && ain.W >= e.opts.w4_min_plane) {          // F(4x4,3x3): 56x56 / 28x28 planes, 14x14 (16 tiles per image, 31 % padding), 7x7 with ALG 13 only
          std::vector<float> pu4(conv_wino4_packed_floats(Cin, Cout16));
          if (ain.H >= 28 && ain.W >= 28) {
            conv_wino4_pack_weights(wp, scale.data(), Cout, Cin, Cout16, pu4.data());
            op.wdev_wino4 = upload(pu4);
          }
          if (ain.H >= 14 && ain.W >= 14) {
            pu4.resize(conv_wino4p_packed_floats(Cin, Cout16));                           // other order (+ slack for the streamed requests)
            conv_wino4p_pack_weights(wp, scale.data(), Cout, Cin, Cout16, pu4.data());
            op.wdev_wino4p = upload(pu4);
          }
          pu4.resize(conv_wino4w_packed_floats(Cin, Cout16));                             // ALG 13: whole-position waves (+ slack)
          conv_wino4w_pack_weights(wp, scale.data(), Cout, Cin, Cout16, pu4.data());
          op.wdev_wino4w = upload(pu4);
        }
        if (ain.H <= e.opts.wg_max_plane && ain.W <= e.opts.wg_max_plane && ain.H * ain.W > 1 && actfn <= 1) {      // 7x7 planes: F(4x4,3x3) as 36 position GEMMs (ALG 11)
          std::vector<float> pg(conv_wino4g_packed_floats(Cin, Cout16));
          conv_wino4g_pack_weights(wp, scale.data(), Cout, Cin, Cout16, pg.data());
          op.wdev_wino4g = upload(pg);
          e.wino4g_scratch_need = std::max(e.wino4g_scratch_need, conv_wino4g_scratch_floats(e.max_batch, ain.H, ain.W, Cin, Cout16));
        }
      }
    }
    push(std::move(op));
    return out_act;
  }

  int conv_bn(const std::string& pfx_conv, const std::string& pfx_bn, int in, int Cin, int Cout, int ks, int stride,
              int relu, int res = -1, bool bias = false, int res_after = 0, Ref into = Ref()) {
    return conv(pfx_conv, pfx_conv, pfx_bn, R(in), Cin, Cout, ks, stride, relu ? 1 : 0, bias,
                res >= 0 ? R(res) : Ref(), res_after, into);
  }

  // ---- blocks --------------------------------------------------------------------------------
  int basic_block(const std::string& p, int x, int C, Ref into = Ref()) {           // hrnet.py:42-58
    int y = conv_bn(p + ".conv1", p + ".bn1", x, C, C, 3, 1, 1);
    return conv_bn(p + ".conv2", p + ".bn2", y, C, C, 3, 1, 1, x, false, 0, into);
  }
  // `cat` >= 0 (stride-1 block with a projection shortcut): x lives in channels [planes, planes + Cin) of the act `cat`.
  // conv2 then writes its output into channels [0, planes) of the same act and
  //     bn3(conv3(t)) + bn_d(conv_d(x))  =  [s3*W3 | sd*Wd] . [t ; x] + (b3 + bd)
  // runs as ONE 1x1 conv over the planes + Cin channels: the 4*planes-wide tensor is written once instead of written
  // by the shortcut conv, re-read as a residual and written again (at 56x56 that is 2 x 205 MB per 64 crops).
  int bottleneck(const std::string& p, int x, int Cin, int planes, int stride, bool down, int cat = -1) {   // :79-99
    if (cat >= 0) {
      const int Cout = planes * 4;
      const int t1 = conv(p + ".conv1", p + ".conv1", p + ".bn1", R(cat, planes), Cin, planes, 1, 1, 1, false);
      conv(p + ".conv2", p + ".conv2", p + ".bn2", R(t1), planes, planes, 3, 1, 1, false, Ref(), 0, R(cat, 0));
      const HostParam* w3 = P(p + ".conv3.weight", {Cout, planes, 1, 1});
      const HostParam* wd = P(p + ".downsample.0.weight", {Cout, Cin, 1, 1});
      std::vector<float> s3, b3, sd, bd;
      bn_fold(p + ".bn3", nullptr, Cout, s3, b3);
      bn_fold(p + ".downsample.1", nullptr, Cout, sd, bd);
      HostParam wm, bm;
      const bool have = !declare && w3 && wd;
      if (have) {
        const int K = planes + Cin;
        wm.shape = {Cout, K, 1, 1};
        wm.data.resize((size_t)Cout * K);
        bm.shape = {Cout};
        bm.data.resize(Cout);
        for (int o = 0; o < Cout; ++o) {
          for (int k = 0; k < planes; ++k) wm.data[(size_t)o * K + k] = (float)((double)s3[o] * w3->data[(size_t)o * planes + k]);
          for (int k = 0; k < Cin; ++k) wm.data[(size_t)o * K + planes + k] = (float)((double)sd[o] * wd->data[(size_t)o * Cin + k]);
          bm.data[o] = (float)((double)b3[o] + (double)bd[o]);
        }
      }
      return conv(p + ".conv3+downsample", "", "", R(cat, 0), planes + Cin, Cout, 1, 1, 1, true, Ref(), 0, Ref(), nullptr, 0,
                  false, true, have ? &wm : nullptr, have ? &bm : nullptr);
    }
    int y = conv_bn(p + ".conv1", p + ".bn1", x, Cin, planes, 1, 1, 1);
    y = conv_bn(p + ".conv2", p + ".bn2", y, planes, planes, 3, stride, 1);
    if (down && stride == 2 && dual && !in_parallel && (planes * 4) % 64 == 0) {
      // bn3(conv3(t)) + bn_d(conv_d(x, stride 2)) as one GEMM over [t ; x(2y,2x)] (launch_gemm1x1_dual): no separate
      // shortcut tensor, no residual read
      const int Cout = planes * 4;
      const HostParam* w3 = P(p + ".conv3.weight", {Cout, planes, 1, 1});
      const HostParam* wd = P(p + ".downsample.0.weight", {Cout, Cin, 1, 1});
      std::vector<float> s3, b3, sd, bd;
      bn_fold(p + ".bn3", nullptr, Cout, s3, b3);
      bn_fold(p + ".downsample.1", nullptr, Cout, sd, bd);
      const Act at = e.acts[y];
      Op op;
      op.type = OP_DUAL1X1; op.name = p + ".conv3+downsample";
      op.in = R(y); op.in2 = R(x); op.Cin = planes; op.C = Cin; op.Cout = Cout; op.stride = 2; op.actfn = 1;
      const int o = new_act(Cout, at.H, at.W);
      op.out = R(o);
      op.flops = 2.0 * at.H * at.W * (double)Cout * (planes + Cin);
      if (!declare && w3 && wd) {
        const int K = planes + Cin;
        std::vector<float> wm((size_t)Cout * K), bm(Cout), ones(Cout, 1.f);
        for (int c = 0; c < Cout; ++c) {
          for (int k = 0; k < planes; ++k) wm[(size_t)c * K + k] = (float)((double)s3[c] * w3->data[(size_t)c * planes + k]);
          for (int k = 0; k < Cin; ++k) wm[(size_t)c * K + planes + k] = (float)((double)sd[c] * wd->data[(size_t)c * Cin + k]);
          bm[c] = (float)((double)b3[c] + (double)bd[c]);
        }
        std::vector<float> packed(conv_packed_weight_floats(K, Cout, 1));
        conv_pack_weights(wm.data(), ones.data(), Cout, K, 1, Cout, packed.data());
        op.wdev = upload(packed); op.bdev = upload(bm);
      }
      push(std::move(op));
      return o;
    }
    int r = x;
    if (down) r = conv_bn(p + ".downsample.0", p + ".downsample.1", x, Cin, planes * 4, 1, stride, 0);
    return conv_bn(p + ".conv3", p + ".bn3", y, planes, planes * 4, 1, 1, 1, r);
  }

  // conv3 + residual + ReLU of block `pa` chained with conv1 + ReLU of block `pb` (planes = 64; csrc/bneck_chain.hip):
  // t [.,64] , x [.,256]  ->  y [.,256] (returned) and u [.,64] (*u_out)
  int chain_op(const std::string& pa, const std::string& pb, int t, int x, int* u_out) {
    const Act at = e.acts[t];
    const HostParam* w3 = P(pa + ".conv3.weight", {256, 64, 1, 1});
    const HostParam* w1 = P(pb + ".conv1.weight", {64, 256, 1, 1});
    std::vector<float> s3, b3, s1, b1;
    bn_fold(pa + ".bn3", nullptr, 256, s3, b3);
    bn_fold(pb + ".bn1", nullptr, 64, s1, b1);
    Op op;
    op.type = OP_CHAIN; op.name = pa + ".conv3+" + pb.substr(pb.rfind('.') + 1) + ".conv1";
    op.in = R(t); op.res = R(x);
    const int y = new_act(256, at.H, at.W), u = new_act(64, at.H, at.W);
    op.out = R(y); op.out2 = R(u);
    op.flops = 2.0 * at.H * at.W * (64.0 * 256 + 256.0 * 64);
    if (!declare && w3 && w1) {
      std::vector<float> p3(conv_packed_weight_floats(64, 256, 1)), p1(conv_packed_weight_floats(256, 64, 1));
      conv_pack_weights(w3->data.data(), s3.data(), 256, 64, 1, 256, p3.data());
      conv_pack_weights(w1->data.data(), s1.data(), 64, 256, 1, 64, p1.data());
      op.wdev = upload(p3); op.bdev = upload(b3);
      op.wdev2 = upload(p1); op.bdev2 = upload(b1);
    }
    push(std::move(op));
    *u_out = u;
    return y;
  }

  // layer1 of HRNet / ResNet-50: `nblk` Bottlenecks with planes = 64 at stride 1, the first with a projection shortcut
  int layer1(const std::string& lp, int x, int cat, int nblk) {
    x = bottleneck(lp + "0", x, 64, 64, 1, true, cat);
    if (!chain) {
      for (int k = 1; k < nblk; ++k) x = bottleneck(lp + std::to_string(k), x, 256, 64, 1, false);
      return x;
    }
    int u = conv_bn(lp + "1.conv1", lp + "1.bn1", x, 256, 64, 1, 1, 1);
    for (int k = 1; k < nblk; ++k) {
      const std::string q = lp + std::to_string(k);
      const int t = conv_bn(q + ".conv2", q + ".bn2", u, 64, 64, 3, 1, 1);
      if (k + 1 < nblk) x = chain_op(q, lp + std::to_string(k + 1), t, x, &u);
      else x = conv_bn(q + ".conv3", q + ".bn3", t, 64, 256, 1, 1, 1, x);
    }
    return x;
  }

  int stem(const std::string& p, int ks, int H) {                   // conv1+bn1+relu from the NCHW image
    const HostParam* w = P(p + "conv1.weight", {64, 3, ks, ks});
    std::vector<float> scale, shift;
    bn_fold(p + "bn1", nullptr, 64, scale, shift);
    const int pad = (ks - 1) / 2;
    const int Ho = (H + 2 * pad - ks) / 2 + 1;
    Op op; op.type = OP_STEM; op.name = p + "conv1"; op.ks = ks; op.n = H;
    op.in = X(X_IMG);
    op.flops = 2.0 * Ho * Ho * 64 * 3 * ks * ks;
    int out = new_act(64, Ho, Ho);
    op.out = R(out);
    if (!declare && w) {
      std::vector<float> wt((size_t)ks * ks * 3 * 64);
      for (int co = 0; co < 64; ++co)
        for (int c = 0; c < 3; ++c)
          for (int t = 0; t < ks * ks; ++t)
            wt[((size_t)t * 3 + c) * 64 + co] = w->data[((size_t)co * 3 + c) * ks * ks + t] * scale[co];
      op.wdev = upload(wt);
      op.bdev = upload(shift);
    }
    push(std::move(op));
    return out;
  }

  void fuse_sum(const std::string& name, const std::vector<std::pair<Ref, int>>& terms, int C, Ref out, int relu) {
    Op op; op.type = OP_FUSE; op.name = name; op.fn = (int)terms.size(); op.frelu = relu; op.C = C;
    for (size_t k = 0; k < terms.size(); ++k) { op.fsrc[k] = terms[k].first; op.fshift[k] = terms[k].second; }
    op.out = out;
    push(std::move(op));
  }

  // Several convs of the same geometry reading the SAME input, run as one launch: weights / folded BN are
  // concatenated along the output channels and the result is one wide tensor whose channel slices the
  // consumers read in place.  ReLU members must come last (act 3 = ReLU for channels >= relu_from).
  struct SubConv { std::string convp, bnp; int Cout; int relu; };
  int conv_multi(const std::string& name, const std::vector<SubConv>& subs, Ref in, int Cin, int ks, int stride,
                 std::vector<int>* offsets, Ref into = Ref()) {
    const Act ain = e.acts[in.act];
    const int pad = (ks - 1) / 2;
    const int Ho = (ain.H + 2 * pad - ks) / stride + 1, Wo = (ain.W + 2 * pad - ks) / stride + 1;
    int Ctot = 0, relu_from = -1;
    offsets->clear();
    for (const SubConv& sc : subs) {
      if (sc.Cout % 16) { ok = false; e.err += "conv_multi: member widths must be multiples of 16; "; }
      if (sc.relu && relu_from < 0) relu_from = Ctot;
      if (!sc.relu && relu_from >= 0) { ok = false; e.err += "conv_multi: ReLU members must come last; "; }
      offsets->push_back(Ctot + (into.act >= 0 ? into.co : 0));
      Ctot += sc.Cout;
    }
    const size_t per = (size_t)Cin * ks * ks;
    std::vector<float> wcat, scat, hcat;
    bool have = !declare;
    for (const SubConv& sc : subs) {
      const HostParam* w = P(sc.convp + ".weight", {sc.Cout, Cin, ks, ks});
      std::vector<float> scale, shift;
      bn_fold(sc.bnp, nullptr, sc.Cout, scale, shift);
      if (declare || !w) { have = false; continue; }
      wcat.insert(wcat.end(), w->data.begin(), w->data.begin() + (size_t)sc.Cout * per);
      scat.insert(scat.end(), scale.begin(), scale.end());
      hcat.insert(hcat.end(), shift.begin(), shift.end());
    }
    Op op;
    op.type = OP_CONV; op.name = name;
    op.in = in; op.Cin = Cin; op.Cout = Ctot; op.ks = ks; op.stride = stride;
    op.actfn = relu_from < 0 ? 0 : (relu_from == 0 ? 1 : 3);
    op.relu_from = std::max(relu_from, 0);
    op.flops = 2.0 * Ho * Wo * (double)Ctot * Cin * ks * ks;
    const int out_act = into.act >= 0 ? into.act : new_act(Ctot, Ho, Wo);
    op.out = into.act >= 0 ? into : R(out_act);
    if (have) {
      std::vector<float> packed(conv_packed_weight_floats(Cin, Ctot, ks));
      conv_pack_weights(wcat.data(), scat.data(), Ctot, Cin, ks, Ctot, packed.data());
      op.wdev = upload(packed);
      op.bdev = upload(hcat);
    }
    push(std::move(op));
    return out_act;
  }

  // HighResolutionModule.forward, hrnet.py:248-266.  last_into: write branch-0 output into a wider
  // concat buffer (POCO-PARE's 480-channel feature map) instead of a fresh tensor.
  // keep_open: the module is followed by another module of the same stage -> its fuse sums and the next module's
  // branch chains form ONE parallel region (lane i = sum_i, then the 8 convs of branch i): one join less per
  // module, and a lane's sum overlaps the other lanes' first convs.
  bool region_open = false;
  // K-merge (round 3; `kmerge`, POCO_NO_KMERGE=1 restores the separate form): the sum of the lowest-resolution branch T = nb-1,
  //     y_T = relu(x_T + sum_{j<T} bn_j(conv_j(t_j)))        (t_j = running tensor of down path j -> T before its last conv),
  // is ONE stride-2 3x3 conv over the channel concatenation [t_{T-2} | ... | t_0 | x_{T-1}] with the BN-folded weights
  // concatenated along K, the summed shifts as bias, x_T as the residual and the ReLU in the epilogue: T launches and the
  // branch's fuse_sum launch become one, the Cout x 7x7 (14x14) partial results are never written, and the GEMM is T times
  // deeper (K = 336 instead of 192 / 96 / 48 for W48 stage 4).  The concat costs nothing: the producers write into channel
  // slices of one buffer `kc` = [A | t_{T-2} | ... | t_0 | x_{T-1}], where [A | t_{T-2}] is the output of the merged first convs
  // that read x_{T-2} (A = last conv of path T-2 -> T-1) and x_{T-1} is written there by conv2 of the branch's last block.
  std::vector<int> hr_module(const std::string& p, std::vector<int> xs, const std::vector<int>& ch, Ref out0 = Ref(),
                             bool keep_open = false) {
    const int nb = (int)xs.size();
    const int T = nb - 1;
    const bool km = kmerge && nb >= 2;
    int kc = -1;
    std::vector<int> kc_off(nb, 0);
    if (km && nb >= 3) {
      int off = ch[T - 1];
      for (int j = T - 2; j >= 0; --j) { kc_off[j] = off; off += ch[j]; }
      kc_off[T - 1] = off; off += ch[T - 1];
      kc = new_act(off, e.acts[xs[T - 1]].H, e.acts[xs[T - 1]].W);
    }
    // phase 1: the branches are independent chains of 8 convs -> one lane (HIP stream) each
    if (!region_open) begin_parallel(1);
    region_open = false;
    const std::string& lmap = e.opts.branch_lanes;
    std::vector<Ref> xr(nb);                 // branch outputs as views (branch T-1 lives inside kc)
    for (int i = 0; i < nb; ++i) {
      lane(i < (int)lmap.size() ? lmap[i] - '0' : i);
      for (int k = 0; k < 4; ++k) {
        const bool to_kc = kc >= 0 && i == T - 1 && k == 3;
        xs[i] = basic_block(p + ".branches." + std::to_string(i) + "." + std::to_string(k), xs[i], ch[i],
                            to_kc ? R(kc, kc_off[i]) : Ref());
      }
      xr[i] = (kc >= 0 && i == T - 1) ? R(kc, kc_off[i]) : R(xs[i]);
    }
    // xdep (round 3): ONE join per module instead of three.  The fuse convs that read x_j follow branch j's chain on ITS lane and
    // the middle convs of a down chain stay on the lane of their source, so chains, first fuse convs and middle convs of a module
    // need no synchronisation at all (round 2 joined all lanes after the chains and again after the first fuse convs and spread the
    // fuse convs round-robin over the lanes): a lane that finishes its chain early does its fuse convs while the slowest chain is
    // still running, and behind the slowest chain only its own fuse convs remain before the join in front of the sums.
    // W48-CLIFF 64 crops 4508 -> 4654 crops/s (+3.2 %), PARE 32 crops +3.3 %, W48 16 crops +6.5 % (same box, POCO_NO_XDEP=1 against default).
    const bool dag = xdep && !region_seq;
    auto BL = [&](int i) { return i < (int)lmap.size() ? lmap[i] - '0' : i; };
    if (!dag) end_parallel();
    // phase 2: cross-resolution terms.  All first convs that read the same branch output xs[j] run as ONE
    // launch each (the 1x1 up-path convs j -> i<j, and the first 3x3 stride-2 convs of the down paths
    // j -> i>j): the input is read once and the small per-path launches disappear.
    std::vector<std::vector<std::pair<Ref, int>>> terms(nb);
    std::vector<std::vector<Ref>> chain(nb, std::vector<Ref>(nb));   // [i][j]: running tensor of down path j -> i
    for (int i = 0; i < nb; ++i) terms[i].resize(nb);
    auto fl = [&](int i, int j) { return p + ".fuse_layers." + std::to_string(i) + "." + std::to_string(j); };
    if (!dag) begin_parallel(2);
    int rr = 0;
    for (int j = 0; j < nb; ++j) {
      if (j > 0) {                       // up paths: 1x1 conv + BN (hrnet.py:196-207), upsampled inside the sum
        std::vector<SubConv> subs;
        for (int i = 0; i < j; ++i) subs.push_back({fl(i, j) + ".0", fl(i, j) + ".1", ch[i], 0});
        std::vector<int> off;
        lane(dag ? BL(j) : rr++);
        const int t = conv_multi(p + ".fuse_up." + std::to_string(j), subs, xr[j], ch[j], 1, 1, &off);
        for (int i = 0; i < j; ++i) terms[i][j] = {R(t, off[i]), j - i};
      }
      if (j + 1 < nb) {                  // down paths: first 3x3 stride-2 conv of every chain (hrnet.py:208-236)
        std::vector<SubConv> subs;
        std::vector<int> who;
        for (int i = j + 1; i < nb; ++i) {
          const bool last = (i == j + 1);
          if (km && last && i == T) continue;        // the only conv of path T-1 -> T: part of the K-merged conv
          subs.push_back({fl(i, j) + ".0.0", fl(i, j) + ".0.1", last ? ch[i] : ch[j], last ? 0 : 1});
          who.push_back(i);
        }
        if (!subs.empty()) {
          std::vector<int> off;
          lane(dag ? BL(j) : rr++);
          // the convs reading x_{T-2} produce exactly [A | t_{T-2}]: straight into the head of kc
          const Ref into = (kc >= 0 && j == T - 2) ? R(kc, 0) : Ref();
          const int t = conv_multi(p + ".fuse_down." + std::to_string(j), subs, xr[j], ch[j], 3, 2, &off, into);
          for (size_t m = 0; m < who.size(); ++m) chain[who[m]][j] = R(t, off[m]);
        }
      }
    }
    if (!dag) end_parallel();
    // phase 2b: the rest of the down chains (different inputs -> separate launches), one lane per chain
    if (!dag) begin_parallel(3);
    rr = 0;
    for (int i = 0; i < nb; ++i)
      for (int j = 0; j < i; ++j) {
        const bool merged = km && i == T;           // the last conv of this chain runs inside the K-merged conv
        const int nconv = i - j - 1 - (merged ? 1 : 0);   // convs of the chain that run here
        if (nconv > 0) lane(dag ? BL(j) : rr++);       // (on another, lighter lane with an event: -0.3 % - more waiting than balance)
        Ref t = chain[i][j];
        for (int k = 1; k < i - j - (merged ? 1 : 0); ++k) {
          const bool lastk = (k == i - j - 1);
          const bool to_kc = merged && kc >= 0 && k == i - j - 2;
          const std::string qq = fl(i, j) + "." + std::to_string(k);
          const int y = conv(qq + ".0", qq + ".0", qq + ".1", t, ch[j], lastk ? ch[i] : ch[j], 3, 2, lastk ? 0 : 1, false,
                             Ref(), 0, to_kc ? R(kc, kc_off[j]) : Ref());
          t = to_kc ? R(kc, kc_off[j]) : R(y);
        }
        terms[i][j] = {t, 0};
      }
    // phase 3: the sums (+ReLU), one lane per output branch - behind the ONE join of a module: every sum needs terms of (nearly)
    // all lanes anyway.  (Measured without this join as well - the sums waiting for exactly their producers' lanes through
    // events: correct in eager mode, but hipStreamEndCapture segfaults on the captured graph as soon as a stage with three
    // branches is scheduled that way, ROCm 7.2; the join costs nothing measurable.)
    end_parallel();
    std::vector<int> outs(nb);
    begin_parallel(4);
    for (int i = 0; i < nb; ++i) {
      lane(dag ? BL(i) : i);
      const Act a = e.acts[xs[i]];
      if (km && i == T) { outs[i] = kmerge_conv(p, fl, ch, T, kc >= 0 ? R(kc, ch[T - 1]) : xr[0], R(xs[T])); continue; }
      terms[i][i] = {xr[i], 0};
      if (i == 0 && out0.act >= 0) { outs[i] = out0.act; fuse_sum(p + ".fuse" + std::to_string(i), terms[i], ch[i], out0, 1); }
      else { outs[i] = new_act(ch[i], a.H, a.W); fuse_sum(p + ".fuse" + std::to_string(i), terms[i], ch[i], R(outs[i]), 1); }
    }
    if (keep_open && !((seq_mask >> 7) & 1)) region_open = true;
    else end_parallel();
    return outs;
  }

  // relu(x_T + sum_j bn_j(conv_j(t_j))) as one conv over the concatenated inputs (see hr_module); K order = kc order:
  // sources T-2, T-3, ..., 0, then T-1
  template <class FL>
  int kmerge_conv(const std::string& p, FL fl, const std::vector<int>& ch, int T, Ref in, Ref res) {
    std::vector<int> order;
    for (int j = T - 2; j >= 0; --j) order.push_back(j);
    order.push_back(T - 1);
    int K = 0;
    for (int j : order) K += ch[j];
    const int Cout = ch[T];
    HostParam wm, bm;
    wm.shape = {Cout, K, 3, 3};
    bm.shape = {Cout};
    bool have = !declare;
    std::vector<double> bsum(Cout, 0.0);
    if (have) wm.data.assign((size_t)Cout * K * 9, 0.f);
    int koff = 0;
    for (int j : order) {
      const std::string q = fl(T, j) + "." + std::to_string(T - j - 1);
      const HostParam* w = P(q + ".0.weight", {Cout, ch[j], 3, 3});
      std::vector<float> scale, shift;
      bn_fold(q + ".1", nullptr, Cout, scale, shift);
      if (declare || !w) have = false;
      if (have) {
        for (int o = 0; o < Cout; ++o) {
          for (int c = 0; c < ch[j]; ++c)
            for (int t = 0; t < 9; ++t)
              wm.data[((size_t)o * K + koff + c) * 9 + t] = (float)((double)scale[o] * w->data[((size_t)o * ch[j] + c) * 9 + t]);
          bsum[o] += (double)shift[o];
        }
      }
      koff += ch[j];
    }
    if (have) { bm.data.resize(Cout); for (int o = 0; o < Cout; ++o) bm.data[o] = (float)bsum[o]; }
    return conv(p + ".fuse" + std::to_string(T) + "+down", "", "", in, K, Cout, 3, 2, 1, true, res, 0, Ref(), nullptr, 0, false, true,
                have ? &wm : nullptr, have ? &bm : nullptr);
  }

  // stem + layer1 + transitions + stages 2-4 (hrnet.py:466-497 / hrnet_cls.py:438-469)
  std::vector<int> hrnet_trunk(const std::string& p, int w, Ref final_out0 = Ref(), bool open_after = false) {
    int x = stem(p, 3, 224);
    const int cat = kcat ? new_act(128, 56, 56) : -1;      // [layer1.0 conv2 output | stem output], see bottleneck()
    x = conv_bn(p + "conv2", p + "bn2", x, 64, 64, 3, 2, 1, -1, false, 0, kcat ? R(cat, 64) : Ref());
    x = layer1(p + "layer1.", x, cat, 4);
    std::vector<int> ys = {x};
    std::vector<int> prev_ch = {256};
    const int nmod[3] = {1, 4, 3};
    for (int s = 0; s < 3; ++s) {
      const int nb = s + 2;
      std::vector<int> ch(nb);
      for (int i = 0; i < nb; ++i) ch[i] = w << i;
      const std::string t = p + "transition" + std::to_string(s + 1);
      std::vector<int> xs(nb);
      // the transition convs are few and large (each fills the chip on its own): measured faster back to back
      // on one stream than as concurrent lanes (19.0-19.2 vs 19.35 ms per 64-crop forward)
      const bool open = region_open;          // the previous stage's last module left its region open (xdep)
      if (!open) {          // (transition 1 as lanes of the stage-2 region, i.e. without its join: W48 -0.6 %, PARE +0.2 % - not kept)
        begin_parallel(5);
        region_seq = !((seq_mask >> 6) & 1);
      }
      for (int i = 0; i < nb; ++i) {
        lane(i);
        const std::string ti = t + "." + std::to_string(i);
        if (i < (int)ys.size()) {
          if (prev_ch[i] != ch[i]) xs[i] = conv_bn(ti + ".0", ti + ".1", ys[i], prev_ch[i], ch[i], 3, 1, 1);
          else xs[i] = ys[i];
        } else {
          // hrnet.py:402-418: new branch = stride-2 conv chain from the last previous branch
          xs[i] = conv_bn(ti + ".0.0", ti + ".0.1", ys.back(), prev_ch.back(), ch[i], 3, 2, 1);
        }
      }
      if (!open) end_parallel();
      for (int m = 0; m < nmod[s]; ++m) {
        const bool last = (s == 2 && m == nmod[s] - 1);
        const bool stage_end_open = xdep && (m + 1 == nmod[s]) && (s < 2 || open_after);
        xs = hr_module(p + "stage" + std::to_string(s + 2) + "." + std::to_string(m), xs, ch, last ? final_out0 : Ref(),
                       m + 1 < nmod[s] || stage_end_open);
      }
      ys = xs;
      prev_ch = ch;
    }
    return ys;
  }
};

// ------------------------------------------------------------------------------------------------
// Variant assembly
// ------------------------------------------------------------------------------------------------
constexpr int XC_BBOX = 2048, XC_STATE = 2052, XC_DIM = 2224;   // CLIFF fc1 input layout
constexpr int XU_DIM = 3296;                                   // PARE uncert input: 3072 feat + 216 rot + pad

void build_smpl(Builder& b) {
  Engine& e = b.e;
  constexpr int V = 6890;
  static_assert(V <= SMPL_MAX_V, "smpl_joints_kernel's unrolled vertex rounds (kernels.h SMPL_JOINTS_ITERS) must cover the body model");
  const HostParam* vt = b.P("smpl.v_template", {V, 3});
  const HostParam* sd = b.P("smpl.shapedirs", {V, 3, 10});
  const HostParam* pd = b.P("smpl.posedirs", {207, V * 3});
  const HostParam* jr = b.P("smpl.J_regressor", {24, V});
  const HostParam* lw = b.P("smpl.lbs_weights", {V, 24});
  const HostParam* je = b.P("smpl.J_regressor_extra", {9, V});
  const HostParam* par = b.P("smpl.parents", {24});
  const HostParam* ev = b.P("smpl.extra_vertex_ids", {21});
  const HostParam* jm = b.P("smpl.joint_map", {49});
  e.a_A = b.new_act(288, 1, 1, true);
  e.a_coef = b.new_act(SMPL_KB + 4, 1, 1, true);
  e.a_j24 = b.new_act(72, 1, 1, true);
  e.a_verts = b.new_act(V * 3, 1, 1, true);
  e.a_j49 = b.new_act(147, 1, 1, true);
  e.a_camt = b.new_act(4, 1, 1, true);
  e.a_fullt = b.new_act(4, 1, 1, true);
  e.a_j2d = b.new_act(100, 1, 1, true);
  if (b.declare || !vt || !sd || !pd || !jr || !lw || !je || !par || !ev || !jm) return;
  // J = J_regressor . (v_template + shapedirs . betas)  ->  fold the regressor (float64 on the host)
  std::vector<float> Jt(72), Js(720), sdt((size_t)10 * V * 3);
  for (int j = 0; j < 24; ++j)
    for (int k = 0; k < 3; ++k) {
      double acc = 0;
      for (int v = 0; v < V; ++v) acc += (double)jr->data[(size_t)j * V + v] * vt->data[(size_t)v * 3 + k];
      Jt[j * 3 + k] = (float)acc;
      for (int l = 0; l < 10; ++l) {
        double a2 = 0;
        for (int v = 0; v < V; ++v) a2 += (double)jr->data[(size_t)j * V + v] * sd->data[((size_t)v * 3 + k) * 10 + l];
        Js[(j * 3 + k) * 10 + l] = (float)a2;
      }
    }
  for (int v = 0; v < V; ++v)
    for (int k = 0; k < 3; ++k)
      for (int l = 0; l < 10; ++l) sdt[(size_t)l * V * 3 + v * 3 + k] = sd->data[((size_t)v * 3 + k) * 10 + l];
  auto to_int = [](const HostParam* p) { std::vector<int> r(p->data.size()); for (size_t i = 0; i < r.size(); ++i) r[i] = (int)std::lround(p->data[i]); return r; };
  e.smpl.V = V;
  e.smpl.v_template = b.upload(vt->data);
  e.smpl.shapedirs = b.upload(sdt);
  e.smpl.posedirs = b.upload(pd->data);
  {   // the blend GEMM's B matrix (kernels_smpl.hip): posedirs | shapedirs | v_template stacked, coordinate-major, V padded to 32
    const int VP = (V + 31) / 32 * 32;
    std::vector<float> bl((size_t)SMPL_KB * 3 * VP, 0.f);
    for (int v = 0; v < V; ++v)
      for (int c = 0; c < 3; ++c) {
        for (int k = 0; k < 207; ++k) bl[((size_t)k * 3 + c) * VP + v] = pd->data[(size_t)k * V * 3 + v * 3 + c];
        for (int l = 0; l < 10; ++l) bl[((size_t)(207 + l) * 3 + c) * VP + v] = sd->data[((size_t)v * 3 + c) * 10 + l];
        bl[((size_t)217 * 3 + c) * VP + v] = vt->data[(size_t)v * 3 + c];
      }
    e.smpl.blend_cm = b.upload(bl);
    e.smpl.VP = VP;
  }
  e.smpl.lbs_weights = b.upload(lw->data);
  e.smpl.J_template = b.upload(Jt);
  e.smpl.J_shapedirs = b.upload(Js);
  e.smpl.J_regressor_extra = b.upload(je->data);
  e.smpl.parents = b.upload_i(to_int(par));
  e.smpl.extra_vertex_ids = b.upload_i(to_int(ev));
  e.smpl.joint_map = b.upload_i(to_int(jm));
}

void build_flow(Builder& b, int in_ctx) {
  Engine& e = b.e;
  const int L = 2 * e.flow_layers, D = 9, H = 64, ctx = 512, K0 = D + ctx;
  // cond_layer is evaluated and discarded by the reference at inference (nf_head.py:82,129-136):
  // declared (so checkpoints load strictly) but never launched in forward.
  b.P("flow_head.cond_layer.weight", {ctx, in_ctx}, 0);
  b.P("flow_head.cond_layer.bias", {ctx}, 0);
  const HostParam* mask = b.P("flow_head.flow.mask", {L, D}, 0);
  const int NM = L * 2;                                   // MLPs: index li*2 + net (net 0 = s, 1 = t)
  std::vector<float> wpack((size_t)NM * 24 * 256), b1((size_t)NM * H), b2((size_t)NM * 16, 0.f), wctx((size_t)NM * H * ctx),
      bctx((size_t)NM * H);
  bool all = mask != nullptr;
  for (int net = 0; net < 2; ++net) {
    const std::string nn = net == 0 ? "s" : "t";
    for (int l = 0; l < L; ++l) {
      const std::string q = "flow_head.flow." + nn + "." + std::to_string(l);
      const HostParam* W0 = b.P(q + ".0.weight", {H, K0}, 0);
      const HostParam* B0 = b.P(q + ".0.bias", {H}, 0);
      const HostParam* W1 = b.P(q + ".2.weight", {H, H}, 0);
      const HostParam* B1 = b.P(q + ".2.bias", {H}, 0);
      const HostParam* W2 = b.P(q + ".4.weight", {D, H}, 0);
      const HostParam* B2 = b.P(q + ".4.bias", {D}, 0);
      if (b.declare) continue;
      if (!W0 || !B0 || !W1 || !B1 || !W2 || !B2) { all = false; continue; }
      const int m = l * 2 + net;
      // first Linear: the 9 state columns (K padded to 16) feed the coupling kernel, the 512 context columns step A
      std::vector<float> w0z((size_t)H * 16, 0.f);
      for (int h = 0; h < H; ++h) {
        for (int i = 0; i < D; ++i) w0z[(size_t)h * 16 + i] = W0->data[(size_t)h * K0 + i];
        for (int k = 0; k < ctx; ++k) wctx[((size_t)m * H + h) * ctx + k] = W0->data[(size_t)h * K0 + D + k];
        bctx[(size_t)m * H + h] = B0->data[h];
        b1[(size_t)m * H + h] = B1->data[h];
      }
      for (int i = 0; i < D; ++i) b2[(size_t)m * 16 + i] = B2->data[i];
      float* wp = wpack.data() + (size_t)m * 24 * 256;
      conv_pack_weights(w0z.data(), nullptr, H, 16, 1, H, wp);                       // 4 quads
      conv_pack_weights(W1->data.data(), nullptr, H, H, 1, H, wp + 4 * 256);          // 16 quads [k slice][n-tile]
      conv_pack_weights(W2->data.data(), nullptr, D, H, 1, 16, wp + 20 * 256);        // 4 quads, rows 9..15 zero
    }
  }
  if (b.declare || !all) return;
  std::vector<float> mask16((size_t)L * 16, 0.f);
  for (int l = 0; l < L; ++l)
    for (int i = 0; i < D; ++i) mask16[(size_t)l * 16 + i] = mask->data[(size_t)l * D + i];
  std::vector<float> wcf(conv_packed_weight_floats(ctx, NM * H, 1));
  conv_pack_weights(wctx.data(), nullptr, NM * H, ctx, 1, NM * H, wcf.data());
  e.flow.L = L; e.flow.ctx = ctx;
  e.flow.mask16 = b.upload(mask16);
  e.flow.wpack = reinterpret_cast<const float4*>(b.upload(wpack));
  e.flow.b1 = b.upload(b1); e.flow.b2 = b.upload(b2);
  e.flow.wctx_frag = b.upload(wcf); e.flow.bctx = b.upload(bctx);
  e.has_flow = true;
}

void add_copy(Builder& b, const std::string& name, Ref src, Ref dst, int n) {
  Op op; op.type = OP_COPY; op.name = name; op.in = src; op.out = dst; op.n = n;
  b.push(std::move(op));
}

// poco_outputs_t.record: the packed per-crop record [rotmat 216 | betas 10 | cam 3 | var_pose 24 | confidence 1] (the payload of the
// multi-GPU all-gather) written by ONE kernel at the end of the confidence lane; the confidence is the reference's
// post-processed global uncertainty (poco_utils.py:21-25,50-60 + the clip of tester.py:245).  Skipped when the caller passes no pointer.
void add_record(Builder& b, Ref var, bool cliff) {
  Op op; op.type = OP_RECORD; op.name = "out.record"; op.in = var; op.out = Builder::X(Y_RECORD); op.n = cliff;
  b.push(std::move(op));
}

void build_tail(Builder& b, Ref betas, Ref rot, Ref cam, bool cliff) {
  Engine& e = b.e;
  build_smpl(b);
  e.smpl_betas = betas; e.smpl_rot = rot; e.cam_ref = cam;
  { Op op; op.type = OP_SMPL; op.name = "smpl.lbs"; op.flops = 2.0 * 7.9e6; b.push(std::move(op)); }
  { Op op; op.type = OP_CAMERA; op.name = "smpl.camera"; op.n = cliff; b.push(std::move(op)); }
}

bool build_graph(Engine& e, bool declare) {
  Builder b(e, declare);
  e.acts.clear();
  e.ops.clear();
  const bool cliff = e.head == "cliff";
  const std::string bp = "backbone.";
  // (ResNet-50 is one chain of kernels and never forks a stream: a fork / join just for the tail costs more than the overlap gives,
  // 808 -> 771 crops/s at one crop, -0.2 % at 64)
  if (e.backbone == "resnet50") b.tail_lanes = false;
  int xc = -1;          // CLIFF fc1 input vector [XC_DIM]
  int feat480 = -1;
  if (cliff) xc = b.new_act(XC_DIM, 1, 1, true);

  // ---- backbone ------------------------------------------------------------------------------
  if (e.backbone == "hrnet_w32") {
    feat480 = b.new_act(480, 56, 56);
    const bool up_lanes = e.opts.up_lanes;
    std::vector<int> ys = b.hrnet_trunk(bp, 32, Builder::R(feat480, 0), /*open_after=*/up_lanes);
    // hrnet.py:515-519: bilinear x2 (align_corners) + conv3x3 + BN + ReLU chains, channel concat.  The three chains are
    // independent: each one continues on the lane that ran its branch's fuse sum of the last module (POCO_NO_UP_LANES=1: one
    // after the other on one stream, the round-2 form)
    const int chs[4] = {32, 64, 128, 256};
    const int offs[4] = {0, 32, 96, 224};
    const bool up_open = b.region_open;
    b.region_open = false;
    for (int br = 1; br < 4; ++br) {
      if (up_open) b.lane(br);
      int y = ys[br];
      for (int t = 0; t < br; ++t) {
        const Act a = e.acts[y];
        Op up; up.type = OP_BILINEAR; up.name = bp + "upsample_stage_" + std::to_string(br + 1) + "." + std::to_string(4 * t);
        up.in = Builder::R(y);
        int u = b.new_act(a.C, a.H * 2, a.W * 2);
        up.out = Builder::R(u);
        b.push(std::move(up));
        const std::string q = bp + "upsample_stage_" + std::to_string(br + 1) + ".";
        const bool lastt = (t == br - 1);
        y = b.conv(q + std::to_string(1 + 4 * t), q + std::to_string(1 + 4 * t), q + std::to_string(2 + 4 * t),
                   Builder::R(u), chs[br], chs[br], 3, 1, 1, false, Ref(), 0,
                   lastt ? Builder::R(feat480, offs[br]) : Ref());
      }
    }
    if (up_open) b.end_parallel();
    // optional copy of the 480-channel map for parity checks (skipped when the caller passes no pointer)
    { Op op; op.type = OP_NCHW_OUT; op.name = "out.backbone_feat"; op.in = Builder::R(feat480); op.out = Builder::X(Y_BBFEAT);
      op.C = 480; b.push(std::move(op)); }
    // present in reference checkpoints, unused by forward (hrnet.py:326-332)
    b.P(bp + "final_layer.weight", {24, 32, 1, 1}, 0);
    b.P(bp + "final_layer.bias", {24}, 0);
  } else if (e.backbone == "hrnet_w48_cls") {
    std::vector<int> ys = b.hrnet_trunk(bp, 48, Ref(), /*open_after=*/true);
    const int hc[4] = {32, 64, 128, 256};
    // the four incre_modules (one Bottleneck per resolution, hrnet_cls.py:306-321) are independent and small
    // (20-50 us kernels): one lane each - the lane that ran the branch's fuse sum of the last module when the trunk left its
    // region open - then the sequential downsample chain
    int inc[4];
    if (!b.region_open) b.begin_parallel(8);
    b.region_open = false;
    for (int i = 0; i < 4; ++i) {
      b.lane(i);
      inc[i] = b.bottleneck(bp + "incre_modules." + std::to_string(i) + ".0", ys[i], 48 << i, hc[i], 1, true);
    }
    b.end_parallel();
    int y = inc[0];
    for (int i = 0; i < 3; ++i) {
      const std::string q = bp + "downsamp_modules." + std::to_string(i);
      // hrnet_cls.py:475-477:  incre(y_{i+1}) + ReLU(BN(conv_s2(y)))   -> residual added after the ReLU
      y = b.conv_bn(q + ".0", q + ".1", y, hc[i] * 4, hc[i + 1] * 4, 3, 2, 1, inc[i + 1], true, 1);
    }
    y = b.conv_bn(bp + "final_layer.0", bp + "final_layer.1", y, 1024, 2048, 1, 1, 1, -1, true);
    Op ap; ap.type = OP_AVGPOOL; ap.name = bp + "avgpool"; ap.in = Builder::R(y); ap.out = Builder::R(xc, 0);
    b.push(std::move(ap));
    b.P(bp + "classifier.weight", {1000, 2048}, 0);
    b.P(bp + "classifier.bias", {1000}, 0);
  } else if (e.backbone == "resnet50") {
    int x = b.stem(bp, 7, 224);
    const Act a = e.acts[x];
    Op mp; mp.type = OP_MAXPOOL; mp.name = bp + "maxpool"; mp.in = Builder::R(x);
    const int Hp = (a.H + 2 - 3) / 2 + 1, Wp = (a.W + 2 - 3) / 2 + 1;
    const int cat = b.kcat ? b.new_act(128, Hp, Wp) : -1;   // [layer1.0 conv2 output | max-pool output]
    int p = b.kcat ? cat : b.new_act(64, Hp, Wp);
    mp.out = b.kcat ? Builder::R(cat, 64) : Builder::R(p);
    b.push(std::move(mp));
    x = p;
    const int nblk[4] = {3, 4, 6, 3};
    int cin = 64;
    x = b.layer1(bp + "layer1.", x, cat, nblk[0]);
    cin = 256;
    for (int li = 1; li < 4; ++li) {
      const int planes = 64 << li;
      for (int k = 0; k < nblk[li]; ++k) {
        const int stride = (k == 0 && li > 0) ? 2 : 1;
        x = b.bottleneck(bp + "layer" + std::to_string(li + 1) + "." + std::to_string(k), x, cin, planes, stride, k == 0);
        cin = planes * 4;
      }
    }
    if (!cliff) { e.err = "resnet50 is only wired to the cliff head"; return false; }
    Op ap; ap.type = OP_AVGPOOL; ap.name = "head.avgpool"; ap.in = Builder::R(x); ap.out = Builder::R(xc, 0);
    b.push(std::move(ap));
  } else {
    e.err = "unknown backbone " + e.backbone;
    return false;
  }

  // ---- head + uncertainty MLP ------------------------------------------------------------------
  const std::string hp = "head.", up = "uncert_head.";
  if (cliff) {
    if (xc < 0) { e.err = "cliff head needs a pooled 2048-d feature"; return false; }
    // state = [pose6d(144) | shape(10) | cam(3)] lives inside the fc1 input vector (cliff_head.py:103-113)
    const HostParam* ip = b.P(hp + "init_pose", {1, 144});
    const HostParam* is = b.P(hp + "init_shape", {1, 10});
    const HostParam* ic = b.P(hp + "init_cam", {1, 3});
    // mlp_fuse: everything from here to rot6d is collected as the sub-ops of ONE persistent launch (OP_MLP, mlp_chain.hip);
    // `cur_stage` numbers the stages (a grid barrier in between): 0 = state / bbox rows, fc1's feature part, the confidence
    // 1 + 3 it ... 3 + 3 it = fc1.state / fc2 / decoders of iteration it, 10 = rot6d and the copies that scatter the final state
    // into the outputs.  The confidence branch's featNet and the uncert_feat copy read only the pooled feature: they ride with the
    // decoder stages 3 and 6, which have 10 x B/16 tile jobs for the grid
    const bool fuse = e.opts.mlp_fuse;
    std::vector<Op> mlp_sub;
    if (fuse) { b.capture = &mlp_sub; b.cur_stage = 0; }
    Op bc; bc.type = OP_BCAST; bc.name = hp + "init_state"; bc.out = Builder::R(xc, XC_STATE); bc.n = 157;
    if (!declare && ip && is && ic) {
      std::vector<float> st(ip->data);
      st.insert(st.end(), is->data.begin(), is->data.end());
      st.insert(st.end(), ic->data.begin(), ic->data.end());
      bc.wdev = b.upload(st);
    }
    b.push(std::move(bc));
    add_copy(b, hp + "bbox_info", Builder::X(X_BBOX), Builder::R(xc, XC_BBOX), 3);
    std::vector<int> perm(2208);
    for (int k = 0; k < 2048; ++k) perm[k] = k;
    for (int k = 0; k < 3; ++k) perm[2048 + k] = XC_BBOX + k;
    for (int k = 0; k < 157; ++k) perm[2051 + k] = XC_STATE + k;
    // decpose/decshape/deccam fused into one [157,1024] matrix: declared separately, stacked here
    const HostParam* dp = b.P(hp + "decpose.weight", {144, 1024});
    const HostParam* dpb = b.P(hp + "decpose.bias", {144});
    const HostParam* ds = b.P(hp + "decshape.weight", {10, 1024});
    const HostParam* dsb = b.P(hp + "decshape.bias", {10});
    const HostParam* dc = b.P(hp + "deccam.weight", {3, 1024});
    const HostParam* dcb = b.P(hp + "deccam.bias", {3});
    HostParam decW, decB;
    const bool have_dec = !declare && dp && ds && dc && dpb && dsb && dcb;
    if (have_dec) {
      decW.shape = {157, 1024}; decB.shape = {157};
      decW.data = dp->data; decW.data.insert(decW.data.end(), ds->data.begin(), ds->data.end());
      decW.data.insert(decW.data.end(), dc->data.begin(), dc->data.end());
      decB.data = dpb->data; decB.data.insert(decB.data.end(), dsb->data.begin(), dsb->data.end());
      decB.data.insert(decB.data.end(), dcb->data.begin(), dcb->data.end());
    }
    // fc1 is linear and [feat, bbox] never change over the 3 iterations (cliff_head.py:103-113): its
    // 2048-column feature part (+bias) is evaluated once, each iteration only adds the 176-column
    // [bbox | pose | shape | cam] part on top (residual epilogue).
    std::vector<int> perm_feat(2208, -1), perm_state(2208, -1);
    for (int k = 0; k < 2048; ++k) perm_feat[k] = k;
    for (int k = 2048; k < 2208; ++k) perm_state[k] = perm[k] - 2048;
    int h0 = b.conv(hp + "fc1.feat", hp + "fc1", "", Builder::R(xc, 0), 2208, 1024, 1, 1, 0, true, Ref(), 0, Ref(),
                    &perm_feat, 2048, true);
    (fuse ? mlp_sub : e.ops).back().flops = 2.0 * 1024 * 2048;
    const int u = b.new_act(448, 1, 1, true);       // [sigmoid(featNet(feat)) 216 | pad | sigmoid(poseNet(R)) 216 | pad] (poco_head.py:122-141)
    if (fuse) {
      b.cur_stage = 3;
      b.conv(up + "uncert_fc_featNet", up + "uncert_fc_featNet", "", Builder::R(xc, 0), 2048, 216, 1, 1, 2, true, Ref(), 0,
             Builder::R(u, 0), nullptr, 0, true);
      b.cur_stage = 6;
      add_copy(b, "out.uncert_feat", Builder::R(xc, 0), Builder::X(Y_UFEAT), 2048);
    }
    int h2 = -1;
    for (int it = 0; it < 3; ++it) {
      const std::string sfx = "#" + std::to_string(it);
      b.cur_stage = 1 + 3 * it;
      int h1 = b.conv(hp + "fc1.state" + sfx, hp + "fc1", "", Builder::R(xc, 2048), 2208, 1024, 1, 1, 0, false, Builder::R(h0), 0,
                      Ref(), &perm_state, XC_DIM - 2048, true);
      (fuse ? mlp_sub : e.ops).back().flops = 2.0 * 1024 * 160;
      b.cur_stage = 2 + 3 * it;
      h2 = b.conv(hp + "fc2" + sfx, hp + "fc2", "", Builder::R(h1), 1024, 1024, 1, 1, 0, true, Ref(), 0, Ref(),
                  nullptr, 0, true);
      // fused decoder: state += dec(h2)  (in place on the state slice of xc)
      b.cur_stage = 3 + 3 * it;
      b.conv(hp + "dec" + sfx, "", "", Builder::R(h2), 1024, 157, 1, 1, 0, true, Builder::R(xc, XC_STATE), 0,
             Builder::R(xc, XC_STATE), nullptr, 0, true, true, have_dec ? &decW : nullptr, have_dec ? &decB : nullptr);
    }
    int rot = b.new_act(224, 1, 1, true);
    b.cur_stage = 10;
    { Op op; op.type = OP_ROT6D; op.name = hp + "rot6d"; op.in = Builder::R(xc, XC_STATE); op.out = Builder::R(rot);
      op.out2 = Builder::X(Y_POSE); b.push(std::move(op)); }
    // the tail is two independent chains of small kernels: SMPL-LBS + camera on one lane, the output copies and the confidence
    // MLP on another (POCO_NO_TAIL_LANES=1: one stream)
    if (!fuse && b.tail_lanes) { b.begin_parallel(9); b.lane(1); }
    add_copy(b, "out.pred_pose6d", Builder::R(xc, XC_STATE), Builder::X(Y_POSE6D), 144);
    add_copy(b, "out.pred_shape", Builder::R(xc, XC_STATE + 144), Builder::X(Y_SHAPE), 10);
    add_copy(b, "out.pred_cam", Builder::R(xc, XC_STATE + 154), Builder::X(Y_CAM), 3);
    if (!fuse) add_copy(b, "out.uncert_feat", Builder::R(xc, 0), Builder::X(Y_UFEAT), 2048);
    add_copy(b, "out.body_feat2", Builder::R(h2), Builder::X(Y_BODY2), 1024);
    e.uncert_feat_dim = 2048;
    if (fuse) {
      b.capture = nullptr;
      Op op; op.type = OP_MLP; op.name = hp + "regressor";
      for (const Op& su : mlp_sub) op.flops += su.flops;
      std::stable_sort(mlp_sub.begin(), mlp_sub.end(), [](const Op& x, const Op& y) { return x.stage < y.stage; });
      op.sub = std::move(mlp_sub);
      b.push(std::move(op));
      if (b.tail_lanes) b.begin_parallel(9);
    }
    if (b.tail_lanes) b.lane(0);
    build_tail(b, Builder::R(xc, XC_STATE + 144), Builder::R(rot), Builder::R(xc, XC_STATE + 154), true);
    if (b.tail_lanes) b.lane(1);
    // poco_head 'feat-pose-net' (poco_head.py:122-141): sigmoid(featNet(feat)) || sigmoid(poseNet(R)) -> fc1 -> sigmoid
    if (!fuse)
      b.conv(up + "uncert_fc_featNet", up + "uncert_fc_featNet", "", Builder::R(xc, 0), 2048, 216, 1, 1, 2, true, Ref(), 0,
             Builder::R(u, 0), nullptr, 0, true);
    b.conv(up + "uncert_fc_poseNet", up + "uncert_fc_poseNet", "", Builder::R(rot), 216, 216, 1, 1, 2, true, Ref(), 0,
           Builder::R(u, 216), nullptr, 224, true);
    int var = b.conv(up + "uncert_fc1", up + "uncert_fc1", "", Builder::R(u), 432, 24, 1, 1, 2, true, Ref(), 0, Ref(),
                     nullptr, 448, true);
    add_copy(b, "out.var_pose", Builder::R(var), Builder::X(Y_VAR), 24);
    add_record(b, Builder::R(var), true);
    if (b.tail_lanes) b.end_parallel();
    build_flow(b, 2048);
  } else if (e.head == "pare") {
    if (feat480 < 0) { e.err = "pare head needs the hrnet_w32 backbone"; return false; }
    auto branch = [&](const std::string& nm) {
      int y = b.conv_bn(hp + nm + ".0", hp + nm + ".1", feat480, 480, 128, 3, 1, 1);
      return b.conv_bn(hp + nm + ".3", hp + nm + ".4", y, 128, 128, 3, 1, 1);
    };
    if (b.tail_lanes) { b.begin_parallel(9); b.lane(0); }
    int kp = branch("keypoint_deconv_layers");
    int heat = b.conv(hp + "keypoint_final_layer", hp + "keypoint_final_layer", "", Builder::R(kp), 128, 25, 1, 1, 0, true);
    if (b.tail_lanes) b.lane(1);
    int sm = branch("smpl_deconv_layers");
    int cs = b.conv(hp + "smpl_final_layer", hp + "smpl_final_layer", "", Builder::R(sm), 128, 64, 1, 1, 0, true);
    if (b.tail_lanes) b.end_parallel();
    int xu = b.new_act(XU_DIM, 1, 1, true);
    int flat = b.new_act(1536, 1, 1, true);
    e.a_attn_scratch = b.new_act((int)(part_attention_scratch_floats(1, 128)), 1, 1, true);
    { Op op; op.type = OP_ATTN; op.name = hp + "attn_pool_pose"; op.in = Builder::R(heat); op.in2 = Builder::R(sm);
      op.C = 128; op.out = Builder::R(xu, 0); op.flops = 2.0 * 24 * 3136 * 128; b.push(std::move(op)); }
    { Op op; op.type = OP_ATTN; op.name = hp + "attn_pool_camshape"; op.in = Builder::R(heat); op.in2 = Builder::R(cs);
      op.C = 64; op.out = Builder::R(flat, 0); op.flops = 2.0 * 24 * 3136 * 64; b.push(std::move(op)); }
    int pose6d = b.new_act(144, 1, 1, true);
    { const HostParam* w = b.P(hp + "pose_mlp.weight", {1, 6, 128, 24, 1, 1});
      Op op; op.type = OP_LC2D; op.name = hp + "pose_mlp"; op.in = Builder::R(xu, 0); op.out = Builder::R(pose6d);
      op.flops = 2.0 * 6 * 128 * 24;
      if (!declare && w) op.wdev = b.upload(w->data);
      b.push(std::move(op)); }
    // shape_mlp (10) and cam_mlp (3) stacked into one [13,1536] matrix (pare_head.py:902-906)
    const HostParam* sw = b.P(hp + "shape_mlp.weight", {10, 1536});
    const HostParam* sb = b.P(hp + "shape_mlp.bias", {10});
    const HostParam* cw = b.P(hp + "cam_mlp.weight", {3, 1536});
    const HostParam* cb = b.P(hp + "cam_mlp.bias", {3});
    HostParam scW, scB;
    const bool have_sc = !declare && sw && sb && cw && cb;
    if (have_sc) {
      scW.shape = {13, 1536}; scB.shape = {13};
      scW.data = sw->data; scW.data.insert(scW.data.end(), cw->data.begin(), cw->data.end());
      scB.data = sb->data; scB.data.insert(scB.data.end(), cb->data.begin(), cb->data.end());
    }
    int sc13 = b.conv(hp + "shape_cam_mlp", "", "", Builder::R(flat), 1536, 13, 1, 1, 0, true, Ref(), 0, Ref(), nullptr, 0,
                      true, true, have_sc ? &scW : nullptr, have_sc ? &scB : nullptr);
    e.acts[sc13].persistent = true;
    { Op op; op.type = OP_ROT6D; op.name = hp + "rot6d"; op.in = Builder::R(pose6d); op.out = Builder::R(xu, 3072);
      op.out2 = Builder::X(Y_POSE); b.push(std::move(op)); }
    for (const char* nm : {"temperature"}) b.P(hp + nm, {}, 0);
    b.P(hp + "init_pose", {1, 144}, 0);
    b.P(hp + "init_shape", {1, 10}, 0);
    b.P(hp + "init_cam", {1, 3}, 0);
    if (b.tail_lanes) { b.begin_parallel(9); b.lane(1); }
    add_copy(b, "out.pred_pose6d", Builder::R(pose6d), Builder::X(Y_POSE6D), 144);
    add_copy(b, "out.pred_shape", Builder::R(sc13, 0), Builder::X(Y_SHAPE), 10);
    add_copy(b, "out.pred_cam", Builder::R(sc13, 10), Builder::X(Y_CAM), 3);
    add_copy(b, "out.uncert_feat", Builder::R(xu, 0), Builder::X(Y_UFEAT), 3072);
    { Op op; op.type = OP_NCHW_OUT; op.name = "out.pred_segm_mask"; op.in = Builder::R(heat); op.out = Builder::X(Y_SEGM);
      op.C = 25; b.push(std::move(op)); }
    e.uncert_feat_dim = 3072;
    // rotmat for SMPL is read from the xu vector (channel 3072.., stride XU_DIM)
    if (b.tail_lanes) b.lane(0);
    build_tail(b, Builder::R(sc13, 0), Builder::R(xu, 3072), Builder::R(sc13, 10), false);
    if (b.tail_lanes) b.lane(1);
    // poco_head 'feat-pose' (poco_head.py:134-141): sigmoid(fc2(sigmoid(fc1([feat; R]))))
    int h = b.conv(up + "uncert_fc1", up + "uncert_fc1", "", Builder::R(xu), 3288, 512, 1, 1, 2, true, Ref(), 0, Ref(),
                   nullptr, XU_DIM, true);
    int var = b.conv(up + "uncert_fc2", up + "uncert_fc2", "", Builder::R(h), 512, 24, 1, 1, 2, true, Ref(), 0, Ref(),
                     nullptr, 0, true);
    add_copy(b, "out.var_pose", Builder::R(var), Builder::X(Y_VAR), 24);
    add_record(b, Builder::R(var), false);
    if (b.tail_lanes) b.end_parallel();
    build_flow(b, 3072);
  } else {
    e.err = "unknown head " + e.head;
    return false;
  }
  if (!b.ok) return false;
  return true;
}

// ------------------------------------------------------------------------------------------------
// Workspace planning: big NHWC activations share memory by liveness; vectors are persistent.
// ------------------------------------------------------------------------------------------------
void touch(Engine& e, const Ref& r, int i) {
  if (r.act < 0) return;
  Act& a = e.acts[r.act];
  a.first = std::min(a.first, i);
  a.last = std::max(a.last, i);
}

void plan_workspace(Engine& e) {
  // liveness in PHASE units: lanes of one phase run concurrently, so memory is only recycled
  // across the joins between phases
  int nphase = 0;
  for (int oi = 0; oi < (int)e.ops.size(); ++oi) {
    Op& op = e.ops[oi];
    const int i = op.phase;
    nphase = std::max(nphase, i + 1);
    touch(e, op.in, i); touch(e, op.in2, i); touch(e, op.res, i); touch(e, op.out, i); touch(e, op.out2, i);
    for (int k = 0; k < op.fn; ++k) touch(e, op.fsrc[k], i);
    for (const Op& su : op.sub) { touch(e, su.in, i); touch(e, su.res, i); touch(e, su.out, i); touch(e, su.out2, i); }
    if (op.type == OP_SMPL || op.type == OP_CAMERA || op.type == OP_RECORD) {
      touch(e, e.smpl_betas, i); touch(e, e.smpl_rot, i); touch(e, e.cam_ref, i);
    }
  }
  struct Blk { size_t off, size; };
  std::vector<Blk> freelist;
  size_t top = 0;
  const size_t B = (size_t)e.max_batch;
  auto align = [](size_t n) { return (n + 63) / 64 * 64; };
  // persistent first
  for (Act& a : e.acts)
    if (a.persistent) { a.off = top; top += align(a.per_crop() * B); }
  // events ordered by op index
  std::vector<std::vector<int>> born(nphase + 1), dies(nphase + 1);
  for (int k = 0; k < (int)e.acts.size(); ++k) {
    const Act& a = e.acts[k];
    if (a.persistent || a.last < 0) continue;
    born[a.first].push_back(k);
    dies[a.last].push_back(k);
  }
  for (int i = 0; i < nphase; ++i) {
    for (int k : born[i]) {
      Act& a = e.acts[k];
      const size_t need = align(a.per_crop() * B);
      int best = -1;
      for (int f = 0; f < (int)freelist.size(); ++f)
        if (freelist[f].size >= need && (best < 0 || freelist[f].size < freelist[best].size)) best = f;
      if (best >= 0) {
        a.off = freelist[best].off;
        if (freelist[best].size > need) { freelist[best].off += need; freelist[best].size -= need; }
        else freelist.erase(freelist.begin() + best);
      } else { a.off = top; top += need; }
    }
    for (int k : dies[i]) {
      const Act& a = e.acts[k];
      Blk nb{a.off, align(a.per_crop() * B)};
      // coalesce with neighbours
      for (int f = 0; f < (int)freelist.size();) {
        if (freelist[f].off + freelist[f].size == nb.off) { nb.off = freelist[f].off; nb.size += freelist[f].size; freelist.erase(freelist.begin() + f); }
        else if (nb.off + nb.size == freelist[f].off) { nb.size += freelist[f].size; freelist.erase(freelist.begin() + f); }
        else ++f;
      }
      freelist.push_back(nb);
    }
  }
  e.ws_floats = top;
}

// ------------------------------------------------------------------------------------------------
// Execution
// ------------------------------------------------------------------------------------------------
struct IO {
  const poco_inputs_t* in;
  const poco_outputs_t* out;
};

const float* ext_in(const IO& io, int slot) {
  switch (slot) {
    case X_IMG: return io.in->img;
    case X_BBOX: return io.in->bbox_info;
    case X_FOCAL: return io.in->focal_length;
    case X_SCALE: return io.in->scale;
    case X_CENTER: return io.in->center;
    case X_ORIG: return io.in->orig_shape;
    default: return nullptr;
  }
}
float* ext_out(const IO& io, int slot) {
  switch (slot) {
    case Y_POSE: return io.out->pred_pose;
    case Y_POSE6D: return io.out->pred_pose6d;
    case Y_SHAPE: return io.out->pred_shape;
    case Y_CAM: return io.out->pred_cam;
    case Y_CAM_T: return io.out->pred_cam_t;
    case Y_FULL_CAM_T: return io.out->pred_fullimg_cam_t;
    case Y_VERTS: return io.out->smpl_vertices;
    case Y_J3D: return io.out->smpl_joints3d;
    case Y_J2D: return io.out->smpl_joints2d;
    case Y_VAR: return io.out->var_pose;
    case Y_UFEAT: return io.out->uncert_feat;
    case Y_SEGM: return io.out->pred_segm_mask;
    case Y_BODY2: return io.out->body_feat2;
    case Y_BBFEAT: return io.out->backbone_feat;
    case Y_RECORD: return io.out->record;
    default: return nullptr;
  }
}

// activations are L16 (common.h): a channel offset inside a wider buffer is a slice-row offset
inline float* aptr(Engine& e, const Ref& r) {
  const Act& a = e.acts[r.act];
  return e.ws + a.off + l16_chan_off(r.co, a.W);
}
inline float* sptr(Engine& e, int act) { const Act& a = e.acts[act]; return e.ws + a.off; }
inline int astride(Engine& e, const Ref& r) { return e.acts[r.act].C; }   // valid for vector acts (H=W=1)

int run_op(Engine& e, Op& op, int B, const IO& io, hipStream_t s) {
  switch (op.type) {
    case OP_STEM: {
      const float* img = ext_in(io, X_IMG);
      if (!img) { poco_set_error("forward: img is NULL"); return POCO_ERR_ARG; }
      launch_stem_conv(img, op.wdev, op.bdev, aptr(e, op.out), B, op.n, op.n, op.ks, s, e.opts.stem_mfma);
      return POCO_OK;
    }
    case OP_CONV: {
      const Act& ai = e.acts[op.in.act];
      const Act& ao = e.acts[op.out.act];
      ConvDesc d{};
      d.in = aptr(e, op.in); d.in_cs = ai.C; d.in_co = 0;
      if (op.res.act >= 0) { d.res = aptr(e, op.res); d.res_cs = e.acts[op.res.act].C; }
      d.out = aptr(e, op.out); d.out_cs = ao.C; d.out_co = 0;
      d.wfrag = op.wdev; d.bias = op.bdev; d.wfrag_wino = op.wdev_wino; d.wfrag_wino4 = op.wdev_wino4; d.wfrag_wino4p = op.wdev_wino4p; d.wfrag_wino4w = op.wdev_wino4w;
      d.wfrag_wino4g = op.wdev_wino4g; d.scratch = e.wino4g_scratch[op.lane & 3]; d.scratch_floats = e.wino4g_scratch_need;
      d.sk_scratch = e.sk_scratch[op.lane & 3]; d.sk_scratch_floats = d.sk_scratch ? gemm1x1sk_scratch_floats() : 0; d.sk_err_host = e.mlp_err_host; d.sk_max_spins = (unsigned)e.opts.debug_wait_spins;
      d.B = B; d.H = ai.H; d.W = ai.W; d.Cin = op.Cin; d.Cout = op.Cout; d.ks = op.ks; d.stride = op.stride;
      d.act = op.actfn; d.res_after_act = op.res_after; d.relu_from = op.relu_from;
      auto it = op.cfg.find(B);
      if (it == op.cfg.end()) it = op.cfg.emplace(B, conv_default_cfg(d)).first;
      if (op.wdev_h && (op.in.co & 31) == 0) {          // split-fp16 experiment: every plain 1x1 conv runs on ALG 12
        d.wfrag_h = op.wdev_h;
        const int nT16 = op.Cout / 16;
        const long Pout = (long)B * ((ai.H - 1) / op.stride + 1) * ((ai.W - 1) / op.stride + 1);
        const bool tiled = Pout >= 2048 && nT16 >= 4;      // 128 x 128 block tiles (LDS-staged, converted once per block)
        const ConvCfg hc{Pout >= 16384 ? 4 : 2, nT16 % 4 == 0 ? 4 : 2, 2, nT16 >= 8 ? 2 : 1, tiled ? 8 : 2, 1, 12};
        return conv_launch(d, hc, s);
      }
      if (it->second.ALG == 11) {
        // consecutive ALG 11 convs of a lane (the 7x7 branch chain): the output transform of this one produces the V of the next
        // one (wg_mid_kernel), and its own output tensor is written only if something else still reads it
        const int ln = op.lane & 3;
        if (e.wg_ready_act[ln] == op.in.act && op.in.co == 0) { d.wg_skip_in = 1; d.wg_vsel = e.wg_ready_vsel[ln]; }
        e.wg_ready_act[ln] = -1;
        const size_t k = (size_t)(&op - e.ops.data());
        if (e.opts.wg_fuse && k + 1 < e.ops.size()) {
          Op& nx = e.ops[k + 1];
          auto nit = nx.cfg.find(B);
          if (nx.type == OP_CONV && nx.phase == op.phase && nx.lane == op.lane && nx.in.act == op.out.act && nx.in.co == 0 &&
              op.out.co == 0 && nit != nx.cfg.end() && nit->second.ALG == 11 && nx.wdev_wino4g && ao.C == op.Cout &&
              conv_wino4g_can_chain(ai.H, ai.W, op.Cout, nx.Cin) &&
              std::max(op.Cin, op.Cout) == std::max(nx.Cin, nx.Cout)) {      // both convs split the lane's scratch into V | M the same way
            d.wg_emit_next = 1;
            d.wg_store_y = !(op.out.act < (int)e.act_uses.size() && e.act_uses[op.out.act] == 1);
            e.wg_ready_act[ln] = op.out.act;
            e.wg_ready_vsel[ln] = d.wg_vsel ^ 1;
          }
        }
      }
      return conv_launch(d, it->second, s);
    }
    case OP_MAXPOOL: {
      const Act& a = e.acts[op.in.act];
      launch_maxpool3x3s2(aptr(e, op.in), aptr(e, op.out), B, a.H, a.W, a.C, e.acts[op.out.act].C, s);
      return POCO_OK;
    }
    case OP_BILINEAR: {
      const Act& a = e.acts[op.in.act];
      launch_bilinear_up2x(aptr(e, op.in), aptr(e, op.out), B, a.H, a.W, a.C, s);
      return POCO_OK;
    }
    case OP_FUSE: {
      FuseArgs fa{};
      fa.n = op.fn;
      for (int k = 0; k < op.fn; ++k) {
        fa.src[k] = aptr(e, op.fsrc[k]); fa.shift[k] = op.fshift[k]; fa.src_cs[k] = e.acts[op.fsrc[k].act].C;
      }
      const Act& ao = e.acts[op.out.act];
      // the identity term (first with shift 0 that is a whole tensor) defines the output plane; op.C the width
      int H = 0, W = 0;
      for (int k = 0; k < op.fn; ++k)
        if (op.fshift[k] == 0) { const Act& a = e.acts[op.fsrc[k].act]; H = a.H; W = a.W; }
      launch_fuse_sum(fa, aptr(e, op.out), B, H, W, op.C, ao.C, op.frelu, s);
      return POCO_OK;
    }
    case OP_AVGPOOL: {
      const Act& a = e.acts[op.in.act];
      launch_avgpool(aptr(e, op.in), aptr(e, op.out), B, a.H, a.W, a.C, astride(e, op.out), s);
      return POCO_OK;
    }
    case OP_ATTN: {
      const Act& ah = e.acts[op.in.act];
      // scratch act was sized for one crop of C=128; its buffer is max_batch x that
      launch_part_attention_pool_ws(aptr(e, op.in), ah.C, aptr(e, op.in2), op.C, aptr(e, op.out), astride(e, op.out), B,
                                    ah.H, ah.W, sptr(e, e.a_attn_scratch), s);
      return POCO_OK;
    }
    case OP_LC2D:
      launch_lc2d_pose(aptr(e, op.in), astride(e, op.in), op.wdev, aptr(e, op.out), B, s);
      return POCO_OK;
    case OP_ROT6D: {
      float* y = ext_out(io, op.out2.ext);
      launch_rot6d(aptr(e, op.in), astride(e, op.in), aptr(e, op.out), astride(e, op.out), y, 216, B, s);
      return POCO_OK;
    }
    case OP_CHAIN: {
      const Act& a = e.acts[op.in.act];
      return launch_bneck_chain(aptr(e, op.in), a.C, aptr(e, op.res), e.acts[op.res.act].C, aptr(e, op.out), e.acts[op.out.act].C,
                                aptr(e, op.out2), e.acts[op.out2.act].C, op.wdev, op.bdev, op.wdev2, op.bdev2, B, a.H, a.W, s);
    }
    case OP_DUAL1X1: {
      const Act& a = e.acts[op.in.act];
      const Act& x = e.acts[op.in2.act];
      return launch_gemm1x1_dual(aptr(e, op.in), a.C, op.Cin, aptr(e, op.in2), x.C, op.C, x.H, x.W, op.stride, op.wdev, op.bdev,
                                 aptr(e, op.out), e.acts[op.out.act].C, op.Cout, B, a.H, a.W, op.actfn, s, e.opts.dual_layout);
    }
    case OP_MLP: {
      MlpProgram p{};
      p.B = B; p.sync = e.mlp_sync; p.err_host = e.mlp_err_host;
      p.max_spins = (unsigned)e.opts.debug_wait_spins;
      if (e.mlp_timeouts_left > 0) { --e.mlp_timeouts_left; p.debug_skip_arrival = 1; }
      int nl = 0, nr = 0, last = -1;
      for (const Op& su : op.sub) {
        if (su.stage < last || su.stage >= MLP_MAX_STAGES) { poco_set_error("forward: " + op.name + ": sub-ops out of stage order"); return POCO_ERR_STATE; }
        for (int sg = last + 1; sg <= su.stage; ++sg) p.stage[sg] = MlpStage{nl, 0, nr, 0};
        last = su.stage;
        MlpStage& st = p.stage[su.stage];
        if (su.type == OP_CONV) {
          const Act& ai = e.acts[su.in.act];
          const Act& ao = e.acts[su.out.act];
          if (nl >= MLP_MAX_LAYERS || su.ks != 1 || ai.H * ai.W != 1 || su.res_after || su.actfn > 2 || (su.Cin & 15) || (su.Cout & 15)) {
            poco_set_error("forward: " + su.name + " is not a Linear layer the fused regressor can run"); return POCO_ERR_STATE;
          }
          MlpLayer& L = p.layer[nl++];
          L.in = aptr(e, su.in); L.in_rs = ai.C;
          if (su.res.act >= 0) { L.res = aptr(e, su.res); L.res_rs = e.acts[su.res.act].C; }
          L.out = aptr(e, su.out); L.out_rs = ao.C;
          L.wfrag = reinterpret_cast<const float4*>(su.wdev); L.bias = su.bdev;
          L.nC16 = su.Cin / 16; L.nT16 = su.Cout / 16; L.act = su.actfn;
          ++st.nlayers;
        } else {
          if (nr >= MLP_MAX_ROWS) { poco_set_error("forward: " + op.name + ": too many row jobs"); return POCO_ERR_STATE; }
          MlpRowJob J{};
          if (su.type == OP_ROT6D) {
            J.kind = MLP_ROW_ROT6D;
            J.src = aptr(e, su.in); J.src_rs = astride(e, su.in);
            J.dst = aptr(e, su.out); J.dst_rs = astride(e, su.out);
            J.dst2 = ext_out(io, su.out2.ext); J.dst2_rs = 216;
          } else if (su.type == OP_BCAST) {
            J.kind = MLP_ROW_BCAST; J.src = su.wdev; J.dst = aptr(e, su.out); J.dst_rs = astride(e, su.out); J.n = su.n;
          } else if (su.type == OP_COPY) {
            J.kind = MLP_ROW_COPY; J.n = su.n;
            if (su.in.ext) { J.src = ext_in(io, su.in.ext); J.src_rs = su.n; if (!J.src) { poco_set_error("forward: missing input for " + su.name); return POCO_ERR_ARG; } }
            else { J.src = aptr(e, su.in); J.src_rs = astride(e, su.in); }
            if (su.out.ext) { J.dst = ext_out(io, su.out.ext); J.dst_rs = su.n; if (!J.dst) continue; }   // output not requested
            else { J.dst = aptr(e, su.out); J.dst_rs = astride(e, su.out); }
          } else { poco_set_error("forward: " + su.name + " cannot be a sub-op of the fused regressor"); return POCO_ERR_STATE; }
          p.row[nr++] = J;
          ++st.nrows;
        }
      }
      p.nstages = last + 1;
      return launch_mlp_chain(p, e.opts.mlp_blocks, s);
    }
    case OP_COPY: {
      const float* src; int sstride;
      if (op.in.ext) { src = ext_in(io, op.in.ext); sstride = op.n; if (!src) { poco_set_error("forward: missing input for " + op.name); return POCO_ERR_ARG; } }
      else { src = aptr(e, op.in); sstride = astride(e, op.in); }
      float* dst; int dstride;
      if (op.out.ext) { dst = ext_out(io, op.out.ext); dstride = op.n; if (!dst) return POCO_OK; }   // output not requested
      else { dst = aptr(e, op.out); dstride = astride(e, op.out); }
      launch_copy_rows(src, sstride, dst, dstride, op.n, B, s);
      return POCO_OK;
    }
    case OP_BCAST:
      launch_broadcast_rows(op.wdev, aptr(e, op.out), astride(e, op.out), op.n, B, s);
      return POCO_OK;
    case OP_SMPL: {
      SmplIO sio{};
      sio.betas = aptr(e, e.smpl_betas); sio.betas_stride = astride(e, e.smpl_betas);
      sio.rotmat = aptr(e, e.smpl_rot); sio.rot_stride = astride(e, e.smpl_rot);
      sio.A = sptr(e, e.a_A);
      sio.coef = sptr(e, e.a_coef);
      sio.joints24 = sptr(e, e.a_j24);
      float* yv = io.out->smpl_vertices;
      sio.verts = yv ? yv : sptr(e, e.a_verts);
      sio.joints49 = sptr(e, e.a_j49);
      sio.joints49_out = io.out->smpl_joints3d;
      launch_smpl_lbs(e.smpl, sio, B, s);
      return POCO_OK;
    }
    case OP_CAMERA: {
      CamArgs c{};
      c.cam = aptr(e, e.cam_ref); c.cam_stride = astride(e, e.cam_ref);
      c.joints49 = sptr(e, e.a_j49);
      c.cliff = op.n;
      if (c.cliff) {
        c.focal = io.in->focal_length; c.scale = io.in->scale; c.center = io.in->center; c.orig_shape = io.in->orig_shape;
        if (!c.focal || !c.scale || !c.center || !c.orig_shape) { poco_set_error("forward: cliff variant needs focal_length/scale/center/orig_shape"); return POCO_ERR_ARG; }
      }
      // outputs the caller did not ask for go to scratch rows (the kernel writes them densely)
      c.cam_t = io.out->pred_cam_t ? io.out->pred_cam_t : sptr(e, e.a_camt);
      c.fullimg_cam_t = io.out->pred_fullimg_cam_t ? io.out->pred_fullimg_cam_t : sptr(e, e.a_fullt);
      c.joints2d = io.out->smpl_joints2d ? io.out->smpl_joints2d : sptr(e, e.a_j2d);
      launch_camera(c, B, s);
      return POCO_OK;
    }
    case OP_RECORD: {
      float* y = ext_out(io, op.out.ext);
      if (!y) return POCO_OK;
      launch_pack_record(aptr(e, e.smpl_rot), astride(e, e.smpl_rot), aptr(e, e.smpl_betas), astride(e, e.smpl_betas),
                         aptr(e, e.cam_ref), astride(e, e.cam_ref), aptr(e, op.in), astride(e, op.in), y, op.n,
                         e.opts.rec_kinematic, e.opts.rec_thr, B, s);
      return POCO_OK;
    }
    case OP_NCHW_OUT: {
      float* y = ext_out(io, op.out.ext);
      if (!y) return POCO_OK;
      const Act& a = e.acts[op.in.act];
      launch_nhwc_to_nchw(aptr(e, op.in), a.C, y, B, a.H, a.W, op.C, s);
      return POCO_OK;
    }
  }
  return POCO_ERR_STATE;
}

Engine* H(poco_handle_t h) { return reinterpret_cast<Engine*>(h); }

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" int poco_abi_version(void) { return POCO_ABI_VERSION; }

extern "C" int poco_create(const char* variant, int max_batch, int num_flow_layers, poco_handle_t* out) {
  return poco_create_ex(variant, max_batch, num_flow_layers, nullptr, out);
}

extern "C" int poco_create_ex(const char* variant, int max_batch, int num_flow_layers, const char* options, poco_handle_t* out) {
  if (!variant || !out || max_batch < 1) { poco_set_error("poco_create: bad arguments"); return POCO_ERR_ARG; }
  std::string v(variant);
  const size_t dash = v.find('-');
  if (dash == std::string::npos) { poco_set_error("variant must be '<backbone>-<head>' (poco.py:41)"); return POCO_ERR_ARG; }
  auto e = std::make_unique<Engine>();
  e->backbone = v.substr(0, dash);
  e->head = v.substr(dash + 1);
  e->max_batch = max_batch;
  e->flow_layers = num_flow_layers;
  { std::string oerr; if (!parse_opts(options, &e->opts, &oerr)) { poco_set_error("poco_create_ex: " + oerr); return POCO_ERR_ARG; } }
  if (!build_graph(*e, /*declare=*/true)) { poco_set_error("poco_create: " + e->err); return POCO_ERR_ARG; }
  *out = reinterpret_cast<poco_handle_t>(e.release());
  return POCO_OK;
}

extern "C" void poco_destroy(poco_handle_t h) { delete H(h); }

extern "C" int poco_num_tensors(poco_handle_t h) { return h ? (int)H(h)->decls.size() : -1; }

extern "C" int poco_tensor_info(poco_handle_t h, int i, char* name, size_t name_cap, int64_t* shape, int* rank,
                                int* required) {
  Engine* e = H(h);
  if (!e || i < 0 || i >= (int)e->decls.size()) { poco_set_error("poco_tensor_info: index out of range"); return POCO_ERR_ARG; }
  const ParamDecl& d = e->decls[i];
  if (name && name_cap) { std::strncpy(name, d.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
  if (rank) *rank = (int)d.shape.size();
  if (shape) for (size_t k = 0; k < d.shape.size() && k < 6; ++k) shape[k] = d.shape[k];
  if (required) *required = d.required;
  return POCO_OK;
}

extern "C" int poco_load_tensor(poco_handle_t h, const char* name, const float* host_data, const int64_t* shape, int rank) {
  Engine* e = H(h);
  if (!e || !name || (!host_data && rank >= 0)) { poco_set_error("poco_load_tensor: bad arguments"); return POCO_ERR_ARG; }
  if (e->finalized) { poco_set_error("poco_load_tensor: engine already finalized"); return POCO_ERR_STATE; }
  auto it = e->decl_index.find(name);
  if (it == e->decl_index.end()) { poco_set_error(std::string("unexpected tensor: ") + name); return POCO_ERR_MISSING; }
  const ParamDecl& d = e->decls[it->second];
  size_t n = 1, nd = 1;
  for (int k = 0; k < rank; ++k) n *= (size_t)shape[k];
  for (int64_t s : d.shape) nd *= (size_t)s;
  // strict: the shapes must agree dimension by dimension once size-1 dimensions are dropped (a checkpoint may store
  // init_pose as [1,144] or [144]); the same element count in a different arrangement (a transposed Linear weight,
  // shapedirs stored [10,V,3]) is an error, as it is for the reference's load_state_dict
  std::vector<int64_t> got, want;
  for (int k = 0; k < rank; ++k) if (shape[k] != 1) got.push_back(shape[k]);
  for (int64_t s : d.shape) if (s != 1) want.push_back(s);
  if (n != nd || got != want) {
    auto fmt = [](const int64_t* v, size_t m) { std::string t = "["; for (size_t k = 0; k < m; ++k) t += (k ? "," : "") + std::to_string(v[k]); return t + "]"; };
    poco_set_error(std::string("shape mismatch for ") + name + ": got " + fmt(shape, (size_t)rank) + ", expected " + fmt(d.shape.data(), d.shape.size()));
    return POCO_ERR_SHAPE;
  }
  HostParam p;
  p.shape.assign(shape, shape + rank);
  p.data.assign(host_data, host_data + n);
  e->params[name] = std::move(p);
  return POCO_OK;
}

static int ensure_sk_scratch(Engine* e, int lane);
extern "C" int poco_finalize(poco_handle_t h) {
  Engine* e = H(h);
  if (!e) return POCO_ERR_ARG;
  if (e->finalized) return POCO_OK;
  std::string missing;
  for (const ParamDecl& d : e->decls)
    if (d.required && !e->params.count(d.name)) missing += d.name + " ";
  if (!missing.empty()) { poco_set_error("poco_finalize: missing required tensors: " + missing); return POCO_ERR_MISSING; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { poco_set_error("poco_finalize: no HIP device (there is no CPU fallback)"); return POCO_ERR_HIP; }
  e->err.clear();
  if (!build_graph(*e, /*declare=*/false)) { poco_set_error("poco_finalize: " + e->err); return POCO_ERR_STATE; }
  plan_workspace(*e);
  POCO_HIP_CHECK(hipMalloc(&e->ws, e->ws_floats * sizeof(float)));
  POCO_HIP_CHECK(hipMemset(e->ws, 0, e->ws_floats * sizeof(float)));
  e->act_uses.assign(e->acts.size(), 0);
  for (const Op& op : e->ops) {
    auto use = [&](const Ref& r) { if (r.act >= 0 && r.act < (int)e->act_uses.size()) ++e->act_uses[r.act]; };
    use(op.in); use(op.in2); use(op.res);
    for (const Op& su : op.sub) { use(su.in); use(su.res); }
    for (int k = 0; k < op.fn && k < 4; ++k) use(op.fsrc[k]);
  }
  if (e->wino4g_scratch_need)          // ALG 11 staging: one buffer per lane (ops of different lanes run concurrently)
    for (int k = 0; k < 4; ++k) POCO_HIP_CHECK(hipMalloc(&e->wino4g_scratch[k], e->wino4g_scratch_need * sizeof(float)));
  if (e->has_flow) {                   // RealNVP step A scratch: planned here, poco_realnvp* never allocates
    const int rows = e->opts.flow_ctx_rows > 0 ? e->opts.flow_ctx_rows : e->max_batch;
    e->flow_scratch_floats = realnvp_scratch_floats(e->flow, rows);
    POCO_HIP_CHECK(hipMalloc(&e->flow_scratch, e->flow_scratch_floats * sizeof(float)));
  }
  POCO_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&e->mlp_err_host), 64, hipHostMallocMapped));
  *e->mlp_err_host = 0;
  for (const Op& op : e->ops)
    if (op.type == OP_MLP && !e->mlp_sync) {
      POCO_HIP_CHECK(hipMalloc(&e->mlp_sync, 1024));
      POCO_HIP_CHECK(hipMemset(e->mlp_sync, 0, 1024));
    }
  e->mlp_timeouts_left = e->opts.debug_mlp_timeouts;
  // (ALG 14's flags + partials - 33 MB per lane - are allocated when a stream-K configuration is first accepted for an op of that
  // lane: ensure_sk_scratch; configurations set before finalize get theirs here)
  for (const Op& op : e->ops)
    for (const auto& kv : op.cfg)
      if (kv.second.ALG == 14)
        if (int rc = ensure_sk_scratch(e, op.lane)) return rc;
  POCO_HIP_CHECK(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
  for (Op& op : e->ops)
    if (op.wait_mask) {
      op.dep_ev = (int)e->ev_dep.size();
      for (int k = 0; k < 4; ++k)
        if (op.wait_mask & (1u << k)) {
          hipEvent_t ev;
          POCO_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
          e->ev_dep.push_back(ev);
        }
    }
  for (int k = 1; k < 4; ++k) {
    POCO_HIP_CHECK(hipStreamCreateWithFlags(&e->lane_stream[k], hipStreamNonBlocking));
    POCO_HIP_CHECK(hipEventCreateWithFlags(&e->ev_join[k], hipEventDisableTiming));
  }
  POCO_HIP_CHECK(hipDeviceSynchronize());
  e->params.clear();   // host copies no longer needed
  e->finalized = true;
  return POCO_OK;
}

// Enqueue the whole program on the caller's stream (lane 0) and the side lanes.
static int enqueue_program(Engine* e, int B, const IO& io, hipStream_t main) {
  hipStream_t* side = e->lane_stream;
  hipEvent_t fork_ev = e->ev_fork;
  hipEvent_t* join_ev = e->ev_join;
  const int nops = (int)e->ops.size();
  for (int l = 0; l < 4; ++l) e->wg_ready_act[l] = -1;
  for (int i = 0; i < nops;) {
    int j = i;
    unsigned lanes = 0;
    while (j < nops && e->ops[j].phase == e->ops[i].phase) { lanes |= 1u << e->ops[j].lane; ++j; }
    const bool fork = e->num_lanes > 1 && (lanes & ~1u);
    if (fork) {
      // fork: the side lanes wait for everything enqueued so far on the main stream
      POCO_HIP_CHECK(hipEventRecord(fork_ev, main));
      for (int k = 1; k < 4; ++k)
        if (lanes & (1u << k)) POCO_HIP_CHECK(hipStreamWaitEvent(side[k], fork_ev, 0));
    }
    for (int k = i; k < j; ++k) {
      Op& op = e->ops[k];
      hipStream_t s = (fork && op.lane > 0) ? side[op.lane] : main;
      if (fork && op.wait_mask) {
        // cross-lane dependencies: everything enqueued so far on those lanes (program order = enqueue order)
        int n = 0;
        for (int l = 0; l < 4; ++l)
          if (op.wait_mask & (1u << l)) {
            hipEvent_t ev = e->ev_dep[op.dep_ev + n++];
            POCO_HIP_CHECK(hipEventRecord(ev, l > 0 ? side[l] : main));
            POCO_HIP_CHECK(hipStreamWaitEvent(s, ev, 0));
          }
      }
      int rc = run_op(*e, op, B, io, s);
      if (rc != POCO_OK) return rc;
    }
    if (fork) {
      // join: the main stream continues only after every side lane of this phase is done
      for (int k = 1; k < 4; ++k)
        if (lanes & (1u << k)) {
          POCO_HIP_CHECK(hipEventRecord(join_ev[k], side[k]));
          POCO_HIP_CHECK(hipStreamWaitEvent(main, join_ev[k], 0));
        }
    }
    i = j;
  }
  return POCO_OK;
}

// ABI 4: the caller's I/O structs say how many bytes they have (include/poco_hip.h).  Copy exactly those into a zeroed struct of
// THIS library's layout: members the caller's header does not have read as NULL, a non-NULL member this library does not know is
// refused, and a size no header of this ABI could have produced (0, not 8-byte granular, a pointer's bit pattern) is refused - a
// binding written from an older field list cannot make the engine read past its struct any more.
template <class T>
static int sized_struct_copy(const T* src, T* dst, const char* what) {
  uint64_t sz;
  std::memcpy(&sz, src, sizeof sz);
  if (sz < 2 * sizeof(void*) || (sz & 7) || sz > 4096) {
    poco_set_error(std::string(what) + ".struct_size = " + std::to_string(sz) + " is not the size of any " + what +
                   " of ABI " + std::to_string(POCO_ABI_VERSION) + " (set it to sizeof(" + what + "); a struct laid out for ABI <= 3 starts with a pointer)");
    return POCO_ERR_ARG;
  }
  std::memset(dst, 0, sizeof(T));
  std::memcpy(dst, src, (size_t)std::min<uint64_t>(sz, sizeof(T)));
  const unsigned char* extra = reinterpret_cast<const unsigned char*>(src);
  for (uint64_t off = sizeof(T); off < sz; ++off)
    if (extra[off]) {
      poco_set_error(std::string(what) + " carries a non-NULL member at byte " + std::to_string(off) + ", beyond the " +
                     std::to_string(sizeof(T)) + " bytes this library knows: it was built from an older header");
      return POCO_ERR_ARG;
    }
  dst->struct_size = sizeof(T);
  return POCO_OK;
}

// ALG 14 (stream-K 1x1 GEMM): flags (zero between launches) + partial accumulators of the lane the op runs on, allocated when the first
// stream-K configuration is accepted for an op of that lane (poco_set_conv_cfg: before any forward / graph capture that could use it) -
// engines whose table holds no ALG 14 entry (every HRNet-PARE engine, the 8-engines-on-one-GPU rehearsal) never pay the 134 MB.
static int ensure_sk_scratch(Engine* e, int lane) {
  float*& q = e->sk_scratch[lane & 3];
  if (q) return POCO_OK;
  POCO_HIP_CHECK(hipMalloc(&q, gemm1x1sk_scratch_floats() * sizeof(float)));
  POCO_HIP_CHECK(hipMemset(q, 0, (size_t)SK_MAX_WAVES * sizeof(float)));
  return POCO_OK;
}

// See include/poco_hip.h.  The in-kernel waits (grid barrier of the fused regressor, partial hand-off of the stream-K GEMM) are bounded; one
// that runs out raises a word in pinned host memory and its launch ends early - outputs invalid, queue alive.  The caller synchronises the
// stream and asks here: the word is read AND cleared, and whatever the aborted launch may have left behind on the device (stream-K flags still
// raised, the regressor's arrival counters if blocks left in different stages) is re-armed, so that the next forward is a normal one.
extern "C" int poco_status(poco_handle_t h) {
  Engine* e = H(h);
  if (!e) { poco_set_error("poco_status: bad handle"); return POCO_ERR_ARG; }
  if (!e->finalized || !e->mlp_err_host) return POCO_OK;
  volatile unsigned* w = reinterpret_cast<volatile unsigned*>(e->mlp_err_host);
  if (*w == 0u) return POCO_OK;
  // nothing of this engine may be running while the device state is reset: the documented contract is "after a stream synchronise"; make it
  // true here rather than trust it (this is the failure path, not the hot path)
  POCO_HIP_CHECK(hipDeviceSynchronize());
  *w = 0u;
  if (e->mlp_sync) POCO_HIP_CHECK(hipMemset(e->mlp_sync, 0, 1024));
  for (float* q : e->sk_scratch)
    if (q) POCO_HIP_CHECK(hipMemset(q, 0, (size_t)SK_MAX_WAVES * sizeof(float)));
  poco_set_error("poco_status: a bounded in-kernel wait (grid barrier of the fused regressor, mlp_chain.hip, or partial hand-off of a stream-K GEMM, "
                 "ALG 14) timed out since the last call: the outputs of the forwards enqueued since then are INVALID.  The engine has been re-armed "
                 "and can run again; if it recurs, build it with the option mlp_fuse=0 / a table without ALG 14 entries");
  return POCO_ERR_HIP;
}

extern "C" int poco_forward(poco_handle_t h, int B, const poco_inputs_t* in, const poco_outputs_t* out, void* stream) {
  Engine* e = H(h);
  if (!e || !in || !out) { poco_set_error("poco_forward: bad arguments"); return POCO_ERR_ARG; }
  poco_inputs_t in_l;
  poco_outputs_t out_l;
  if (int rc = sized_struct_copy(in, &in_l, "poco_inputs_t")) return rc;            // (checked first: testable without a GPU)
  if (int rc = sized_struct_copy(out, &out_l, "poco_outputs_t")) return rc;
  if (!e->finalized) { poco_set_error("poco_forward: call poco_finalize first"); return POCO_ERR_STATE; }
  if (B < 1 || B > e->max_batch) { poco_set_error("poco_forward: batch " + std::to_string(B) + " outside 1.." + std::to_string(e->max_batch)); return POCO_ERR_ARG; }
  if (e->mlp_err_host && *reinterpret_cast<volatile unsigned*>(e->mlp_err_host)) {
    // (a caller that never asks poco_status still cannot run on top of an aborted forward unnoticed)
    poco_set_error("poco_forward: a bounded wait of the fused regressor (mlp_chain.hip) or of a stream-K GEMM (ALG 14) timed out in an earlier "
                   "forward - its outputs are invalid; synchronise the stream and call poco_status(), which reports it and re-arms the engine");
    return POCO_ERR_HIP;
  }
  hipStream_t caller = (hipStream_t)stream;
  IO io{&in_l, &out_l};
  int rc = enqueue_program(e, B, io, caller);
  if (rc != POCO_OK) return rc;
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) { poco_set_error(std::string("poco_forward: ") + hipGetErrorString(err)); return POCO_ERR_HIP; }
  return POCO_OK;
}

extern "C" int poco_set_num_lanes(poco_handle_t h, int n) {
  Engine* e = H(h);
  if (!e || n < 1 || n > 4) { poco_set_error("poco_set_num_lanes: n must be 1..4"); return POCO_ERR_ARG; }
  e->num_lanes = n;
  return POCO_OK;
}

extern "C" int poco_num_ops(poco_handle_t h) { return h ? (int)H(h)->ops.size() : -1; }

extern "C" int poco_op_info(poco_handle_t h, int i, char* name, size_t name_cap, double* flops_per_crop, int* type) {
  Engine* e = H(h);
  if (!e || i < 0 || i >= (int)e->ops.size()) { poco_set_error("poco_op_info: index out of range"); return POCO_ERR_ARG; }
  const Op& op = e->ops[i];
  if (name && name_cap) { std::strncpy(name, op.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
  if (flops_per_crop) *flops_per_crop = op.flops;
  if (type) *type = op.type;
  return POCO_OK;
}

extern "C" int poco_op_sched(poco_handle_t h, int i, int* sched, int cap) {
  Engine* e = H(h);
  if (!e || !sched || i < 0 || i >= (int)e->ops.size()) { poco_set_error("poco_op_sched: bad arguments"); return POCO_ERR_ARG; }
  const Op& op = e->ops[i];
  std::vector<OpAccess> rd, wr;
  op_accesses(*e, op, &rd, &wr);
  std::vector<int> v = {op.phase, op.lane, (int)op.wait_mask, (int)rd.size()};
  for (const OpAccess& a : rd) { v.push_back(a.act); v.push_back(a.lo); v.push_back(a.hi); }
  v.push_back((int)wr.size());
  for (const OpAccess& a : wr) { v.push_back(a.act); v.push_back(a.lo); v.push_back(a.hi); }
  if ((int)v.size() > cap) { poco_set_error("poco_op_sched: buffer too small"); return POCO_ERR_ARG; }
  std::copy(v.begin(), v.end(), sched);
  return POCO_OK;
}

extern "C" int poco_profile_ops(poco_handle_t h, int B, const poco_inputs_t* in, const poco_outputs_t* out, int iters,
                                float* ms_per_op, int cap, void* stream) {
  Engine* e = H(h);
  if (!e || !e->finalized || !ms_per_op || !in || !out) { poco_set_error("poco_profile_ops: bad state/arguments"); return POCO_ERR_ARG; }
  poco_inputs_t in_l;
  poco_outputs_t out_l;
  if (int rc = sized_struct_copy(in, &in_l, "poco_inputs_t")) return rc;
  if (int rc = sized_struct_copy(out, &out_l, "poco_outputs_t")) return rc;
  const int n = (int)e->ops.size();
  if (cap < n) { poco_set_error("poco_profile_ops: buffer too small"); return POCO_ERR_ARG; }
  hipStream_t s = (hipStream_t)stream;
  std::vector<hipEvent_t> ev(n + 1);
  for (auto& x : ev) POCO_HIP_CHECK(hipEventCreate(&x));
  std::vector<double> acc(n, 0.0);
  IO io{&in_l, &out_l};
  for (int it = 0; it < iters + 1; ++it) {
    for (int l = 0; l < 4; ++l) e->wg_ready_act[l] = -1;
    POCO_HIP_CHECK(hipEventRecord(ev[0], s));
    for (int i = 0; i < n; ++i) {
      int rc = run_op(*e, e->ops[i], B, io, s);
      if (rc != POCO_OK) return rc;
      POCO_HIP_CHECK(hipEventRecord(ev[i + 1], s));
    }
    POCO_HIP_CHECK(hipStreamSynchronize(s));
    if (int rc = poco_status(h)) return rc;      // a timed-out wait makes the times (and outputs) of this pass meaningless
    if (it == 0) continue;   // warm-up
    for (int i = 0; i < n; ++i) { float ms = 0; POCO_HIP_CHECK(hipEventElapsedTime(&ms, ev[i], ev[i + 1])); acc[i] += ms; }
  }
  for (int i = 0; i < n; ++i) ms_per_op[i] = (float)(acc[i] / iters);
  for (auto& x : ev) (void)hipEventDestroy(x);
  return POCO_OK;
}

extern "C" size_t poco_workspace_bytes(poco_handle_t h) { return h ? H(h)->ws_floats * sizeof(float) : 0; }

extern "C" int poco_uncert_feat_dim(poco_handle_t h) { return h ? H(h)->uncert_feat_dim : -1; }

extern "C" int poco_set_conv_cfg(poco_handle_t h, int op_index, int B, const int* cfg7) {
  Engine* e = H(h);
  if (!e || op_index < 0 || op_index >= (int)e->ops.size() || !cfg7 || e->ops[op_index].type != OP_CONV) {
    poco_set_error("poco_set_conv_cfg: bad arguments");
    return POCO_ERR_ARG;
  }
  // validate against this op's geometry at batch B (a table entry measured at another batch size may not fit)
  const Op& op = e->ops[op_index];
  const Act& ai = e->acts[op.in.act];
  ConvDesc d{};
  d.B = B; d.H = ai.H; d.W = ai.W; d.Cin = op.Cin; d.Cout = op.Cout; d.ks = op.ks; d.stride = op.stride;
  d.in_cs = ai.C; d.out_cs = e->acts[op.out.act].C; d.act = op.actfn;
  const ConvCfg c = conv_cfg_from(cfg7);
  const size_t lds = conv_lds_bytes(d, c);
  if (B < 1 || lds == 0 || lds > 160 * 1024 || ((c.ALG == 3 || c.ALG == 4) && (op.wdev_wino == nullptr && e->finalized)) ||
      ((c.ALG == 3 || c.ALG == 4) && (op.actfn == 3 || op.actfn == 2)) ||
      (c.ALG == 7 && ((op.wdev_wino4 == nullptr && e->finalized) || ai.H < 28 || ai.W < 28 || op.actfn >= 2)) ||
      (c.ALG == 8 && ((op.wdev_wino4p == nullptr && e->finalized) || ai.H < 14 || ai.W < 14 || op.actfn >= 2)) ||
      (c.ALG == 13 && ((op.wdev_wino4w == nullptr && e->finalized) || ai.H < e->opts.w4_min_plane || ai.W < e->opts.w4_min_plane || op.actfn >= 2)) ||
      (c.ALG == 11 && ((op.wdev_wino4g == nullptr && e->finalized) || ai.H > e->opts.wg_max_plane || ai.W > e->opts.wg_max_plane || ai.H * ai.W <= 1 || op.actfn >= 2))) {     // (its scratch is sized for max_batch; poco_forward refuses larger batches)
    poco_set_error("poco_set_conv_cfg: configuration does not fit op '" + op.name + "' at this batch size");
    return POCO_ERR_ARG;
  }
  if (c.ALG == 14 && e->finalized)
    if (int rc = ensure_sk_scratch(e, op.lane)) return rc;
  e->ops[op_index].cfg[B] = c;
  return POCO_OK;
}

extern "C" int poco_get_conv_cfg(poco_handle_t h, int op_index, int B, int* cfg7) {
  Engine* e = H(h);
  if (!e || op_index < 0 || op_index >= (int)e->ops.size() || !cfg7 || e->ops[op_index].type != OP_CONV || B < 1) {
    poco_set_error("poco_get_conv_cfg: bad arguments");
    return POCO_ERR_ARG;
  }
  const Op& op = e->ops[op_index];
  ConvCfg c;
  auto it = op.cfg.find(B);
  if (it != op.cfg.end()) c = it->second;
  else {
    const Act& ai = e->acts[op.in.act];
    ConvDesc d{};
    d.B = B; d.H = ai.H; d.W = ai.W; d.Cin = op.Cin; d.Cout = op.Cout; d.ks = op.ks; d.stride = op.stride;
    d.in_cs = ai.C; d.out_cs = e->acts[op.out.act].C; d.act = op.actfn;
    c = conv_default_cfg(d);
  }
  const int v[7] = {c.MT, c.NT, c.WM, c.WN, c.R, c.NI, c.ALG};
  for (int k = 0; k < 7; ++k) cfg7[k] = v[k];
  return POCO_OK;
}

extern "C" int poco_get_conv_desc(poco_handle_t h, int op_index, int* desc8) {
  Engine* e = H(h);
  if (!e || op_index < 0 || op_index >= (int)e->ops.size() || !desc8) return POCO_ERR_ARG;
  const Op& op = e->ops[op_index];
  if (op.type != OP_CONV) return POCO_ERR_ARG;
  const Act& ai = e->acts[op.in.act];
  desc8[0] = ai.H; desc8[1] = ai.W; desc8[2] = op.Cin; desc8[3] = op.Cout; desc8[4] = op.ks; desc8[5] = op.stride;
  desc8[6] = ai.C; desc8[7] = e->acts[op.out.act].C;
  return POCO_OK;
}

// ---- stand-alone SMPL / RealNVP operators bound to an engine's loaded model -----------------------
extern "C" int poco_smpl_lbs(poco_handle_t h, int B, const float* d_betas, const float* d_rotmat, float* d_verts,
                             float* d_joints49, void* stream) {
  Engine* e = H(h);
  if (!e || !e->finalized || !d_betas || !d_rotmat || !d_verts || !d_joints49 || B < 1 || B > e->max_batch) {
    poco_set_error("poco_smpl_lbs: bad state/arguments");
    return POCO_ERR_ARG;
  }
  SmplIO sio{};
  sio.betas = d_betas; sio.betas_stride = 10;
  sio.rotmat = d_rotmat; sio.rot_stride = 216;
  sio.A = e->ws + e->acts[e->a_A].off;
  sio.coef = e->ws + e->acts[e->a_coef].off;
  sio.joints24 = e->ws + e->acts[e->a_j24].off;
  sio.verts = d_verts;
  sio.joints49 = d_joints49;
  launch_smpl_lbs(e->smpl, sio, B, (hipStream_t)stream);
  return POCO_OK;
}

extern "C" int poco_realnvp_rep(poco_handle_t h, int N, const float* d_x, const float* d_ctx, int rep, float* d_out,
                                int forward, void* stream) {
  Engine* e = H(h);
  if (!e || !e->finalized || !e->has_flow) { poco_set_error("poco_realnvp: flow_head.flow.* tensors were not loaded"); return POCO_ERR_STATE; }
  if (!d_x || !d_ctx || !d_out || N < 1 || rep < 1) { poco_set_error("poco_realnvp: bad arguments"); return POCO_ERR_ARG; }
  const int ctx_rows = (N + rep - 1) / rep;
  if (realnvp_scratch_floats(e->flow, ctx_rows) > e->flow_scratch_floats) {
    poco_set_error("poco_realnvp: " + std::to_string(ctx_rows) + " context rows, the engine's scratch was planned for " +
                   std::to_string(e->opts.flow_ctx_rows > 0 ? e->opts.flow_ctx_rows : e->max_batch) +
                   " (create the engine with poco_create_ex(..., \"flow_ctx_rows=<n>\"); default = max_batch, one context per crop)");
    return POCO_ERR_ARG;
  }
  return launch_realnvp(e->flow, d_x, d_ctx, rep, d_out, N, forward, e->flow_scratch, (hipStream_t)stream);
}

extern "C" int poco_realnvp(poco_handle_t h, int N, const float* d_x, const float* d_ctx, float* d_out, int forward,
                            void* stream) {
  return poco_realnvp_rep(h, N, d_x, d_ctx, 1, d_out, forward, stream);
}
