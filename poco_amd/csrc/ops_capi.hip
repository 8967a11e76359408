// C-ABI entry points for the stand-alone operators (used by tests, the tuner and bench.py).
// Declarations + reference citations: include/poco_hip.h.
#include "../../include/poco_hip.h"
#include "common.h"

#include <cmath>
#include <mutex>
#include <vector>

static thread_local std::string g_last_error;
void poco_set_error(const std::string& msg) { g_last_error = msg; }

extern "C" const char* poco_last_error(void) { return g_last_error.c_str(); }

namespace {
struct DevBuf {
  float* p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  hipError_t upload(const std::vector<float>& h) {
    hipError_t e = hipMalloc(&p, h.size() * sizeof(float));
    if (e != hipSuccess) return e;
    return hipMemcpy(p, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
  }
};
}  // namespace

static int conv_common(const float* d_in, int B, int H, int W, int Cin, const float* h_w,
                       const float* h_scale, const float* h_shift, int Cout, int ks, int stride,
                       const float* d_res, int relu, float* d_out, const int* cfg7, int iters,
                       float* ms_out, hipStream_t stream) {
  if (!d_in || !h_w || !d_out) {
    poco_set_error("conv2d: null pointer");
    return POCO_ERR_ARG;
  }
  const int Cout16 = (Cout + 15) / 16 * 16;
  if (Cout16 != Cout) {
    poco_set_error("conv2d op: Cout must be a multiple of 16 (the engine pads; the bare op does not)");
    return POCO_ERR_ARG;
  }
  std::vector<float> packed(conv_packed_weight_floats(Cin, Cout16, ks));
  conv_pack_weights(h_w, h_scale, Cout, Cin, ks, Cout16, packed.data());
  std::vector<float> shift(Cout16, 0.f);
  if (h_shift)
    for (int i = 0; i < Cout; ++i) shift[i] = h_shift[i];
  DevBuf dw, db, dwu, dwu4, dwu4p, dwu4w, dwu4g, dscr, dsk;
  POCO_HIP_CHECK(dw.upload(packed));
  POCO_HIP_CHECK(db.upload(shift));
  ConvDesc d{};
  DevBuf dwh;
#if POCO_EXPERIMENTS
  if (ks == 1 && cfg7 && cfg7[6] == 12 && Cin % 32 == 0) {       // split-fp16 experiment: hi / lo halves of the weights
    std::vector<float> ph(gemm1x1h_packed_floats(Cin, Cout16));
    gemm1x1h_pack_weights(h_w, h_scale, Cout, Cin, Cout16, ph.data());
    POCO_HIP_CHECK(dwh.upload(ph));
    d.wfrag_h = dwh.p;
  }
#endif
  if (ks == 3 && stride == 1) {
    std::vector<float> wt, pu(conv_packed_weight_floats(Cin, Cout16, 4));
    conv_wino_transform_weights(h_w, Cout, Cin, &wt);
    conv_pack_weights(wt.data(), h_scale, Cout, Cin, 4, Cout16, pu.data());
    POCO_HIP_CHECK(dwu.upload(pu));
    d.wfrag_wino = dwu.p;
    if (cfg7 && cfg7[6] == 7) {                   // F(4x4,3x3): 36-position fragments
      std::vector<float> pu4(conv_wino4_packed_floats(Cin, Cout16));
      conv_wino4_pack_weights(h_w, h_scale, Cout, Cin, Cout16, pu4.data());
      POCO_HIP_CHECK(dwu4.upload(pu4));
      d.wfrag_wino4 = dwu4.p;
    }
    if (cfg7 && cfg7[6] == 8) {                   // the same fragments in the LDS order of the specialised-wave kernel
      std::vector<float> pu4(conv_wino4p_packed_floats(Cin, Cout16));
      conv_wino4p_pack_weights(h_w, h_scale, Cout, Cin, Cout16, pu4.data());
      POCO_HIP_CHECK(dwu4p.upload(pu4));
      d.wfrag_wino4p = dwu4p.p;
    }
    if (cfg7 && cfg7[6] == 13) {                  // ... and in the quad order of the whole-position kernel
      std::vector<float> pu4(conv_wino4w_packed_floats(Cin, Cout16));
      conv_wino4w_pack_weights(h_w, h_scale, Cout, Cin, Cout16, pu4.data());
      POCO_HIP_CHECK(dwu4w.upload(pu4));
      d.wfrag_wino4w = dwu4w.p;
    }
    if (cfg7 && cfg7[6] == 11) {                  // F(4x4,3x3) as 36 position GEMMs: per-position fragments + V / M staging
      std::vector<float> pg(conv_wino4g_packed_floats(Cin, Cout16));
      conv_wino4g_pack_weights(h_w, h_scale, Cout, Cin, Cout16, pg.data());
      POCO_HIP_CHECK(dwu4g.upload(pg));
      d.wfrag_wino4g = dwu4g.p;
      d.scratch_floats = conv_wino4g_scratch_floats(B, H, W, Cin, Cout16);
      POCO_HIP_CHECK(hipMalloc(&dscr.p, d.scratch_floats * sizeof(float)));
      d.scratch = dscr.p;
    }
  }
  if (cfg7 && cfg7[6] == 14) {                    // stream-K 1x1 GEMM: flags (zero) + partial accumulators, the pinned error word
    static unsigned* sk_err = nullptr;
    if (!sk_err) { POCO_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&sk_err), 64, hipHostMallocMapped)); *sk_err = 0; }
    d.sk_scratch_floats = gemm1x1sk_scratch_floats();
    POCO_HIP_CHECK(hipMalloc(&dsk.p, d.sk_scratch_floats * sizeof(float)));
    POCO_HIP_CHECK(hipMemset(dsk.p, 0, (size_t)SK_MAX_WAVES * sizeof(float)));
    d.sk_scratch = dsk.p;
    d.sk_err_host = sk_err;
  }
  d.in = d_in; d.in_cs = Cin; d.in_co = 0;
  d.res = d_res; d.res_cs = Cout; d.res_co = 0;
  d.out = d_out; d.out_cs = Cout; d.out_co = 0;
  d.wfrag = dw.p; d.bias = db.p;
  d.B = B; d.H = H; d.W = W; d.Cin = Cin; d.Cout = Cout16;
  d.ks = ks; d.stride = stride; d.act = relu; d.res_after_act = 0;
  ConvCfg cfg = conv_default_cfg(d);
  if (cfg7 && cfg7[0] > 0) cfg = conv_cfg_from(cfg7);
  int rc = conv_launch(d, cfg, stream);
  if (rc != POCO_OK) return rc;
  if (iters > 0 && ms_out) {
    hipEvent_t e0, e1;
    POCO_HIP_CHECK(hipEventCreate(&e0));
    POCO_HIP_CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) conv_launch(d, cfg, stream);
    POCO_HIP_CHECK(hipEventRecord(e0, stream));
    for (int i = 0; i < iters; ++i) conv_launch(d, cfg, stream);
    POCO_HIP_CHECK(hipEventRecord(e1, stream));
    POCO_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    POCO_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    *ms_out = ms / iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  }
  POCO_HIP_CHECK(hipStreamSynchronize(stream));
  if (d.sk_err_host && *reinterpret_cast<volatile unsigned*>(d.sk_err_host)) { poco_set_error("conv: a wait of the stream-K GEMM (ALG 14) timed out"); return POCO_ERR_HIP; }
  return POCO_OK;
}

extern "C" int poco_op_conv2d(const float* d_in, int B, int H, int W, int Cin, const float* h_weight,
                              const float* h_scale, const float* h_shift, int Cout, int ks, int stride,
                              const float* d_res, int relu, float* d_out, const int* cfg7,
                              void* stream) {
  return conv_common(d_in, B, H, W, Cin, h_weight, h_scale, h_shift, Cout, ks, stride, d_res, relu,
                     d_out, cfg7, 0, nullptr, (hipStream_t)stream);
}

extern "C" int poco_bench_conv2d(const float* d_in, int B, int H, int W, int Cin, const float* h_weight,
                                 int Cout, int ks, int stride, float* d_out, const int* cfg7, int iters,
                                 float* ms_out, int* cfg_used7, void* stream) {
  if (cfg_used7) {
    ConvDesc d{};
    d.B = B; d.H = H; d.W = W; d.Cin = Cin; d.Cout = Cout; d.ks = ks; d.stride = stride;
    ConvCfg c = (cfg7 && cfg7[0] > 0) ? conv_cfg_from(cfg7) : conv_default_cfg(d);
    cfg_used7[0] = c.MT; cfg_used7[1] = c.NT; cfg_used7[2] = c.WM;
    cfg_used7[3] = c.WN; cfg_used7[4] = c.R;  cfg_used7[5] = c.NI; cfg_used7[6] = c.ALG;
  }
  return conv_common(d_in, B, H, W, Cin, h_weight, nullptr, nullptr, Cout, ks, stride, nullptr, 1, d_out,
                     cfg7, iters, ms_out, (hipStream_t)stream);
}

// Time a list of tile configurations for one conv shape (weights/activations allocated and filled
// here once).  ms_out[i] < 0 marks a configuration that is invalid for the shape.
extern "C" int poco_tune_conv(int B, int H, int W, int Cin, int Cout, int ks, int stride, const int* cfgs7,
                              int ncfg, int iters, float* ms_out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!cfgs7 || !ms_out || ncfg < 1 || Cin % 16 || Cout % 16) {
    poco_set_error("poco_tune_conv: bad arguments");
    return POCO_ERR_ARG;
  }
  const bool with_res = iters < 0;                        // iters < 0: |iters| launches of the residual form
  if (with_res) {
    iters = -iters;
    if (Cin != Cout || stride != 1) { poco_set_error("poco_tune_conv: iters < 0 (residual form) needs Cin == Cout and stride 1"); return POCO_ERR_ARG; }
  }
  const int pad = (ks - 1) / 2;
  const int Ho = (H + 2 * pad - ks) / stride + 1, Wo = (W + 2 * pad - ks) / stride + 1;
  const size_t nin = (size_t)B * H * W * Cin, nout = (size_t)B * Ho * Wo * Cout;
  const size_t nw = conv_packed_weight_floats(Cin, Cout, ks);
  uint32_t st = 12345u;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 32768.0f - 1.0f; };
  std::vector<float> hin(nin), hw(nw), hb(Cout);
  for (auto& v : hin) v = rnd();
  const float ws = 1.0f / sqrtf((float)(Cin * ks * ks));
  for (auto& v : hw) v = rnd() * ws;
  for (auto& v : hb) v = rnd() * 0.1f;
  DevBuf din, dw, db, dout, dwu, dwu4, dwu4g, dscr;
  bool any11 = false;
  for (int i = 0; i < ncfg; ++i) any11 = any11 || cfgs7[CONV_CFG_INTS * i + 6] == 11;
  if (ks == 3 && stride == 1) {
    std::vector<float> hu((size_t)16 * Cin * Cout);
    for (auto& v : hu) v = rnd() * ws;
    POCO_HIP_CHECK(dwu.upload(hu));
    bool any7 = false;
    for (int i = 0; i < ncfg; ++i) any7 = any7 || cfgs7[CONV_CFG_INTS * i + 6] == 7 || cfgs7[CONV_CFG_INTS * i + 6] == 8 || cfgs7[CONV_CFG_INTS * i + 6] == 13;
    if (any7) {
      std::vector<float> hu4((size_t)36 * Cin * Cout + 2 * 9 * 256);          // (+ the slack ALG 13 reads behind the last n-tile)
      for (auto& v : hu4) v = rnd() * ws;
      POCO_HIP_CHECK(dwu4.upload(hu4));
    }
  }
  POCO_HIP_CHECK(din.upload(hin));
  POCO_HIP_CHECK(dw.upload(hw));
  POCO_HIP_CHECK(db.upload(hb));
  POCO_HIP_CHECK(hipMalloc(&dout.p, nout * sizeof(float)));
  ConvDesc d{};
  d.in = din.p; d.in_cs = Cin; d.out = dout.p; d.out_cs = Cout; d.wfrag = dw.p; d.bias = db.p;
  d.wfrag_wino = dwu.p; d.wfrag_wino4 = dwu4.p; d.wfrag_wino4p = dwu4.p; d.wfrag_wino4w = dwu4.p;     // timing only: random fragments serve all orders
  if (any11 && ks == 3 && stride == 1 && H <= 16 && W <= 16) {
    std::vector<float> hg(conv_wino4g_packed_floats(Cin, Cout));
    for (auto& v : hg) v = rnd() * ws;
    POCO_HIP_CHECK(dwu4g.upload(hg));
    d.wfrag_wino4g = dwu4g.p;
    d.scratch_floats = conv_wino4g_scratch_floats(B, H, W, Cin, Cout);
    POCO_HIP_CHECK(hipMalloc(&dscr.p, d.scratch_floats * sizeof(float)));
    d.scratch = dscr.p;
  }
  DevBuf dsk;
  bool any14 = false;
  for (int i = 0; i < ncfg; ++i) any14 = any14 || cfgs7[CONV_CFG_INTS * i + 6] == 14;
  if (any14) {                                            // stream-K 1x1 GEMM: flags (zero) + partials, the pinned error word
    static unsigned* sk_err = nullptr;
    if (!sk_err) { POCO_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&sk_err), 64, hipHostMallocMapped)); *sk_err = 0; }
    d.sk_scratch_floats = gemm1x1sk_scratch_floats();
    POCO_HIP_CHECK(hipMalloc(&dsk.p, d.sk_scratch_floats * sizeof(float)));
    POCO_HIP_CHECK(hipMemset(dsk.p, 0, (size_t)SK_MAX_WAVES * sizeof(float)));
    d.sk_scratch = dsk.p;
    d.sk_err_host = sk_err;
  }
  DevBuf dwh;
  bool any12 = false;
  for (int i = 0; i < ncfg; ++i) any12 = any12 || cfgs7[CONV_CFG_INTS * i + 6] == 12;
#if POCO_EXPERIMENTS
  if (any12 && ks == 1 && Cin % 32 == 0) {               // split-fp16 experiment: timing only, hi / lo halves of random weights
    std::vector<float> hw2((size_t)Cout * Cin), ph(gemm1x1h_packed_floats(Cin, Cout));
    for (auto& v : hw2) v = rnd() * ws;
    gemm1x1h_pack_weights(hw2.data(), nullptr, Cout, Cin, Cout, ph.data());
    POCO_HIP_CHECK(dwh.upload(ph));
    d.wfrag_h = dwh.p;
  }
#endif
  d.B = B; d.H = H; d.W = W; d.Cin = Cin; d.Cout = Cout; d.ks = ks; d.stride = stride; d.act = 1;
  if (with_res) { d.res = din.p; d.res_cs = Cin; d.res_co = 0; }      // the residual form (conv2 of a BasicBlock: `out += x`): the input doubles as the residual
  hipEvent_t e0, e1;
  POCO_HIP_CHECK(hipEventCreate(&e0));
  POCO_HIP_CHECK(hipEventCreate(&e1));
  for (int i = 0; i < ncfg; ++i) {
    const int* c = cfgs7 + CONV_CFG_INTS * i;
    ConvCfg cfg = conv_cfg_from(c);
    if (c[0] <= 0) cfg = conv_default_cfg(d);
    ms_out[i] = -1.f;
    if (conv_launch(d, cfg, stream) != POCO_OK) continue;
    conv_launch(d, cfg, stream);
    POCO_HIP_CHECK(hipEventRecord(e0, stream));
    for (int k = 0; k < iters; ++k) conv_launch(d, cfg, stream);
    POCO_HIP_CHECK(hipEventRecord(e1, stream));
    POCO_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    POCO_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms_out[i] = ms / iters;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return POCO_OK;
}

#include "kernels.h"
static int crop_common(const unsigned char* d_frame, int H, int W, const void* d_boxes, int f64, int N, double bbox_scale,
                       int res, float* d_out, void* stream) {
  if (!d_frame || !d_boxes || !d_out || N < 0 || H < 1 || W < 1 || res < 1 || H > 32767 || W > 32767) {
    poco_set_error("poco_crop_normalize: bad arguments (frame up to 32767 x 32767: cv2.warpAffine's int16 coordinate maps)");
    return POCO_ERR_ARG;
  }
  if (N == 0) return POCO_OK;
  if (f64) launch_crop_normalize_f64(d_frame, H, W, (const double*)d_boxes, bbox_scale, d_out, N, res, (hipStream_t)stream);
  else launch_crop_normalize(d_frame, H, W, (const float*)d_boxes, bbox_scale, d_out, N, res, (hipStream_t)stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { poco_set_error(std::string("poco_crop_normalize: ") + hipGetErrorString(e)); return POCO_ERR_HIP; }
  return POCO_OK;
}

extern "C" int poco_crop_normalize(const unsigned char* d_frame, int H, int W, const float* d_boxes, int N,
                                   double bbox_scale, int res, float* d_out, void* stream) {
  return crop_common(d_frame, H, W, d_boxes, 0, N, bbox_scale, res, d_out, stream);
}

extern "C" int poco_crop_normalize_multi(const unsigned char* const* d_frames, int nframes, const int* d_frame_idx, int H, int W,
                                         const float* d_boxes, int N, double bbox_scale, int res, float* d_out, void* stream) {
  if (!d_frames || !d_frame_idx || nframes < 1 || !d_boxes || !d_out || N < 0 || H < 1 || W < 1 || res < 1 || H > 32767 || W > 32767) {
    poco_set_error("poco_crop_normalize_multi: bad arguments (frames up to 32767 x 32767)");
    return POCO_ERR_ARG;
  }
  if (N == 0) return POCO_OK;
  launch_crop_normalize_multi(d_frames, nframes, d_frame_idx, H, W, d_boxes, bbox_scale, d_out, N, res, (hipStream_t)stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { poco_set_error(std::string("poco_crop_normalize_multi: ") + hipGetErrorString(e)); return POCO_ERR_HIP; }
  return POCO_OK;
}

extern "C" int poco_crop_normalize_f64(const unsigned char* d_frame, int H, int W, const double* d_boxes, int N,
                                       double bbox_scale, int res, float* d_out, void* stream) {
  return crop_common(d_frame, H, W, d_boxes, 1, N, bbox_scale, res, d_out, stream);
}

// ---- head operators on their own (parity tests against vectors made by the reference's modules) --------------------------
#include "kernels.h"

extern "C" int poco_op_part_attention(const float* d_heat, int heat_cs, const float* d_feat, int C, int B, int H, int W,
                                      float* d_out, void* stream) {
  if (!d_heat || !d_feat || !d_out) { poco_set_error("part_attention: null pointer"); return POCO_ERR_ARG; }
  if (heat_cs < 32 || (heat_cs & 15) || C < 16 || C > 128 || (C & 15) || B < 1 || H < 1 || W < 1) {
    poco_set_error("part_attention: needs heat_cs >= 32 and C <= 128, both multiples of 16 (L16 layout)");
    return POCO_ERR_ARG;
  }
  float* scratch = nullptr;
  POCO_HIP_CHECK(hipMalloc(&scratch, part_attention_scratch_floats(B, C) * sizeof(float)));
  launch_part_attention_pool_ws(d_heat, heat_cs, d_feat, C, d_out, C * 24, B, H, W, scratch, (hipStream_t)stream);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  (void)hipFree(scratch);
  POCO_HIP_CHECK(e);
  return POCO_OK;
}

// `iters` back-to-back launches of the pool between two HIP events on `stream` (scratch allocated once, outside the events).
extern "C" int poco_bench_part_attention(const float* d_heat, int heat_cs, const float* d_feat, int C, int B, int H, int W,
                                         float* d_out, int iters, float* ms_out, void* stream) {
  if (!d_heat || !d_feat || !d_out || !ms_out || iters < 1) { poco_set_error("bench_part_attention: bad argument"); return POCO_ERR_ARG; }
  if (heat_cs < 32 || (heat_cs & 15) || C < 16 || C > 128 || (C & 15) || B < 1 || H < 1 || W < 1) {
    poco_set_error("bench_part_attention: needs heat_cs >= 32 and C <= 128, both multiples of 16 (L16 layout)");
    return POCO_ERR_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  float* scratch = nullptr;
  POCO_HIP_CHECK(hipMalloc(&scratch, part_attention_scratch_floats(B, C) * sizeof(float)));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  launch_part_attention_pool_ws(d_heat, heat_cs, d_feat, C, d_out, C * 24, B, H, W, scratch, s);
  (void)hipEventRecord(e0, s);
  for (int i = 0; i < iters; ++i) launch_part_attention_pool_ws(d_heat, heat_cs, d_feat, C, d_out, C * 24, B, H, W, scratch, s);
  (void)hipEventRecord(e1, s);
  hipError_t e = hipEventSynchronize(e1);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(scratch);
  POCO_HIP_CHECK(e);
  *ms_out = ms / iters;
  return POCO_OK;
}

extern "C" int poco_op_lc2d_pose(const float* d_x, const float* d_w, float* d_pose6d, int B, void* stream) {
  if (!d_x || !d_w || !d_pose6d || B < 1) { poco_set_error("lc2d_pose: bad argument"); return POCO_ERR_ARG; }
  launch_lc2d_pose(d_x, 128 * 24, d_w, d_pose6d, B, (hipStream_t)stream);
  POCO_HIP_CHECK(hipGetLastError());
  return POCO_OK;
}

extern "C" int poco_op_rot6d(const float* d_in, float* d_rotmat, int B, void* stream) {
  if (!d_in || !d_rotmat || B < 1) { poco_set_error("rot6d: bad argument"); return POCO_ERR_ARG; }
  launch_rot6d(d_in, 144, d_rotmat, 216, nullptr, 0, B, (hipStream_t)stream);
  POCO_HIP_CHECK(hipGetLastError());
  return POCO_OK;
}
