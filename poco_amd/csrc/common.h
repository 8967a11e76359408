// Shared declarations for the POCO MI355X (gfx950) HIP library.
// Everything here is internal; the public surface is include/poco_hip.h.
#pragma once
#ifndef POCO_PROBES
#define POCO_PROBES 0     // 1: timing-probe builds (tools/build_exp.sh): the POCO_CONV_DBG / POCO_CONV_REPEAT environment switches exist; never in the shipped library
#endif
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define POCO_HIP_CHECK(expr)                                                        \
  do {                                                                              \
    hipError_t _e = (expr);                                                         \
    if (_e != hipSuccess) {                                                         \
      poco_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));            \
      return POCO_ERR_HIP;                                                          \
    }                                                                               \
  } while (0)

enum { POCO_OK = 0, POCO_ERR_ARG = 1, POCO_ERR_HIP = 2, POCO_ERR_STATE = 3, POCO_ERR_MISSING = 4,
       POCO_ERR_SHAPE = 5 };

void poco_set_error(const std::string& msg);
// Compute units of the current device (hipDeviceAttributeMultiprocessorCount; 256 on MI355X), cached per thread and device.
int poco_num_cus();

// ---------------------------------------------------------------------------------------------
// Convolution (implicit GEMM on v_mfma_f32_16x16x4_f32), NHWC activations.
// ---------------------------------------------------------------------------------------------

// Tile configuration for one conv launch.  MT/NT select the template instantiation (register
// tile of 16x16 MFMA blocks per wave: MT along pixels, NT along output channels); the rest are
// runtime parameters of the block decomposition.
struct ConvCfg {
  int MT;      // 16-pixel sub-tiles per wave            (4, 7 or 13)
  int NT;      // 16-channel sub-tiles per wave          (1..4)
  int WM;      // waves along pixels
  int WN;      // waves along output channels  (block = WM*WN waves)
  int R;       // output rows per slab
  int NI;      // slabs (row bands / whole images) per block
  int ALG;     // 0: register-staged single LDS buffer; 1: LDS-DMA double-buffered (patch + weights);
               // 2: ALG 1 persistent over tiles; 3: Winograd F(2x2,3x3) (MT ignored, NT in {1,2}, R even);
               // 4: Winograd, half-position waves + pipelined transform (WN = 2 halves, WM <= 4, NT <= 3)
               // 5: small-M linear (H = W = 1, ks = 1): K split over WM waves per 16 outputs (linear_mfma.hip)
               // 7: Winograd F(4x4,3x3) for planes >= 28x28 (conv_wino4.hip): NT 1..3, WM = 2 tile
               //    groups, WN = 4 position quarters, R = output rows per slab (multiple of 4), NI slabs (<= 32 tiles)
               // 8: ALG 7's arithmetic and geometry with specialised waves (conv_wino4p.hip): 8 MFMA waves + 4 producer
               //    waves (LDS-DMA + input transform, V staged in LDS); same cfg fields as ALG 7
               // 13: F(4x4,3x3) with WHOLE-POSITION MFMA waves (conv_wino4w.hip; round 5): a block = 2 NT MFMA waves (all 36 positions of
               //    one 16-tile group x one n-tile each, register-only output transform: no exchange) + 2 producer waves, slice pipeline
               //    continuous across items; NT 1..3, WM = 2, WN = 1, R / NI as ALG 8 (flat items: R = 4, NI = 0)
               // 6: 1x1 conv (stride 1|2) as a register-direct GEMM, no LDS / barriers (gemm1x1.hip):
               //    (MT,NT) in {(2,4),(4,2),(4,4),(7,2),(7,4),(8,2)}, R = operand prefetch depth (2|3), NI = load schedule 1..6
               // 9: ALG 6 with coalesced global traffic: pixel / output tiles turned into the MFMA lane order through
               //    wave-private LDS (gemm1x1t.hip): (MT,NT) in {(4,4),(7,2),(7,4),(8,2)}, R = NI = 1
               // 12: EXPERIMENT, -DPOCO_EXPERIMENTS=1 builds only (python -m poco_amd.build --experiments), never chosen by the table:
               //     1x1 conv in split fp16 (exp/gemm1x1h.hip)
               // 11: Winograd F(4x4,3x3) as 36 position GEMMs with V / M staged in memory, for planes <= 16x16 (conv_wino4g.hip):
               //    three launches (input transform, GEMM, output transform); (MT,NT) in {(2,4),(4,2),(4,4),(8,2)}, R = depth 2|3
               // 10: 3x3 conv (stride 1|2) as a register-direct gather GEMM over K = 9*Cin, no LDS / barriers (gemm3x3.hip):
               //    (MT,NT) in {(2,4),(4,2..4),(7,2..4),(8,2)}, R = operand prefetch depth (2|3), NI = load schedule 1|3|6
};
constexpr int CONV_CFG_INTS = 7;   // ints per configuration in the C ABI / tuning table
inline ConvCfg conv_cfg_from(const int* c) { return ConvCfg{c[0], c[1], c[2], c[3], c[4], c[5], c[6]}; }

// Activation layout ("L16", channel-slice-major NHWC): element (b, y, x, c) of a buffer with C channels lives at
//   ((b*H + y) * (C/16) + c/16) * (W*16) + x*16 + c%16
// i.e. a 16-channel slice of an image row is one contiguous run of W*64 bytes, so the per-slice halo-patch
// fetches of the conv kernels read whole cache lines.  For vectors (H = W = 1) this is plain [B][C].
// Channel offset `co` of a slice inside a wider buffer -> float offset (co/16)*W*16 + co%16.
inline size_t l16_chan_off(int co, int W) { return (size_t)(co >> 4) * W * 16 + (co & 15); }

struct ConvDesc {
  // activations: L16 (see above), each buffer may be a channel slice of a wider buffer
  const float* in;  int in_cs,  in_co;    // channel stride (channels per pixel of the buffer), offset
  const float* res; int res_cs, res_co;   // optional residual (same spatial shape as the output)
  float*       out; int out_cs, out_co;
  const float* wfrag;                     // weights in MFMA fragment order (see conv_pack_weights)
  const float* wfrag_wino;                // 3x3 stride-1 only: Winograd-transformed weights (ALG 3), nullable
  const float* wfrag_wino4 = nullptr;     // 3x3 stride-1 only: F(4x4,3x3) weight fragments, 36 positions (ALG 7), nullable
  const float* wfrag_wino4p = nullptr;    // the same weights in the LDS order of ALG 8 (conv_wino4p.hip), nullable
  const float* wfrag_wino4w = nullptr;    // the same weights in the LDS order of ALG 13 (conv_wino4w.hip), nullable
  const float* wfrag_wino4g = nullptr;    // 3x3 stride-1 convs on planes <= 8x8: per-position GEMM fragments of ALG 11 (conv_wino4g.hip)
  const float* wfrag_h = nullptr;         // EXPERIMENT (ALG 12, gemm1x1h.hip): hi / lo fp16 halves of the 1x1 weights, nullable
  float* scratch = nullptr;               // ALG 11: V + M staging (conv_wino4g_scratch_floats), owned by the caller
  size_t scratch_floats = 0;
  float* sk_scratch = nullptr;            // ALG 14 (gemm1x1sk.hip): flags + partial accumulators (gemm1x1sk_scratch_floats), zeroed once by the owner
  size_t sk_scratch_floats = 0;
  unsigned* sk_err_host = nullptr;        // ... and the pinned host word its bounded waits raise
  unsigned sk_max_spins = 0;              // poll bound of those waits (0 = 2^21)
  // ALG 11 chaining (engine only): the previous conv already left this conv's V in scratch half `wg_vsel`; this conv leaves the
  // next conv's V in the other half (wg_mid_kernel) and writes its own output tensor only if somebody else still reads it
  int wg_skip_in = 0, wg_vsel = 0, wg_emit_next = 0, wg_store_y = 1;
  const float* bias;                      // [Cout_padded] folded BN shift / conv bias
  int B, H, W, Cin, Cout;                 // Cout = padded to a multiple of 16
  int ks, stride;                         // ks in {1,3}; pad = (ks-1)/2; stride in {1,2}
  int act;             // 0 none, 1 ReLU, 2 sigmoid, 3 ReLU on output channels >= relu_from only
  int res_after_act;   // residual added after the activation instead of before
  int relu_from;       // act == 3: first output channel (multiple of 16) that gets the ReLU (merged convs)
};

// Size (floats) of the packed weight buffer for a conv.
size_t conv_packed_weight_floats(int Cin, int Cout16, int ks);
// Pack OIHW weights (host) * per-output-channel scale into fragment order (host buffer).
// Cout16 >= Cout is Cout rounded up to a multiple of 16 (extra channels are zero).
void conv_pack_weights(const float* w_oihw, const float* scale /*nullable*/, int Cout, int Cin,
                       int ks, int Cout16, float* dst);
// Heuristic tile choice.
ConvCfg conv_default_cfg(const ConvDesc& d);
// Validate + launch.  Returns POCO_OK or an error code (message via poco_set_error).
int conv_launch(const ConvDesc& d, const ConvCfg& cfg, hipStream_t stream);
// LDS bytes a configuration needs (0 if invalid).
size_t conv_lds_bytes(const ConvDesc& d, const ConvCfg& cfg);

// ---- small-M linear layers (linear_mfma.hip), ALG 5 -------------------------------------------------
bool linear_cfg_valid(const ConvDesc& d, const ConvCfg& cfg);
int linear_launch(const ConvDesc& d, const ConvCfg& cfg, hipStream_t stream);

// ---- 1x1 convs as a register-direct GEMM (gemm1x1.hip), ALG 6 ---------------------------------------
bool gemm1x1_cfg_valid(const ConvDesc& d, const ConvCfg& cfg);
int gemm1x1_launch(const ConvDesc& d, const ConvCfg& cfg, hipStream_t stream);
// ---- the same with coalesced global traffic and an LDS transposition (gemm1x1t.hip), ALG 9 -----------------
bool gemm1x1t_cfg_valid(const ConvDesc& d, const ConvCfg& cfg);
// ---- stream-K 1x1 GEMM (gemm1x1sk.hip), ALG 14 ---------------------------------------------------------
constexpr int SK_MAX_WAVES = 2048;                 // flags at the head of the scratch buffer
constexpr size_t SK_PART_FLOATS = (size_t)32768 * 256;   // partial accumulators: waves x tiles per wave tile x 256 floats
size_t gemm1x1sk_scratch_floats();
bool gemm1x1sk_cfg_valid(const ConvDesc& d, const ConvCfg& cfg);
int gemm1x1sk_launch(const ConvDesc& d, const ConvCfg& cfg, hipStream_t stream);
size_t gemm1x1t_lds_bytes(const ConvDesc& d, const ConvCfg& cfg);
int gemm1x1t_launch(const ConvDesc& d, const ConvCfg& cfg, hipStream_t stream);

#ifndef POCO_EXPERIMENTS
#define POCO_EXPERIMENTS 0      // 1: also build the labelled experiments (csrc/exp/*.hip, 3-deep rings of ALG 4); not in the shipped library
#endif
#if POCO_EXPERIMENTS
// ---- EXPERIMENT: 1x1 convs in split fp16 (hi + lo, 3 MFMAs of 16x16x32_f16 per product; exp/gemm1x1h.hip), ALG 12 -----------
size_t gemm1x1h_packed_floats(int Cin, int Cout16);
void gemm1x1h_pack_weights(const float* w_oi, const float* scale, int Cout, int Cin, int Cout16, float* dst);
bool gemm1x1h_cfg_valid(const ConvDesc& d, const ConvCfg& cfg);
int gemm1x1h_launch(const ConvDesc& d, const ConvCfg& cfg, hipStream_t stream);
#endif

// ---- Winograd F(4x4,3x3) as a position-batched GEMM for small planes (conv_wino4g.hip), ALG 11 --------------
size_t conv_wino4g_packed_floats(int Cin, int Cout16);
void conv_wino4g_pack_weights(const float* w_oihw, const float* scale, int Cout, int Cin, int Cout16, float* dst);
size_t conv_wino4g_scratch_floats(int B, int H, int W, int Cin, int Cout);
bool conv_wino4g_can_chain(int H, int W, int Cout, int next_Cin);
bool conv_wino4g_cfg_valid(const ConvDesc& d, const ConvCfg& cfg);
int conv_wino4g_launch(const ConvDesc& d, const ConvCfg& cfg, hipStream_t stream);

// ---- 3x3 convs as a register-direct gather GEMM (gemm3x3.hip), ALG 10 -------------------------------------
bool gemm3x3_cfg_valid(const ConvDesc& d, const ConvCfg& cfg);
int gemm3x3_launch(const ConvDesc& d, const ConvCfg& cfg, hipStream_t stream);

#include <vector>
// ---- Winograd F(4x4,3x3) (conv_wino4.hip), ALG 7 -------------------------------------------
size_t conv_wino4_packed_floats(int Cin, int Cout16);
void conv_wino4_pack_weights(const float* w_oihw, const float* scale, int Cout, int Cin, int Cout16, float* dst);
size_t conv_wino4_lds_bytes(const ConvDesc& d, const ConvCfg& cfg);
int conv_wino4_launch(const ConvDesc& d, const ConvCfg& cfg, hipStream_t stream);

// ---- Winograd F(4x4,3x3) with specialised waves (conv_wino4p.hip), ALG 8 ---------------------
size_t conv_wino4p_packed_floats(int Cin, int Cout16);
void conv_wino4p_pack_weights(const float* w_oihw, const float* scale, int Cout, int Cin, int Cout16, float* dst);
size_t conv_wino4p_lds_bytes(const ConvDesc& d, const ConvCfg& cfg);
int conv_wino4p_launch(const ConvDesc& d, const ConvCfg& cfg, hipStream_t stream);

// ---- Winograd F(4x4,3x3) with whole-position MFMA waves (conv_wino4w.hip), ALG 13 ---------------------
size_t conv_wino4w_packed_floats(int Cin, int Cout16);
void conv_wino4w_pack_weights(const float* w_oihw, const float* scale, int Cout, int Cin, int Cout16, float* dst);
size_t conv_wino4w_lds_bytes(const ConvDesc& d, const ConvCfg& cfg);
int conv_wino4w_launch(const ConvDesc& d, const ConvCfg& cfg, hipStream_t stream);

// ---- Winograd F(2x2,3x3) variant (conv_wino.hip) --------------------------------------------------
#include <vector>
// [Cout][Cin][16] transformed filters (G g G^T, float64 on the host); pack with conv_pack_weights(ks=4).
void conv_wino_transform_weights(const float* w_oihw, int Cout, int Cin, std::vector<float>* out);
size_t conv_wino_lds_bytes(const ConvDesc& d, const ConvCfg& cfg);
int conv_wino_launch(const ConvDesc& d, const ConvCfg& cfg, hipStream_t stream);
