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
 *   - activations handed to the stand-alone operators are NHWC fp32 unless stated otherwise.
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

/* ---- stand-alone operators (parity tests, tuner, micro-benchmarks) -------------------------- */

/* conv(ks x ks, stride, pad=(ks-1)/2, no groups/dilation) * scale[co] + shift[co] (+ residual) (ReLU)
 * == nn.Conv2d -> nn.BatchNorm2d(eval) [-> "+= residual"] [-> ReLU] of
 * pocolib/models/backbone/hrnet.py:42-58,79-99 / hrnet_cls.py / resnet.py:101-121.
 * d_in  [B,H,W,Cin] NHWC, h_weight [Cout,Cin,ks,ks] (host, torch OIHW order),
 * h_scale/h_shift [Cout] (host, nullable), d_res [B,Ho,Wo,Cout] NHWC (nullable), d_out NHWC.
 * Cin and Cout must be multiples of 16.  cfg6 = {MT,NT,WM,WN,R,NI} or NULL for the heuristic. */
int poco_op_conv2d(const float* d_in, int B, int H, int W, int Cin, const float* h_weight,
                   const float* h_scale, const float* h_shift, int Cout, int ks, int stride,
                   const float* d_res, int relu, float* d_out, const int* cfg6, void* stream);

/* Time `iters` launches of the same conv (+ReLU) with hipEvents; ms_out = mean ms per launch.
 * cfg_used6 (nullable) receives the tile configuration that ran. */
int poco_bench_conv2d(const float* d_in, int B, int H, int W, int Cin, const float* h_weight, int Cout,
                      int ks, int stride, float* d_out, const int* cfg6, int iters, float* ms_out,
                      int* cfg_used6, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* POCO_HIP_H */
