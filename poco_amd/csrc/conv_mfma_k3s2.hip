// Kernel instances of the direct conv for ks = 3, stride = 2 (see conv_mfma_kernels.h).
#include "conv_mfma_kernels.h"

int conv_launch_k3s2(int alg, int MT, int NT, const ConvKParams& kp, dim3 grid, int nthreads, size_t lds, hipStream_t stream) {
  return launch_mtnt<3, 2>(alg, MT, NT, kp, grid, nthreads, lds, stream);
}
