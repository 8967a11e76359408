// Which SIMD / wave slot / CU / XCD do the waves of a launch land on?  hipcc -O3 --offload-arch=gfx950 tools/probes/hwid.hip -o tools/probes/bin/hwid
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(64) probe(unsigned* out, int spin) {
  extern __shared__ float lds[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(10);
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = xcc; }
  if (spin < 0) lds[threadIdx.x] = 0;
}
int main() {
  const int nb = 4096;
  unsigned* out; (void)hipMalloc(&out, nb * 8);
  for (int ldsk : {80, 40, 16}) {     // 80 KiB -> 1-2 blocks per CU ... 16 KiB -> up to 10
    (void)hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, ldsk * 1024);
    probe<<<nb, 64, ldsk * 1024>>>(out, 20000);
    std::vector<unsigned> h(nb * 2);
    (void)hipMemcpy(h.data(), out, nb * 8, hipMemcpyDeviceToHost);
    printf("LDS %d KiB per block: block -> xcc se cu simd slot\n", ldsk);
    for (int b = 0; b < 48; ++b) {
      const unsigned hw = h[b * 2];
      printf("  %4d: xcc %u se %u cu %2u simd %u slot %u   (hw %08x)\n", b, h[b * 2 + 1] & 15, (hw >> 13) & 7, (hw >> 8) & 15, (hw >> 4) & 3, hw & 15, hw);
    }
    int hist[16] = {0};
    for (int b = 0; b < nb; ++b) hist[h[b * 2] & 15]++;
    printf("  slot histogram:"); for (int i = 0; i < 16; ++i) printf(" %d", hist[i]); printf("\n");
  }
  return 0;
}
