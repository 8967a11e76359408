// micro-benchmark of grid barriers on gfx950: hipcc --offload-arch=gfx950 -O3 tools/probe/gridbar.hip -o tools/probe/gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ unsigned ld_agent(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// MODE 0: flat counter + __threadfence both sides; 1: flat counter, no fences (s_waitcnt only); 2: per-XCD counters (blockIdx % 8), last arriver of
// an XCD bumps the global one; 3: as 1 but every block polls with all its lanes idle (thread 0 only) and s_sleep 0
template <int MODE>
__global__ void __launch_bounds__(512) bar_kernel(unsigned* sync, int nbar, float* data) {
  unsigned target = 0, xt = 0;
  const unsigned G = gridDim.x;
  const unsigned xcd = blockIdx.x & 7, nx = (G - xcd + 7) / 8;   // blocks on this XCD (round-robin placement)
  for (int i = 0; i < nbar; ++i) {
    data[blockIdx.x * 512 + threadIdx.x] += 1.f;
    if (MODE != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      if (MODE == 0) __threadfence();
      if (MODE == 2) {
        xt += nx; target += 1;
        const unsigned old = __hip_atomic_fetch_add(&sync[16 + 16 * xcd], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == xt) __hip_atomic_fetch_add(&sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned want = (target) * min(G, 8u);
        while (ld_agent(&sync[0]) < want) __builtin_amdgcn_s_sleep(1);
      } else {
        target += G;
        __hip_atomic_fetch_add(&sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (ld_agent(&sync[0]) < target) __builtin_amdgcn_s_sleep(MODE == 3 ? 0 : 1);
      }
      if (MODE == 0) __threadfence();
      if (MODE == 4) asm volatile("buffer_inv sc1" ::: "memory");
    }
    __syncthreads();
  }
}
template <int MODE> float run(int G, int nbar, unsigned* sync, float* data) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e9;
  for (int rep = 0; rep < 5; ++rep) {
    hipMemset(sync, 0, 4096);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(bar_kernel<MODE>, dim3(G), dim3(512), 0, 0, sync, nbar, data);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
  }
  return best;
}
int main() {
  unsigned* sync; float* data;
  hipMalloc(&sync, 4096); hipMalloc(&data, 256 * 512 * 4); hipMemset(data, 0, 256 * 512 * 4);
  for (int G : {8, 32, 64, 128, 256}) {
    const int n1 = 20, n2 = 220;
    float t[5][2];
    t[4][0] = run<4>(G, n1, sync, data); t[4][1] = run<4>(G, n2, sync, data);
    t[0][0] = run<0>(G, n1, sync, data); t[0][1] = run<0>(G, n2, sync, data);
    t[1][0] = run<1>(G, n1, sync, data); t[1][1] = run<1>(G, n2, sync, data);
    t[2][0] = run<2>(G, n1, sync, data); t[2][1] = run<2>(G, n2, sync, data);
    t[3][0] = run<3>(G, n1, sync, data); t[3][1] = run<3>(G, n2, sync, data);
    printf("G=%3d  us/barrier: fence %.2f  nofence %.2f  per-xcd %.2f  nofence-nosleep %.2f  inv-only %.2f\n", G,
           (t[0][1] - t[0][0]) * 1e3 / (n2 - n1), (t[1][1] - t[1][0]) * 1e3 / (n2 - n1), (t[2][1] - t[2][0]) * 1e3 / (n2 - n1), (t[3][1] - t[3][0]) * 1e3 / (n2 - n1), (t[4][1] - t[4][0]) * 1e3 / (n2 - n1));
  }
  return 0;
}
