// The CLIFF regressor as ONE persistent launch (round 5).
//
// cliff_head.py:96-118 after the pooled feature is a chain of ten dependent small-M GEMMs (fc1's feature part once, then three
// times fc1's state part -> fc2 -> the stacked decoders) plus the row copies that scatter the state into the outputs and
// rot6d_to_rotmat (utils/geometry.py:247-261).  With 1 ... 128 crops every link is 0.01-0.3 GFLOP against 0.6-8 MB of weights:
// as separate launches (ALG 5, linear_mfma.hip) the chain costs a dispatch + two dependent memory round trips + a drain per
// link - 130 us of the 64-crop forward for 0.4 GFLOP and 24 MB, and the same 130 us of the one-crop forward.
//
// Here the chain is a PROGRAM of stages executed by one grid of co-resident blocks with a grid barrier between stages:
//   * a stage is a set of independent tile jobs: (layer, 16 output features, 16 rows) of a Linear layer, or 16 rows of a row
//     job (copy / broadcast / rot6d); the blocks take the jobs of a stage round-robin;
//   * a Linear tile job is what one block of linear_mfma_kernel does: the K slices are dealt round-robin to the block's 8 waves,
//     every wave has the operand loads of (up to) 8 slices in flight before its first MFMA, the partial sums meet in LDS and are
//     added in a fixed order (wave 0 ... 7): results do not depend on the grid size or on which block ran the job;
//   * the grid barrier is a monotonic arrival counter in device memory.  The L2s of the eight XCDs are not coherent with each
//     other for plain loads and stores, and the textbook protocol (cooperative-groups grid.sync(): an agent-scope release
//     fence before the arrival, an acquire fence after the wait) makes every block write back and invalidate its XCD's whole
//     L2: measured 9 us per barrier at 128 blocks, 16 us at 256 (tools/probe/gridbar.hip) - the first version of this kernel
//     took 188 us for its 11 stages, longer than the launches it replaced.  Instead, every activation a stage exchanges with
//     the next one is written and read with AGENT-SCOPE accesses (relaxed 64-bit atomics = global_load / store ... sc1: write
//     through to, and always fetched from, the memory-side coherence point), the weights (never written) with plain loads, and
//     the barrier itself needs no cache maintenance: s_waitcnt vmcnt(0) (the stores are acknowledged), block barrier, one
//     arrival per block, poll: 2.1 us at 128 blocks, 1.4 us at 64.  The launch is an
//     ordinary one (capturable in the forward's hipGraph); it is deadlock-free because the grid is at most one small block per
//     CU (8 KiB of LDS, <= 128 VGPRs): every block becomes resident as soon as a CU has 8 free wave slots, and kernels of other
//     lanes never wait for this one.  The wait is bounded: a block that does not see the others within ~3 s of spinning raises
//     the error word (pinned host memory, read by poco_forward) and leaves (the forward then fails loudly instead of hanging the queue).
//   * the last block to leave resets the counter, so a replayed graph needs no memset node.
#include "kernels.h"
#include <algorithm>
#include <string>

namespace {

constexpr int MLP_WAVES = 8;
constexpr int MLP_UNROLL = 8;   // K slices a wave requests together

__device__ __forceinline__ unsigned ld_agent(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// agent-scope (sc1) accesses to the data the stages exchange: coherent across the XCDs without cache maintenance
__device__ __forceinline__ float4 ld4_dev(const float* p) {            // 16-byte aligned
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
  const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_float4(__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)),
                     __uint_as_float((unsigned)hi), __uint_as_float((unsigned)(hi >> 32)));
}
__device__ __forceinline__ void st4_dev(float* p, float4 v) {
  unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
  __hip_atomic_store(q, (unsigned long long)__float_as_uint(v.x) | ((unsigned long long)__float_as_uint(v.y) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(q + 1, (unsigned long long)__float_as_uint(v.z) | ((unsigned long long)__float_as_uint(v.w) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld1_dev(const float* p) {
  return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st1_dev(float* p, float v) {
  __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// sync[0] = groups that arrived (monotonic within a launch), sync[16 + 16 x] = arrivals of group x, sync[1] = blocks that left, sync[2] = "a block gave up" (this launch);
// err_host = the sticky error word the host reads before the next forward
__device__ __forceinline__ void rot6d_one(const float* x, float* R) {   // utils/geometry.py:247-261, same operation order as rot6d_kernel
  const float a1[3] = {x[0], x[2], x[4]}, a2[3] = {x[1], x[3], x[5]};
  const float n1 = fmaxf(sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]), 1e-12f);
  const float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
  const float d = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
  const float u[3] = {a2[0] - d * b1[0], a2[1] - d * b1[1], a2[2] - d * b1[2]};
  const float n2 = fmaxf(sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), 1e-12f);
  const float b2[3] = {u[0] / n2, u[1] / n2, u[2] / n2};
  const float b3[3] = {b1[1] * b2[2] - b1[2] * b2[1], b1[2] * b2[0] - b1[0] * b2[2], b1[0] * b2[1] - b1[1] * b2[0]};
  for (int r = 0; r < 3; ++r) { R[r * 3 + 0] = b1[r]; R[r * 3 + 1] = b2[r]; R[r * 3 + 2] = b3[r]; }
}

// job j of stage st -> (layer, n-tile, m-tile); the Linear jobs of a stage's layers are numbered consecutively, m-tile innermost
// (neighbouring blocks share a weight fragment in the L2)
__device__ __forceinline__ bool decode_job(const MlpProgram& p, const MlpStage& st, int j, int mT, int* li, int* nt, int* mt) {
  for (int l = st.layer0; l < st.layer0 + st.nlayers; ++l) {
    const int n = p.layer[l].nT16 * mT;
    if (j < n) { *li = l; *nt = j / mT; *mt = j - (j / mT) * mT; return true; }
    j -= n;
  }
  return false;
}

__global__ void __launch_bounds__(MLP_WAVES * 64)
mlp_chain_kernel(const MlpProgram p) {
  __shared__ float4 red[MLP_WAVES * 64];
  __shared__ int bar_ok;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int idx = lane & 15, g = lane >> 4;
  const int mT = (p.B + 15) >> 4;
  const int G = (int)gridDim.x;
  // two-level arrival: the blocks with the same blockIdx & 7 (the same XCD under the round-robin placement; only the cost depends
  // on that) count on their own word, the last of a group bumps the global one: 2.9 instead of 3.9 us at 256 blocks
  const unsigned grp = blockIdx.x & 7u, ngrp = min((unsigned)G, 8u), ngrp_blocks = ((unsigned)G - grp + 7u) / 8u;
  unsigned target = 0, grp_target = 0;
  const unsigned max_spins = p.max_spins ? p.max_spins : (1u << 21);
  int nt_trace = 0;
  auto stamp = [&]() { if (p.trace && blockIdx.x == 0 && threadIdx.x == 0) p.trace[nt_trace++] = wall_clock64(); };
  stamp();
  // The weights of a stage do not depend on the stage before it: the fragments of this block's FIRST job of the next stage (this
  // wave's first MLP_UNROLL slices) are requested after the block has arrived at the barrier and travel while it waits.
  float4 apf[MLP_UNROLL];
#pragma unroll
  for (int u = 0; u < MLP_UNROLL; ++u) apf[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  auto prefetch = [&](int s) {
    int li, nt, mt;
    if (s < p.nstages && decode_job(p, p.stage[s], (int)blockIdx.x, mT, &li, &nt, &mt)) {
      const MlpLayer& L = p.layer[li];
      const float4* wl = L.wfrag + (size_t)nt * 64 + lane;
      const size_t wstride = (size_t)L.nT16 * 64;
#pragma unroll
      for (int u = 0; u < MLP_UNROLL; ++u) apf[u] = wl[(size_t)min(wave + u * MLP_WAVES, L.nC16 - 1) * wstride];
    }
  };
  prefetch(0);
  for (int s = 0; s < p.nstages; ++s) {
    const MlpStage st = p.stage[s];
    // ---- Linear tile jobs ----
    for (int k = 0;; ++k) {
      int li, nt, mt;
      if (!decode_job(p, st, (int)blockIdx.x + k * G, mT, &li, &nt, &mt)) break;
      const MlpLayer& L = p.layer[li];
      const int row = min(mt * 16 + idx, p.B - 1);                        // dead rows re-read the last one
      const float* xrow = L.in + (size_t)row * L.in_rs + 4 * g;
      const float4* wl = L.wfrag + (size_t)nt * 64 + lane;
      const size_t wstride = (size_t)L.nT16 * 64;                         // float4 per K slice
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int c0 = wave; c0 < L.nC16; c0 += MLP_WAVES * MLP_UNROLL) {
        float4 a[MLP_UNROLL], b[MLP_UNROLL];
#pragma unroll
        for (int u = 0; u < MLP_UNROLL; ++u)
          b[u] = ld4_dev(xrow + (size_t)min(c0 + u * MLP_WAVES, L.nC16 - 1) * 16);   // clamped duplicates are masked below
        if (k == 0 && c0 == wave) {                                       // wave-uniform
#pragma unroll
          for (int u = 0; u < MLP_UNROLL; ++u) a[u] = apf[u];
        } else {
#pragma unroll
          for (int u = 0; u < MLP_UNROLL; ++u) a[u] = wl[(size_t)min(c0 + u * MLP_WAVES, L.nC16 - 1) * wstride];
        }
#pragma unroll
        for (int u = 0; u < MLP_UNROLL; ++u) {
          if (c0 + u * MLP_WAVES < L.nC16) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, b[u].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, b[u].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, b[u].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, b[u].w, acc, 0, 0, 0);
          }
        }
      }
      red[wave * 64 + lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
      __syncthreads();
      if (wave == 0) {
        float4 sum = red[lane];
#pragma unroll
        for (int w = 1; w < MLP_WAVES; ++w) {
          const float4 t = red[w * 64 + lane];
          sum.x += t.x; sum.y += t.y; sum.z += t.z; sum.w += t.w;
        }
        const int r = mt * 16 + idx;
        if (r < p.B) {
          const int co = nt * 16 + g * 4;
          const float4 sh = *reinterpret_cast<const float4*>(L.bias + co);
          float v[4] = {sum.x + sh.x, sum.y + sh.y, sum.z + sh.z, sum.w + sh.w};
          if (L.res) {
            const float4 rr = ld4_dev(L.res + (size_t)r * L.res_rs + co);
            v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
          }
          if (L.act == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          } else if (L.act == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = 1.f / (1.f + __expf(-v[e]));
          }
          st4_dev(L.out + (size_t)r * L.out_rs + co, make_float4(v[0], v[1], v[2], v[3]));
        }
      }
      __syncthreads();                                                    // `red` is reused by the block's next job
    }
    // ---- row jobs: 16 rows each, dealt from the LAST block downwards (the Linear jobs fill the grid from block 0) ----
    for (int j = G - 1 - (int)blockIdx.x; j < st.nrows * mT; j += G) {
      const MlpRowJob& J = p.row[st.row0 + j / mT];
      const int r0 = (j % mT) * 16, nr = min(p.B, r0 + 16) - r0;
      if (J.kind == MLP_ROW_ROT6D) {
        for (int i = threadIdx.x; i < nr * 24; i += blockDim.x) {
          const int r = r0 + i / 24, q = i % 24;
          float x6[6], R[9];
          for (int e = 0; e < 6; ++e) x6[e] = ld1_dev(J.src + (size_t)r * J.src_rs + q * 6 + e);
          rot6d_one(x6, R);
          if (J.dst) for (int e = 0; e < 9; ++e) st1_dev(J.dst + (size_t)r * J.dst_rs + q * 9 + e, R[e]);
          if (J.dst2) for (int e = 0; e < 9; ++e) J.dst2[(size_t)r * J.dst2_rs + q * 9 + e] = R[e];
        }
      } else if (J.kind == MLP_ROW_COPY && !((J.n | J.src_rs | J.dst_rs) & 3) && !(((size_t)J.src | (size_t)J.dst) & 15)) {
        const int n4 = J.n >> 2;
        for (int i = threadIdx.x; i < nr * n4; i += blockDim.x) {
          const int r = r0 + i / n4, c = (i - (i / n4) * n4) * 4;
          st4_dev(J.dst + (size_t)r * J.dst_rs + c, ld4_dev(J.src + (size_t)r * J.src_rs + c));
        }
      } else {
        const int n = J.n;
        for (int i = threadIdx.x; i < nr * n; i += blockDim.x) {
          const int r = r0 + i / n, c = i - (i / n) * n;
          st1_dev(J.dst + (size_t)r * J.dst_rs + c, ld1_dev(J.src + (J.kind == MLP_ROW_BCAST ? 0 : (size_t)r * J.src_rs) + c));
        }
      }
    }
    stamp();
    if (s + 1 == p.nstages) break;
    // ---- grid barrier (no cache maintenance: see the header) with the next stage's weight prefetch inside ----
    target += ngrp; grp_target += ngrp_blocks;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's (write-through) stores of the stage are acknowledged
    __syncthreads();
    if (threadIdx.x == 0 && !(p.debug_skip_arrival && s == 0 && blockIdx.x == 0)) {
      const unsigned old = __hip_atomic_fetch_add(&p.sync[16 + 16 * grp], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old + 1 == grp_target) __hip_atomic_fetch_add(&p.sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    prefetch(s + 1);
    if (threadIdx.x == 0) {
      int good = 1;
      unsigned spins = 0;
      while (ld_agent(&p.sync[0]) < target) {
        __builtin_amdgcn_s_sleep(1);
        // bounded: a time-out (or another block's: checked every 256 polls, so that one late block does not cost every stage its own
        // full wait) ends the launch with the error word raised instead of a hung queue
        if (++spins > max_spins || ((spins & 255u) == 0u && ld_agent(&p.sync[2]))) { good = 0; break; }
      }
      if (!good) {
        __hip_atomic_store(&p.sync[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p.err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      } else if (ld_agent(&p.sync[2])) good = 0;         // another block gave up: leave too
      bar_ok = good;
    }
    __syncthreads();
    stamp();
    if (!bar_ok) break;
  }
  // the last block to leave re-arms the counters for the next launch (every block has passed the last barrier by then)
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned left = __hip_atomic_fetch_add(&p.sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (left + 1 == gridDim.x) {
      __hip_atomic_store(&p.sync[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&p.sync[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&p.sync[2], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (unsigned x = 0; x < 8; ++x) __hip_atomic_store(&p.sync[16 + 16 * x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace

int mlp_chain_grid(const MlpProgram& p, int max_blocks) {
  const int mT = (p.B + 15) / 16;
  int need = 1;
  for (int s = 0; s < p.nstages; ++s) {
    int jobs = 0;
    for (int li = p.stage[s].layer0; li < p.stage[s].layer0 + p.stage[s].nlayers; ++li) jobs += p.layer[li].nT16 * mT;
    jobs += p.stage[s].nrows * mT;
    need = std::max(need, jobs);
  }
  return std::max(1, std::min(need, max_blocks));
}

int mlp_chain_resident_blocks() {
  static int cached = 0;
  if (cached > 0) return cached;
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mlp_chain_kernel, MLP_WAVES * 64, 0) != hipSuccess || per_cu < 1) per_cu = 1;
  cached = std::max(1, poco_num_cus() * per_cu);
  return cached;
}

int launch_mlp_chain(const MlpProgram& p, int max_blocks, hipStream_t s) {
  if (p.nstages < 1 || p.nstages > MLP_MAX_STAGES || p.B < 1 || !p.sync) { poco_set_error("mlp_chain: bad program"); return POCO_ERR_ARG; }
  for (int st = 0; st < p.nstages; ++st) {
    const MlpStage& g = p.stage[st];
    if (g.layer0 < 0 || g.layer0 + g.nlayers > MLP_MAX_LAYERS || g.row0 < 0 || g.row0 + g.nrows > MLP_MAX_ROWS) {
      poco_set_error("mlp_chain: stage " + std::to_string(st) + " out of range"); return POCO_ERR_ARG;
    }
  }
  // every block of the grid must be resident at once: clamp to what THIS device holds - a partitioned / smaller part
  // (CPX: 32 CUs) would otherwise spin every barrier into its time-out.  Results do not depend on the grid (fixed-order reductions).
  const int grid = mlp_chain_grid(p, std::min(max_blocks, mlp_chain_resident_blocks()));
  hipLaunchKernelGGL(mlp_chain_kernel, dim3(grid), dim3(MLP_WAVES * 64), 0, s, p);
  POCO_HIP_CHECK(hipGetLastError());
  return POCO_OK;
}
