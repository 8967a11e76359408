// stage-by-stage timing of the fused CLIFF regressor (csrc/mlp_chain.hip) on synthetic weights:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/probe/mlp_probe.hip -o tools/probe/mlp_probe && tools/probe/mlp_probe [B] [blocks]
#include "../../poco_amd/csrc/mlp_chain.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
void poco_set_error(const std::string& m) { fprintf(stderr, "error: %s\n", m.c_str()); }
static float* dalloc(size_t n, float v) { float* d; hipMalloc(&d, n * 4); std::vector<float> h(n, v); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); return d; }
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 64, blocks = argc > 2 ? atoi(argv[2]) : 128;
  const int XC = 2224;
  float* xc = dalloc((size_t)B * XC, 0.01f);
  float* h0 = dalloc((size_t)B * 1024, 0.f); float* h1 = dalloc((size_t)B * 1024, 0.f); float* h2 = dalloc((size_t)B * 1024, 0.f);
  float* u = dalloc((size_t)B * 448, 0.f); float* rot = dalloc((size_t)B * 224, 0.f); float* ext = dalloc((size_t)B * 4096, 0.f);
  float* init = dalloc(160, 0.1f);
  auto W = [&](int K, int N) { return reinterpret_cast<const float4*>(dalloc((size_t)K * N, 1e-3f)); };
  const float* bias = dalloc(2048, 0.f);
  MlpProgram p{};
  p.B = B;
  hipMalloc(&p.sync, 1024); hipMemset(p.sync, 0, 1024);
  hipHostMalloc(reinterpret_cast<void**>(&p.err_host), 64, hipHostMallocMapped); *p.err_host = 0;
  hipMalloc(&p.trace, 64 * 8);
  int nl = 0, nr = 0, ns = 0;
  auto layer = [&](const float* in, int in_rs, int K, float* out, int out_rs, int N, const float* res, int res_rs, int act) {
    MlpLayer& L = p.layer[nl++]; L.in = in; L.in_rs = in_rs; L.nC16 = K / 16; L.out = out; L.out_rs = out_rs; L.nT16 = N / 16; L.res = res; L.res_rs = res_rs;
    L.act = act; L.wfrag = W(K, N); L.bias = bias; ++p.stage[ns].nlayers; };
  auto row = [&](int kind, const float* src, int srs, float* dst, int drs, int n) {
    MlpRowJob& J = p.row[nr++]; J.kind = kind; J.src = src; J.src_rs = srs; J.dst = dst; J.dst_rs = drs; J.n = n; ++p.stage[ns].nrows; };
  auto stage = [&]() { ++ns; p.stage[ns] = MlpStage{nl, 0, nr, 0}; };
  p.stage[0] = MlpStage{0, 0, 0, 0};
  row(MLP_ROW_BCAST, init, 0, xc + 2052, XC, 157); row(MLP_ROW_COPY, ext, 3, xc + 2048, XC, 3);
  layer(xc, XC, 2048, h0, 1024, 1024, nullptr, 0, 0);
  for (int it = 0; it < 3; ++it) {
    stage(); layer(xc + 2048, XC, 176, h1, 1024, 1024, h0, 1024, 0);
    stage(); layer(h1, 1024, 1024, h2, 1024, 1024, nullptr, 0, 0);
    stage(); layer(h2, 1024, 1024, xc + 2052, XC, 160, xc + 2052, XC, 0);
    if (it == 0) layer(xc, XC, 2048, u, 448, 224, nullptr, 0, 2);          // featNet rides with the 40-job decoder stage
    if (it == 1) row(MLP_ROW_COPY, xc, XC, ext, 2048, 2048);                // ... and the uncert_feat copy with the next one
  }
  stage(); row(MLP_ROW_ROT6D, xc + 2052, XC, rot, 224, 0); row(MLP_ROW_COPY, xc + 2052, XC, ext, 144, 144); row(MLP_ROW_COPY, h2, 1024, ext, 1024, 1024);
  p.nstages = ns + 1;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e9;
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(a); launch_mlp_chain(p, blocks, 0); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
  }
  long long t[64]; hipMemcpy(t, p.trace, sizeof t, hipMemcpyDeviceToHost);
  printf("B=%d blocks=%d grid=%d: %.1f us (events), err=%u\n", B, blocks, mlp_chain_grid(p, blocks), best * 1e3, *p.err_host);
  for (int s = 0; s < p.nstages; ++s)
    printf("  stage %2d: jobs %6.2f us  barrier %6.2f us\n", s, (t[2 * s + 1] - t[2 * s]) * 0.01, s + 1 < p.nstages ? (t[2 * s + 2] - t[2 * s + 1]) * 0.01 : 0.0);
  return 0;
}
