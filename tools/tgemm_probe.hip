// tgemm_probe.hip -- the training GEMM (csrc/ixt_train_kernels.h: k_tgemm<TA, TB>) against a host dot on all four operand layouts, and its rate
// at the trainer's shapes.  Not product code.  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/tgemm_probe.hip -o /tmp/tgemm_probe && /tmp/tgemm_probe
#include "../maxtext-indextts2_amd/csrc/ixt_train_kernels.h"
#include <cstdio>
#include <vector>
#include <cmath>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static float b2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f2b(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }

template <bool TA, bool TB>
int run(int M, int N, int K, bool check) {
  std::vector<uint16_t> ha((size_t)M * K), hb((size_t)K * N);
  unsigned s = 99;
  for (auto& v : ha) { s = s * 1664525u + 1013904223u; v = f2b(((int)((s >> 8) & 0xffff) - 32768) / 32768.0f); }
  for (auto& v : hb) { s = s * 1664525u + 1013904223u; v = f2b(((int)((s >> 8) & 0xffff) - 32768) / 32768.0f); }
  // A(m, k): TA ? stored [k][m] : [m][k];  B(k, n): TB ? stored [k][n] : [n][k]
  TGemm g{};
  uint16_t *da, *db; float* dc;
  CK(hipMalloc(&da, ha.size() * 2)); CK(hipMalloc(&db, hb.size() * 2)); CK(hipMalloc(&dc, (size_t)M * N * 4));
  CK(hipMemcpy(da, ha.data(), ha.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
  g.A = da; g.sam = TA ? 1 : K; g.sak = TA ? M : 1;
  g.B = db; g.sbk = TB ? N : 1; g.sbn = TB ? 1 : K;
  g.C = dc; g.scm = N; g.M = M; g.N = N; g.K = K; g.nb2 = 1; g.out_bf16 = 0; g.accumulate = 0;
  const dim3 grid((N + 127) / 128, (M + 127) / 128, 1);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int it = 0; it < (check ? 1 : 5); ++it) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_tgemm<TA, TB>), grid, dim3(256), 0, 0, g);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  CK(hipGetLastError());
  if (check) {
    std::vector<float> hc((size_t)M * N);
    CK(hipMemcpy(hc.data(), dc, hc.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0; int bad = 0;
    for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) {
      double t = 0;
      for (int k = 0; k < K; ++k) t += (double)b2f(ha[TA ? (size_t)k * M + m : (size_t)m * K + k]) * b2f(hb[TB ? (size_t)k * N + n : (size_t)n * K + k]);
      const double d = fabs(t - hc[(size_t)m * N + n]);
      if (d > worst) worst = d;
      if (d > 1e-3 * (1 + fabs(t))) { if (bad < 4) printf("   (%d,%d): %f vs %f\n", m, n, hc[(size_t)m * N + n], t); ++bad; }
    }
    printf("TA=%d TB=%d  M=%d N=%d K=%d: max abs err %.3g, %d bad\n", TA, TB, M, N, K, worst, bad);
  } else {
    printf("TA=%d TB=%d  M=%d N=%d K=%d: %.1f us = %.1f TFLOP/s\n", TA, TB, M, N, K, best * 1e3, 2.0 * M * N * K / best / 1e9);
  }
  hipFree(da); hipFree(db); hipFree(dc);
  return 0;
}

int main() {
  for (int K : {64, 136}) {
    run<false, false>(136, 200, K, true); run<true, false>(136, 200, K, true); run<false, true>(136, 200, K, true); run<true, true>(136, 200, K, true);
  }
  // the trainer's shapes at 16 384 tokens: forward (B row-contiguous), d-activation (both k-contiguous), d-weight (both row-contiguous)
  run<false, true>(16384, 7168, 2048, false);
  run<false, false>(16384, 2048, 7168, false);
  run<true, true>(2048, 7168, 16384, false);
  run<false, true>(16384, 2048, 2048, false);
  return 0;
}
