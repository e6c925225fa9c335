// ring_probe.hip -- diagnostic build of the LDS-ring GEMM (csrc/ixt_gemm.h) with in-kernel cycle stamps: how many cycles of an iteration
// go to the counted wait, the barrier, the issue of the LDS-DMA loads and the LDS reads + MFMAs.  Not product code; run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/ring_probe.hip -o /tmp/ring_probe && /tmp/ring_probe
#define IXT_RING_STAMPS 1
#include "../maxtext-indextts2_amd/csrc/ixt_kernels.h"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int WR, int WC, int RB, int CT, int KS, int NS>
int run(const char* label, int R, int K, int N) {
  const size_t nx = (size_t)((R + 15) / 16) * 16 * K, nw = (size_t)K * ((N + 127) / 128) * 128;
  uint16_t *x, *w; float* out; unsigned long long* dbg;
  CK(hipMalloc(&x, nx * 2)); CK(hipMalloc(&w, nw * 2 * 4)); CK(hipMalloc(&out, (size_t)R * N * 4));
  std::vector<uint16_t> hx(nx), hw(nw);
  unsigned s = 12345;
  for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = (uint16_t)(0x3c00 + ((s >> 16) & 0x3ff)) | ((s >> 3) & 0x8000); }
  for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = (uint16_t)(0x3a00 + ((s >> 16) & 0x3ff)) | ((s >> 3) & 0x8000); }
  CK(hipMemcpy(x, hx.data(), nx * 2, hipMemcpyHostToDevice));
  for (int c = 0; c < 4; ++c) CK(hipMemcpy(w + c * nw, hw.data(), nw * 2, hipMemcpyHostToDevice));
  const int ncolg = (N / 16 + WC * CT - 1) / (WC * CT), zb = (R + 16 * WR * RB - 1) / (16 * WR * RB);
  const int nblk = ((ncolg + 7) / 8) * 8 * zb;
  CK(hipMalloc(&dbg, (size_t)nblk * 32 * 8)); CK(hipMemset(dbg, 0, (size_t)nblk * 32 * 8));
  constexpr size_t lds = (size_t)NS * (WC * CT + WR * RB) * KS * 1024; constexpr int NW = WR * WC;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_ring<WR, WC, RB, CT, KS, NS, GEPI_F32, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  GemmEpi e{};
  e.out = out; e.amax_val = reinterpret_cast<float*>(dbg);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9f;
  for (int it = 0; it < 8; ++it) {
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((k_gemm_ring<WR, WC, RB, CT, KS, NS, GEPI_F32, 0>), dim3(nblk), dim3(NW * 64), lds, 0, reinterpret_cast<const u32x4*>(x),
                       reinterpret_cast<const u32x4*>(w + (size_t)(it % 4) * nw), K, N, R, ncolg, zb, 0, e);
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
  }
  std::vector<unsigned long long> h((size_t)nblk * NW * 4);
  CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
  double sums[4] = {0, 0, 0, 0}; int n = 0;
  for (int bl = 0; bl < nblk; ++bl) for (int wv = 0; wv < NW; ++wv) { const unsigned long long* d = &h[((size_t)bl * NW + wv) * 4]; if (d[3] == 0) continue; for (int j = 0; j < 4; ++j) sums[j] += (double)d[j]; ++n; }
  const int nch = (K / 32) / KS;
  printf("%-18s R=%d K=%d N=%d blocks=%d  %.1f us (stamped build)  per iteration [cycles]: wait %.0f  barrier %.0f  issue %.0f  lds+mfma %.0f  (sum %.0f)\n", label, R, K, N, nblk,
         best * 1e3, sums[0] / n / nch, sums[1] / n / nch, sums[2] / n / nch, sums[3] / n / nch, (sums[0] + sums[1] + sums[2] + sums[3]) / n / nch);
  hipFree(x); hipFree(w); hipFree(out); hipFree(dbg);
  return 0;
}

int main() {
  run<2, 2, 4, 4, 2, 4>("128x128/4w wi", 256, 2048, 14336);
  run<2, 4, 4, 2, 2, 4>("128x128/8w wi", 256, 2048, 14336);
  run<2, 4, 4, 2, 2, 4>("128x128/8w o", 256, 2048, 2048);
  run<2, 2, 4, 2, 2, 5>("128x64/4w qkv", 256, 2048, 6144);
  run<2, 4, 4, 1, 2, 5>("128x64/8w qkv", 256, 2048, 6144);
  run<2, 2, 2, 1, 4, 6>("64x32/4w o", 256, 2048, 2048);
  run<4, 2, 1, 1, 4, 6>("64x32/8w o", 256, 2048, 2048);
  run<2, 2, 2, 1, 4, 6>("64x32/4w wo", 256, 7168, 2048);
  run<4, 2, 1, 1, 4, 6>("64x32/8w wo", 256, 7168, 2048);
  run<2, 4, 2, 1, 4, 4>("64x64/8w 128k wo", 256, 7168, 2048);
  return 0;
}
