// Dump hardware golden vectors for v_mfma_f32_16x16x32_bf16: n cases of (a[32], b[32], c) -> d, cell (0,0) of a tile.
// Output: raw little-endian file  [int32 n][n*32 u16 a][n*32 u16 b][n f32 c][n f32 d]   (tests/golden/make_mfma_golden.py -> .npz)
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_dump tools/mfma_dump.hip && /tmp/mfma_dump gpurun_out/mfma_hw.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <random>

typedef short frag8 __attribute__((ext_vector_type(8)));
typedef float frag_cd __attribute__((ext_vector_type(4)));

__global__ void k_probe(const uint16_t* A, const uint16_t* B, const float* C, float* D) {
  const int t = blockIdx.x, l = threadIdx.x;
  const uint16_t* a = A + (size_t)t * 32; const uint16_t* b = B + (size_t)t * 32;
  frag_cd acc = {0.f, 0.f, 0.f, 0.f};
  if (l == 0) acc[0] = C[t];
  frag8 fa = {0,0,0,0,0,0,0,0}, fb = {0,0,0,0,0,0,0,0};
  if ((l & 15) == 0) for (int j = 0; j < 8; ++j) { fa[j] = (short)a[8 * (l >> 4) + j]; fb[j] = (short)b[8 * (l >> 4) + j]; }
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc, 0, 0, 0);
  if (l == 0) D[t] = acc[0];
}
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
  const char* path = argc > 1 ? argv[1] : "mfma_hw.bin";
  const int N = 48000;
  std::mt19937 rng(2026);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::uniform_int_distribution<int> ex(-10, 10);
  std::vector<uint16_t> A((size_t)N * 32, 0), B((size_t)N * 32, 0); std::vector<float> C(N, 0.f), D(N);
  for (int t = 0; t < N; ++t) {
    uint16_t* a = &A[(size_t)t * 32]; uint16_t* b = &B[(size_t)t * 32];
    const int cls = t % 8;
    if (cls < 4) {            // gaussians with growing exponent spread, all 32 k or a random prefix
      const int nact = (t % 3 == 0) ? 32 : 1 + (int)(rng() % 32);
      for (int k = 0; k < nact; ++k) { a[k] = f2bf(nd(rng) * ldexpf(1.f, cls ? ex(rng) * cls / 3 : 0)); b[k] = f2bf(nd(rng) * ldexpf(1.f, cls ? ex(rng) * cls / 3 : 0)); }
      C[t] = (t % 5 == 4) ? 0.f : nd(rng) * ldexpf(1.f, cls ? ex(rng) : 0);
      if (t % 7 == 3) { double s = 0; for (int k = 0; k < 32; ++k) s += (double)bf2f(a[k]) * bf2f(b[k]); C[t] = -(float)s * (1.f + (float)((int)(rng() % 5) - 2) * 1.1920929e-7f); }
      if (t % 11 == 5) C[t] *= 65536.f;
      if (t % 13 == 6) C[t] *= 1.f / 65536.f;
    } else if (cls == 4) {    // one group, wide spread, accumulator 0..11 binades above the largest product exponent
      const int np = 2 + (int)(rng() % 7); int ep = -1000;
      for (int k = 0; k < np; ++k) {
        a[k] = (uint16_t)(((rng() & 1) << 15) | ((127 - rng() % 15) << 7) | (rng() & 0x7f)); b[k] = (uint16_t)(((rng() & 1) << 15) | ((127 - rng() % 15) << 7) | (rng() & 0x7f));
        int e = ((a[k] >> 7) & 0xff) + ((b[k] >> 7) & 0xff) - 254; if (e > ep) ep = e;
      }
      uint32_t u = ((rng() & 1u) << 31) | ((uint32_t)(127 + ep + (int)(rng() % 12)) << 23) | (rng() & 0x7fffff); memcpy(&C[t], &u, 4);
    } else if (cls == 5) {    // sparse operands, signed zeros
      for (int k = 0; k < 32; ++k) { a[k] = (rng() % 3 == 0) ? (uint16_t)((rng() & 1) << 15) : f2bf(nd(rng)); b[k] = (rng() % 5 == 0) ? (uint16_t)0x8000 : f2bf(nd(rng)); }
      C[t] = (t % 3 == 0) ? ((t & 8) ? -0.f : 0.f) : nd(rng);
    } else if (cls == 6) {    // bf16 subnormal / tiny operands against 2^100
      for (int k = 0; k < 32; ++k) { a[k] = (uint16_t)((rng() & 0x807f) | ((rng() % 4 == 0) ? 0 : (1 + rng() % 3) << 7)); b[k] = f2bf(nd(rng) * ldexpf(1.f, 100)); }
      C[t] = nd(rng) * ldexpf(1.f, -30 + (int)(rng() % 8));
    } else {                  // magnitudes of a trained layer: activations ~1, weights ~0.02, running sum ~1
      for (int k = 0; k < 32; ++k) { a[k] = f2bf(nd(rng)); b[k] = f2bf(nd(rng) * 0.02f); }
      C[t] = nd(rng) * 0.5f;
    }
  }
  uint16_t *dA, *dB; float *dC, *dD;
  hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, D.size() * 4);
  hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_probe, dim3(N), dim3(64), 0, 0, dA, dB, dC, dD);
  if (hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("hip failure\n"); return 1; }
  FILE* f = fopen(path, "wb"); if (!f) { printf("cannot open %s\n", path); return 1; }
  int32_t n = N; fwrite(&n, 4, 1, f); fwrite(A.data(), 2, A.size(), f); fwrite(B.data(), 2, B.size(), f); fwrite(C.data(), 4, N, f); fwrite(D.data(), 4, N, f); fclose(f);
  printf("wrote %d cases to %s\n", N, path);
  return 0;
}
