// Characterise v_mfma_f32_16x16x32_bf16 arithmetic on gfx950: is it emulatable bit-exactly on a CPU?
// Models tried: acc = RN( exact(acc + sum_{k in group} a_k*b_k) ) with the 32 k processed in groups of G (ascending k),
// G in {1(sequential fma),2,4,8,16,32}; and truncating variants.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <random>

typedef short frag_ab __attribute__((ext_vector_type(8)));
typedef float frag_cd __attribute__((ext_vector_type(4)));

// A [16][32] bf16 row-major, B [32][16] bf16 row-major (B[k][n]), C/D [16][16] f32
__global__ void k_mfma(const uint16_t* A, const uint16_t* B, const float* C, float* D, int ntests) {
  const int t = blockIdx.x, l = threadIdx.x;
  if (t >= ntests) return;
  const uint16_t* a = A + (size_t)t * 512; const uint16_t* b = B + (size_t)t * 512; const float* c = C + (size_t)t * 256; float* d = D + (size_t)t * 256;
  frag_ab fa, fb;
  for (int j = 0; j < 8; ++j) {
    const int k = 8 * (l >> 4) + j;
    fa[j] = (short)a[(l & 15) * 32 + k];
    fb[j] = (short)b[k * 16 + (l & 15)];
  }
  frag_cd acc;
  for (int j = 0; j < 4; ++j) acc[j] = c[((l >> 4) * 4 + j) * 16 + (l & 15)];
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc, 0, 0, 0);
  for (int j = 0; j < 4; ++j) d[((l >> 4) * 4 + j) * 16 + (l & 15)] = acc[j];
}

static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }

// exact value m * 2^e of a finite float (m integer)
static void decomp(float f, long long* m, int* e) {
  uint32_t u; memcpy(&u, &f, 4);
  int ex = (u >> 23) & 0xff; long long man = u & 0x7fffff;
  if (ex == 0) { *e = -149; } else { man |= 0x800000; *e = ex - 150; }
  *m = (u >> 31) ? -man : man;
}
// round an exact integer*2^e to float, round-to-nearest-even (or toward zero if rz)
static float round_i128(__int128 v, int e, bool rz) {
  if (v == 0) return 0.0f;
  bool neg = v < 0; unsigned __int128 a = neg ? (unsigned __int128)(-v) : (unsigned __int128)v;
  int msb = 127; while (!((a >> msb) & 1)) --msb;
  if (msb > 23) {
    int sh = msb - 23;
    unsigned __int128 q = a >> sh, rem = a & ((((unsigned __int128)1) << sh) - 1), half = ((unsigned __int128)1) << (sh - 1);
    if (!rz && (rem > half || (rem == half && (q & 1)))) q++;
    a = q; e += sh;
  }
  float r = ldexpf((float)(unsigned long long)a, e);
  return neg ? -r : r;
}
static float model_group(const uint16_t* a, const uint16_t* b, float c, int m, int n, int G, const int* korder, bool rz = false) {
  float acc = c;
  for (int g0 = 0; g0 < 32; g0 += G) {
    long long ms[33]; int es[33]; int cnt = 0;
    decomp(acc, &ms[cnt], &es[cnt]); cnt++;
    for (int i = g0; i < g0 + G; ++i) {
      int k = korder[i];
      long long m1, m2; int e1, e2;
      decomp(bf2f(a[m * 32 + k]), &m1, &e1); decomp(bf2f(b[k * 16 + n]), &m2, &e2);
      ms[cnt] = (m1 >> 16) * (m2 >> 16); es[cnt] = e1 + e2 + 32; cnt++;   // bf16 mantissas have 16 trailing zero bits
    }
    int emin = 1 << 30;
    for (int i = 0; i < cnt; ++i) if (ms[i] != 0 && es[i] < emin) emin = es[i];
    __int128 s = 0;
    for (int i = 0; i < cnt; ++i) if (ms[i] != 0) s += ((__int128)ms[i]) << (es[i] - emin);
    acc = (emin == (1 << 30)) ? 0.0f : round_i128(s, emin, rz);
  }
  return acc;
}

int main() {
  const int NT = 4096;
  std::mt19937 rng(123);
  std::vector<uint16_t> A((size_t)NT * 512), B((size_t)NT * 512);
  std::vector<float> C((size_t)NT * 256), D((size_t)NT * 256), D2((size_t)NT * 256);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::uniform_int_distribution<int> ex(-12, 12);
  for (int t = 0; t < NT; ++t) {
    const int spread = t % 4;  // 0: plain gaussians, 1..3: growing exponent spread
    for (int i = 0; i < 512; ++i) {
      A[(size_t)t * 512 + i] = f2bf(nd(rng) * ldexpf(1.f, spread ? ex(rng) * spread / 3 : 0));
      B[(size_t)t * 512 + i] = f2bf(nd(rng) * ldexpf(1.f, spread ? ex(rng) * spread / 3 : 0));
    }
    for (int i = 0; i < 256; ++i) C[(size_t)t * 256 + i] = (t % 8 == 7) ? 0.f : nd(rng) * ldexpf(1.f, spread ? ex(rng) : 0);
  }
  // hand-made single-rounding probes in test 0: row 0 col 0: C = 2^24, products 1 and 1 (k=0, k=17)
  for (int i = 0; i < 512; ++i) { A[i] = 0; B[i] = 0; }
  for (int i = 0; i < 256; ++i) C[i] = 0.f;
  A[0 * 32 + 0] = f2bf(1.f); B[0 * 16 + 0] = f2bf(1.f); A[0 * 32 + 17] = f2bf(1.f); B[17 * 16 + 0] = f2bf(1.f); C[0] = 16777216.f;
  // row 1 col 1: 2^40, -2^40, 1 with C = 0  (k = 0, 1, 2)
  A[1 * 32 + 0] = f2bf(1048576.f); B[0 * 16 + 1] = f2bf(1048576.f); A[1 * 32 + 1] = f2bf(-1048576.f); B[1 * 16 + 1] = f2bf(1048576.f);
  A[1 * 32 + 2] = f2bf(1.f); B[2 * 16 + 1] = f2bf(1.f);
  // row 2 col 2: same but the small term first and the big ones in another lane group (k = 0 small, k = 8, 9 big)
  A[2 * 32 + 0] = f2bf(1.f); B[0 * 16 + 2] = f2bf(1.f); A[2 * 32 + 8] = f2bf(1048576.f); B[8 * 16 + 2] = f2bf(1048576.f);
  A[2 * 32 + 9] = f2bf(-1048576.f); B[9 * 16 + 2] = f2bf(1048576.f);
  // row 3 col 3: 2^60 - 2^60 + 1
  A[3 * 32 + 0] = f2bf(1073741824.f); B[0 * 16 + 3] = f2bf(1073741824.f); A[3 * 32 + 1] = f2bf(-1073741824.f); B[1 * 16 + 3] = f2bf(1073741824.f);
  A[3 * 32 + 2] = f2bf(1.f); B[2 * 16 + 3] = f2bf(1.f);
  // row 4 col 4: subnormal-range product: 2^-70 * 2^-70 = 2^-140 (f32 subnormal), C = 0
  A[4 * 32 + 0] = f2bf(ldexpf(1.f, -70)); B[0 * 16 + 4] = f2bf(ldexpf(1.f, -70));

  uint16_t *dA, *dB; float *dC, *dD;
  hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, D.size() * 4);
  hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_mfma, dim3(NT), dim3(64), 0, 0, dA, dB, dC, dD, NT);
  hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  printf("probe single rounding (2^24+1+1): got %.1f (fused -> 16777218, sequential -> 16777216)\n", D[0]);
  printf("probe 2^40-2^40+1 same lane group: got %g ; small-first other group: got %g ; 2^60-2^60+1: got %g ; 2^-140: got %g\n", D[1 * 16 + 1], D[2 * 16 + 2], D[3 * 16 + 3], D[4 * 16 + 4]);

  // permutation invariance: permute k identically in A and B
  std::vector<uint16_t> A2 = A, B2 = B;
  std::vector<int> perm(32);
  for (int t = 1; t < NT; ++t) {
    for (int i = 0; i < 32; ++i) perm[i] = i;
    std::shuffle(perm.begin(), perm.end(), rng);
    for (int m = 0; m < 16; ++m) for (int k = 0; k < 32; ++k) A2[(size_t)t * 512 + m * 32 + k] = A[(size_t)t * 512 + m * 32 + perm[k]];
    for (int k = 0; k < 32; ++k) for (int n = 0; n < 16; ++n) B2[(size_t)t * 512 + k * 16 + n] = B[(size_t)t * 512 + perm[k] * 16 + n];
  }
  hipMemcpy(dA, A2.data(), A2.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B2.data(), B2.size() * 2, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_mfma, dim3(NT), dim3(64), 0, 0, dA, dB, dC, dD, NT);
  hipMemcpy(D2.data(), dD, D2.size() * 4, hipMemcpyDeviceToHost);
  long diff = 0, tot = 0;
  for (size_t i = 256; i < D.size(); ++i) { tot++; if (memcmp(&D[i], &D2[i], 4)) diff++; }
  printf("k-permutation changes %ld of %ld outputs (%.3f%%)\n", diff, tot, 100.0 * diff / tot);

  int korder[32];
  for (int i = 0; i < 32; ++i) korder[i] = i;
  for (int G : {1, 2, 4, 8, 16, 32}) {
    long bad = 0, n = 0; double worst = 0;
    for (int t = 1; t < NT; ++t)
      for (int m = 0; m < 16; ++m)
        for (int nn = 0; nn < 16; ++nn) {
          float ref = model_group(&A[(size_t)t * 512], &B[(size_t)t * 512], C[(size_t)t * 256 + m * 16 + nn], m, nn, G, korder);
          float got = D[(size_t)t * 256 + m * 16 + nn];
          n++;
          if (memcmp(&ref, &got, 4)) { bad++; double e = fabs((double)ref - got) / (fabs((double)ref) + 1e-30); if (e > worst) worst = e; }
        }
    printf("model group=%2d (ascending k): mismatches %ld / %ld (%.4f%%), worst rel diff %.3g\n", G, bad, n, 100.0 * bad / n, worst);
  }
  // per spread class for G=32
  for (int sp = 0; sp < 4; ++sp) {
    long bad = 0, n = 0;
    for (int t = 1; t < NT; ++t) {
      if (t % 4 != sp) continue;
      for (int m = 0; m < 16; ++m) for (int nn = 0; nn < 16; ++nn) {
        float ref = model_group(&A[(size_t)t * 512], &B[(size_t)t * 512], C[(size_t)t * 256 + m * 16 + nn], m, nn, 32, korder);
        float got = D[(size_t)t * 256 + m * 16 + nn];
        n++; if (memcmp(&ref, &got, 4)) bad++;
      }
    }
    printf("  exact-32 model, exponent spread class %d: mismatches %ld / %ld\n", sp, bad, n);
  }
  return 0;
}
