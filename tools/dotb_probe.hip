// Stand-alone timing of the split-K matrix-core dot (k_dotb) on the decode shapes, with pieces switched off to find
// where the time goes.  Weights rotate over NL distinct buffers so that nothing stays in L2 / MALL between launches.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I maxtext-indextts2_amd/csrc -o /tmp/dotb_probe tools/dotb_probe.hip
#include "ixt_kernels.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// MODE bits: 1 = no x staging (LDS left as is), 2 = no MFMA, 4 = no stores, 8 = no weight loads
template <int NRB, int MODE>
__global__ __launch_bounds__(256, 2) void k_var(const u32x4* __restrict__ xt, const u32x4* __restrict__ wt, int K, int N, int R, float* __restrict__ out, int nwaves) {
  extern __shared__ __attribute__((aligned(16))) u32x4 xs[];
  const int tid = threadIdx.x, nthr = blockDim.x, wpb = nthr >> 6;
  const int wave = tid >> 6, lane = tid & 63;
  const int wid = blockIdx.x * wpb + wave;
  const int nsteps = K >> 5;
  const int nrbk = (R + 15) >> 4;
  const int c = blockIdx.y;
  const int r_in = lane & 15, cq = lane >> 4;
  const int tile = wid;
  const bool tile_ok = wid < nwaves;
  const u32x4* wbase = wt + (size_t)(tile_ok ? tile : 0) * nsteps * 64 + lane;
  u32x4 wv[16];
  const int s0 = c * 16;
  const int cs = min(16, nsteps - s0);
#pragma unroll
  for (int s = 0; s < 16; ++s) { if (!(MODE & 8) && tile_ok && s < cs) wv[s] = __builtin_nontemporal_load(wbase + (size_t)(s0 + s) * 64); else wv[s] = u32x4(lane + s); }
  if (!(MODE & 1)) {
    for (int e = tid; e < NRB * cs * 64; e += nthr) {
      const int b = e / (cs * 64), rem = e - b * (cs * 64);
      xs[b * 1024 + rem] = (b < nrbk) ? xt[((size_t)b * nsteps + s0) * 64 + rem] : u32x4(0);
    }
  }
  __syncthreads();
  f32x4 acc[NRB];
#pragma unroll
  for (int b = 0; b < NRB; ++b) acc[b] = f32x4(0.0f);
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    if (s < cs) {
      const bf16x8 wf = __builtin_bit_cast(bf16x8, wv[s]);
#pragma unroll
      for (int b = 0; b < NRB; ++b) {
        if (!(MODE & 2)) acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(bf16x8, xs[b * 1024 + s * 64 + lane]), acc[b], 0, 0, 0);
        else { acc[b][0] += __uint_as_float(wv[s].x); acc[b][1] += __uint_as_float(wv[s].y); acc[b][2] += __uint_as_float(wv[s].z); acc[b][3] += __uint_as_float(wv[s].w); }
      }
    }
  }
#pragma unroll
  for (int b = 0; b < NRB; ++b) {
    const int row = b * 16 + r_in;
    if (tile_ok && row < R && (!(MODE & 4) || acc[b][0] == 1.2345f))
      *reinterpret_cast<float4*>(out + ((size_t)c * R + row) * N + tile * 16 + cq * 4) = make_float4(acc[b][0], acc[b][1], acc[b][2], acc[b][3]);
  }
}

template <int MODE>
static float run(const char* what, int K, int N, int R, int wpb, int NL, bf16_t* w, bf16_t* x, float* out, int iters) {
  const int chains = (K + 511) / 512, ntile = N / 16;
  dim3 grid((ntile + wpb - 1) / wpb, chains, 1), block(64 * wpb);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const size_t welems = (size_t)K * N;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_var<4, MODE>), grid, block, 65536, 0, (const u32x4*)x, (const u32x4*)(w + (size_t)(i % NL) * welems), K, N, R, out, ntile);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k_var<4, MODE>), grid, block, 65536, 0, (const u32x4*)x, (const u32x4*)(w + (size_t)(i % NL) * welems), K, N, R, out, ntile);
  CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const float us = ms * 1000.f / iters;
  printf("  %-34s K=%5d N=%6d R=%3d wpb=%d grid=(%d,%d): %7.2f us/launch  %6.2f TB/s of weights\n", what, K, N, R, wpb, grid.x, grid.y, us, (double)welems * 2 / us / 1e6);
  return us;
}

int main() {
  const size_t wbytes = (size_t)3 << 30;   // 3 GiB of weights to rotate over
  bf16_t *w, *x; float* out;
  CK(hipMalloc(&w, wbytes)); CK(hipMalloc(&x, (size_t)64 * 8192 * 2)); CK(hipMalloc(&out, (size_t)16 * 64 * 16384 * 4));
  CK(hipMemset(w, 0x3c, wbytes)); CK(hipMemset(x, 0x3c, (size_t)64 * 8192 * 2));
  struct Shape { const char* name; int K, N; } shapes[] = {{"qkv", 2048, 6144}, {"o", 2048, 2048}, {"wi", 2048, 14336}, {"wo", 7168, 2048}};
  for (const Shape& s : shapes) {
    const int NL = (int)(wbytes / ((size_t)s.K * s.N * 2));
    printf("%s: %d rotating buffers\n", s.name, NL);
    for (int wpb : {4, 2}) {
      run<0>("full kernel", s.K, s.N, 64, wpb, NL, w, x, out, 200);
      run<1>("no x staging", s.K, s.N, 64, wpb, NL, w, x, out, 200);
      run<2>("no MFMA", s.K, s.N, 64, wpb, NL, w, x, out, 200);
      run<4>("no stores", s.K, s.N, 64, wpb, NL, w, x, out, 200);
      run<8>("no weight loads", s.K, s.N, 64, wpb, NL, w, x, out, 200);
      run<1 | 2 | 4>("weights only", s.K, s.N, 64, wpb, NL, w, x, out, 200);
    }
    run<0>("full kernel, 16 rows", s.K, s.N, 16, 4, NL, w, x, out, 200);
  }
  return 0;
}
