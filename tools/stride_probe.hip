// stride_probe.hip -- does the 64 KiB spacing of the weight-tile streams cost bandwidth?  Every block of the ring GEMM streams 8 weight tiles
// (+ 8 activation row blocks) whose starts are exactly 64 KiB apart (tile pitch = K / 32 fragments of 1 KiB at K = 2048), 2 KiB of each per
// 64-k chunk, all blocks marching through k together.  This program issues that pattern with LDS-DMA loads -- per wave 4 streams, 1 KiB of
// each per iteration -- with the streams 64 KiB apart and with a skew of a few KiB added to the pitch.
// hipcc --offload-arch=gfx950 -O3 tools/stride_probe.hip -o /tmp/stride_probe && /tmp/stride_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int NW>
__global__ __launch_bounds__(NW * 64) void k_streams(const u32x4* __restrict__ src, size_t pitch_v, int iters, u32x4* sink) {
  extern __shared__ __attribute__((aligned(16))) u32x4 lds[];   // [NW][4 buffers][4][64]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t s0 = ((size_t)blockIdx.x * NW + wave) * 4;       // first of this wave's 4 streams
  u32x4* mine = lds + (size_t)wave * 4 * 4 * 64;
  for (int it = 0; it < iters; ++it) {
    u32x4* dst = mine + (it & 3) * 4 * 64;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32x4* p = src + (s0 + j) * pitch_v + (size_t)it * 64 + lane;    // 1 KiB per stream and iteration
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p, (__attribute__((address_space(3))) void*)(dst + j * 64), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");     // three groups stay in flight (the ring's depth)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const u32x4 v = mine[lane];
  if (v.x == 0x12345678u && v.y == 0x9abcdef0u) sink[0] = v;
}

int main() {
  const int blocks = 224, NW = 8, iters = 64;                    // 224 blocks x 8 waves x 4 streams x 32 x 2 KiB = 58.7 MB: one MLP-up launch
  const size_t streams = (size_t)blocks * NW * 4;
  u32x4 *src, *sink;
  const size_t bytes = streams * (65536 + 16384) + (1 << 20);
  CK(hipMalloc(&src, 2 * bytes)); CK(hipMalloc(&sink, 64)); CK(hipMemset(src, 1, 2 * bytes));
  const size_t lds = (size_t)NW * 4 * 4 * 64 * 16;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_streams<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int skew_kb : {0, 1, 2, 4, 8, 3}) {
    const size_t pitch_v = (65536 + (size_t)skew_kb * 1024) / 16;
    float best = 1e9f;
    for (int r = 0; r < 6; ++r) {
      const u32x4* base = src + (r & 1) * (bytes / 16);          // alternate two copies: no reuse out of the 256 MiB cache between repeats
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(k_streams<8>, dim3(blocks), dim3(NW * 64), lds, 0, base, pitch_v, iters, sink);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    CK(hipGetLastError());
    const double total = (double)streams * iters * 1024.0;
    printf("stream pitch 64 KiB + %d KiB: %7.1f us for %.1f MB = %.2f TB/s (%.1f GB/s per CU)\n", skew_kb, best * 1e3, total / 1e6, total / best / 1e9, total / best / 1e6 / blocks);
  }
  return 0;
}
