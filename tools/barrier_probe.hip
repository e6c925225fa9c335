// barrier_probe.hip -- what does one trip of the ring GEMM's loop skeleton cost?  256 blocks (one per CU) of 4 or 8 waves run N iterations of
//   (a) s_barrier only   (b) s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier   (c) as (b) plus one LDS-DMA load per wave and iteration
// hipcc --offload-arch=gfx950 -O3 tools/barrier_probe.hip -o /tmp/barrier_probe && /tmp/barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void k_loop(const u32x4* __restrict__ src, int iters, uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) u32x4 lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint32_t x = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 2) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ((size_t)blockIdx.x * 8 + wave) * 64 + lane),
                                                   (__attribute__((address_space(3))) void*)(lds + (wave * 2 + (it & 1)) * 64), 16, 0, 0);
    if (MODE >= 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    x = x * 1664525u + 1013904223u;
  }
  if (x == 0x12345678u) sink[0] = x;
}

template <int MODE, int NW>
int run(const char* name, const u32x4* src, uint32_t* sink, int lds_kb) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_loop<MODE, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
  float t[2];
  for (int k = 0; k < 2; ++k) {
    const int iters = k ? 4032 : 32;
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL((k_loop<MODE, NW>), dim3(256), dim3(NW * 64), (size_t)lds_kb * 1024, 0, src, iters, sink);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    t[k] = best;
  }
  CK(hipGetLastError());
  printf("%-34s %d waves, %3d KiB LDS: launch with 32 trips %6.1f us; per trip %6.1f ns\n", name, NW, lds_kb, t[0] * 1e3, (t[1] - t[0]) * 1e6 / 4000.0);
  return 0;
}

int main() {
  u32x4* src; uint32_t* sink;
  CK(hipMalloc(&src, 4 << 20)); CK(hipMalloc(&sink, 64)); CK(hipMemset(src, 1, 4 << 20));
  run<0, 4>("s_barrier", src, sink, 16); run<0, 8>("s_barrier", src, sink, 16); run<0, 8>("s_barrier", src, sink, 128);
  run<1, 4>("s_waitcnt + s_barrier", src, sink, 16); run<1, 8>("s_waitcnt + s_barrier", src, sink, 128);
  run<2, 4>("LDS-DMA + s_waitcnt(0) + s_barrier", src, sink, 16); run<2, 8>("LDS-DMA + s_waitcnt(0) + s_barrier", src, sink, 128);
  return 0;
}
