// clock_probe.hip -- what clock do the CUs actually run at?  (a) one wave per CU: a chain of dependent v_add_f32 (4 cycles per wave
// instruction); (b) every SIMD saturated with independent v_mfma_f32_16x16x32_bf16 (16 cycles each): the sustained matrix-core clock under
// full load; each also reports s_memtime ticks per microsecond.
// hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__global__ void k_valu(int iters, float* sink, unsigned long long* ticks) {
  float x = threadIdx.x;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 64; ++u) asm volatile("v_add_f32 %0, %0, %0" : "+v"(x));
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  if (x == 12345.0f) sink[0] = x;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}
__global__ __launch_bounds__(256) void k_mfma(int iters, float* sink, unsigned long long* ticks) {
  f32x4 a0 = f32x4(0.0f), a1 = f32x4(0.0f), a2 = f32x4(0.0f), a3 = f32x4(0.0f);
  bf16x8 b = bf16x8((short)(0x3f80 + threadIdx.x % 3)), c = bf16x8((short)0x3c00);
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, c, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, c, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, c, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, c, a3, 0, 0, 0);
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  const f32x4 s = a0 + a1 + a2 + a3;
  if (s[0] == 12345.0f) sink[0] = s[1];
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

int main() {
  float* sink; unsigned long long* ticks; CK(hipMalloc(&sink, 64)); CK(hipMalloc(&ticks, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    float ms; unsigned long long t;
    const int iv = 20000;
    CK(hipEventRecord(e0, 0)); hipLaunchKernelGGL(k_valu, dim3(256), dim3(64), 0, 0, iv, sink, ticks); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost));
    printf("dependent v_add_f32, 1 wave/CU: %8.1f us for %d instr -> %.2f GHz if 4 cycles each (issue-to-issue may be longer); s_memtime %.1f ticks/us\n", ms * 1e3, iv * 64,
           iv * 64 * 4.0 / (ms * 1e6), t / (ms * 1e3));
    for (int blocks : {256, 512}) {   // 4 or 8 waves per CU: 1 or 2 per SIMD
      const int im = 4000;
      CK(hipEventRecord(e0, 0)); hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, im, sink, ticks); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost));
      const double mfma_per_simd = (double)im * 32 * (blocks / 256);
      printf("v_mfma_f32_16x16x32_bf16, %d wave(s)/SIMD on all 256 CUs: %8.1f us -> %.1f ns per MFMA per SIMD = %.2f GHz at 16 cycles, %.0f TFLOP/s; s_memtime %.1f ticks/us\n", blocks / 256,
             ms * 1e3, ms * 1e6 / mfma_per_simd, mfma_per_simd * 16.0 / (ms * 1e6), mfma_per_simd * 1024.0 * 16384.0 / (ms * 1e-3) / 1e12, t / (ms * 1e3));
    }
  }
  return 0;
}
