// ingest_probe.hip -- how many bytes per clock can ONE CU bring from L2 into LDS?  The LDS-ring GEMMs of this build (csrc/ixt_gemm.h) are bound by
// what a block ingests (weight tile + activation tile), not by HBM or the matrix pipe; this program measures the three ways in:
//   dma    global_load_lds_dwordx4 (1 KiB per wave-instruction straight into LDS)
//   regs   global_load_dwordx4 into VGPRs, then ds_write_b128
//   load   global_load_dwordx4 only (what the vector memory path delivers without the LDS leg)
// each with 4 or 8 waves per block, one block per CU, over a 4 MB (L2-resident) and a 512 MB (HBM) source.
// hipcc --offload-arch=gfx950 -O3 tools/ingest_probe.hip -o /tmp/ingest_probe && /tmp/ingest_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// every wave streams `iters` groups of 8 KiB (8 wave-loads of 1 KiB) from its own stretch of src (wrapping inside `span` vectors)
template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void k_ingest(const u32x4* __restrict__ src, size_t span, int iters, u32x4* sink) {
  extern __shared__ __attribute__((aligned(16))) u32x4 lds[];   // [NW][2 buffers][8][64]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  size_t off = ((size_t)blockIdx.x * NW + wave) * 8 * 64 * 37 % span;   // scattered starts
  u32x4* mine = lds + (size_t)wave * 2 * 8 * 64;
  u32x4 acc = u32x4(0);
  for (int it = 0; it < iters; ++it) {
    const u32x4* p = src + off + lane;
    u32x4* dst = mine + (it & 1) * 8 * 64;
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + i * 64), (__attribute__((address_space(3))) void*)(dst + i * 64), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // the previous group has landed; this one flies on
    } else {
      u32x4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = p[i * 64];
      if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) dst[i * 64 + lane] = v[i];
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc ^= v[i];
      }
    }
    off += (size_t)gridDim.x * NW * 8 * 64;
    if (off + 8 * 64 > span) off %= (span - 8 * 64);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (MODE != 2) acc ^= mine[lane];
  if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc;
}

template <int MODE, int NW>
int run(const char* name, const u32x4* src, size_t span, u32x4* sink) {
  const int iters = 2048, blocks = 256;
  const size_t lds = (size_t)NW * 2 * 8 * 64 * 16;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ingest<MODE, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int r = 0; r < 4; ++r) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_ingest<MODE, NW>), dim3(blocks), dim3(NW * 64), lds, 0, src, span, iters, sink);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  CK(hipGetLastError());
  const double bytes = (double)blocks * NW * iters * 8192.0;
  printf("%-5s %d waves/CU, source %6.0f MB: %8.1f us  %7.2f TB/s chip  %6.1f GB/s per CU  (%.1f B/clk/CU at 2.4 GHz)\n", name, NW, span * 16.0 / 1e6, best * 1e3,
         bytes / best / 1e9, bytes / best / 1e6 / blocks, bytes / best / 1e6 / blocks / 2.4);
  return 0;
}

int main() {
  u32x4 *src, *sink;
  const size_t big = (size_t)512 << 20;
  CK(hipMalloc(&src, big)); CK(hipMalloc(&sink, 64)); CK(hipMemset(src, 1, big));
  for (size_t span : {(size_t)(4 << 20) / 16, big / 16}) {
    run<0, 4>("dma", src, span, sink); run<0, 8>("dma", src, span, sink);
    run<1, 4>("regs", src, span, sink); run<1, 8>("regs", src, span, sink);
    run<2, 4>("load", src, span, sink); run<2, 8>("load", src, span, sink);
  }
  return 0;
}
