// attn_ceiling.hip -- how fast can the chip READ the bytes one decode-attention launch reads, in the pattern it reads them?
// The bench's decode step attends 256 rows x 16 kv heads to ~768 keys each from a paged pool ([page][kv head][...]: K piece and V piece of
// 8 KiB each per (page, head), pages scattered by a random page map).  This program reads exactly those bytes with no arithmetic:
//   (a) "paged":   grid (16 heads, 256 rows) x 256 threads as k_attn_decode, each block its row's pages of its head -- K pieces first, then V pieces
//   (b) "linear":  the same number of bytes as one contiguous stream, grid-stride (the streaming ceiling of the part)
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/attn_ceiling.hip -o /tmp/attn_ceiling && /tmp/attn_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <random>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int NKV = 16, HD = 128, P = 32;                 // C2 geometry: 16 kv heads of 128, 32-token pages
constexpr size_t PIECE = (size_t)P * HD * 2;              // bytes of one (page, head) K piece (and of one V piece): 8 KiB
constexpr size_t PIECE_V = PIECE / 16;                    // in 16-byte vectors

// block (head, row): the row's pages of that head; waves take pages round robin, a wave reads a piece as 8 x 1 KiB wave-loads
__global__ __launch_bounds__(256) void k_paged(const u32x4* __restrict__ kpool, const u32x4* __restrict__ vpool, const int32_t* __restrict__ page_map,
                                               const int32_t* __restrict__ lens, int mpg, u32x4* sink) {
  const int h = blockIdx.x, r = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int pages = (lens[r] + P - 1) / P;
  const int32_t* prow = page_map + (size_t)r * mpg;
  u32x4 acc = u32x4(0);
  for (int pass = 0; pass < 2; ++pass) {
    const u32x4* pool = pass ? vpool : kpool;
    for (int p = wave; p < pages; p += 4) {
      const u32x4* src = pool + ((size_t)prow[p] * NKV + h) * PIECE_V + lane;
#pragma unroll
      for (int i = 0; i < 8; ++i) { const u32x4 v = src[i * 64]; acc ^= v; }
    }
  }
  if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc;   // never true: keeps the loads
}

__global__ __launch_bounds__(256) void k_linear(const u32x4* __restrict__ a, size_t n, u32x4* sink) {
  u32x4 acc = u32x4(0);
  const size_t stride = (size_t)gridDim.x * 256 * 8;
  for (size_t i = (size_t)blockIdx.x * 256 * 8 + threadIdx.x; i < n; i += stride) {
#pragma unroll
    for (int u = 0; u < 8; ++u) if (i + (size_t)u * 256 < n) { const u32x4 v = a[i + (size_t)u * 256]; acc ^= v; }
  }
  if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc;
}

int main() {
  const int R = 256, Tmax = 1280, mpg = Tmax / P;
  const int num_pages = R * mpg + 1;
  const size_t pool_bytes = (size_t)num_pages * NKV * PIECE;     // one layer's K pool (V pool the same)
  u32x4 *kpool, *vpool, *sink; int32_t *pm, *lens;
  CK(hipMalloc(&kpool, pool_bytes)); CK(hipMalloc(&vpool, pool_bytes)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(kpool, 1, pool_bytes)); CK(hipMemset(vpool, 2, pool_bytes));
  std::vector<int32_t> hpm((size_t)R * mpg), hl(R);
  std::vector<int32_t> perm(num_pages - 1);
  for (int i = 0; i < num_pages - 1; ++i) perm[i] = i + 1;
  std::mt19937 rng(7);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int variant = 0; variant < 2; ++variant) {
    // variant 0: pages in allocation order (slot by slot, as the engine's page manager hands them out); 1: a random permutation
    if (variant) std::shuffle(perm.begin(), perm.end(), rng);
    for (int r = 0; r < R; ++r) for (int p = 0; p < mpg; ++p) hpm[(size_t)r * mpg + p] = perm[(size_t)r * mpg + p];
    for (int len : {256 + 512, 1280}) {
      double bytes = 0;
      for (int r = 0; r < R; ++r) { hl[r] = len; bytes += (double)((len + P - 1) / P) * NKV * PIECE * 2; }
      CK(hipMalloc(&pm, hpm.size() * 4)); CK(hipMalloc(&lens, R * 4));
      CK(hipMemcpy(pm, hpm.data(), hpm.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(lens, hl.data(), R * 4, hipMemcpyHostToDevice));
      float best = 1e9f;
      for (int it = 0; it < 6; ++it) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_paged, dim3(NKV, R), dim3(256), 0, 0, kpool, vpool, pm, lens, mpg, sink);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
      }
      printf("paged   %-9s %4d keys/row: %7.1f us for %.3f GB = %.2f TB/s (%.3f of 8 TB/s)\n", variant ? "random" : "in-order", len, best * 1e3, bytes / 1e9,
             bytes / best / 1e9, bytes / best / 1e9 / 8.0);
      if (!variant) {
        float bl = 1e9f;
        const size_t n = (size_t)(bytes / 2 / 16);
        for (int it = 0; it < 6; ++it) {
          CK(hipEventRecord(e0, 0));
          hipLaunchKernelGGL(k_linear, dim3(256 * 16), dim3(256), 0, 0, it & 1 ? vpool : kpool, n, sink);
          hipLaunchKernelGGL(k_linear, dim3(256 * 16), dim3(256), 0, 0, it & 1 ? kpool : vpool, n, sink);
          CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1)); bl = std::min(bl, ms);
        }
        printf("linear  (two launches)  same bytes: %7.1f us = %.2f TB/s (%.3f of 8 TB/s)\n", bl * 1e3, bytes / bl / 1e9, bytes / bl / 1e9 / 8.0);
      }
      hipFree(pm); hipFree(lens);
    }
  }
  return 0;
}
