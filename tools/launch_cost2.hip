// Micro-benchmark 2: chains of DISTINCT small producer->consumer kernels in a hipGraph (what a decode layer looks like).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <vector>

__global__ __launch_bounds__(128) void k_a(const float* part, int slices, size_t stride, const float* hin, float* hout, int E) {
  const int c = blockIdx.x * 128 + threadIdx.x, r = blockIdx.y;
  float v[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) v[u] = u < slices ? part[(size_t)u * stride + (size_t)r * E + c] : 0.0f;
  float t = 0.0f;
#pragma unroll
  for (int u = 0; u < 16; ++u) t += v[u];
  hout[(size_t)r * E + c] = hin[(size_t)r * E + c] + t;
}
__global__ __launch_bounds__(256) void k_b(const float* h, float* xn, int E) {  // row norm with LDS + sync
  __shared__ float red[4];
  const int r = blockIdx.x, tid = threadIdx.x;
  float s = 0.0f;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { v[j] = h[(size_t)r * E + tid * 8 + j]; s += v[j] * v[j]; }
  for (int off = 1; off < 64; off <<= 1) s += __shfl_xor(s, off, 64);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  const float tot = red[0] + red[1] + red[2] + red[3];
  const float rs = 1.0f / sqrtf(tot / E + 1e-5f);
#pragma unroll
  for (int j = 0; j < 8; ++j) xn[(size_t)r * E + tid * 8 + j] = v[j] * rs;
}
__global__ __launch_bounds__(512) void k_c(const float* xn, float* part, int E, int slices, size_t stride) {  // fake "dot": writes slices from xn
  const int c = blockIdx.x * 512 + threadIdx.x;
  const int s = blockIdx.y;
  for (int r = 0; r < 8; ++r) part[(size_t)s * stride + (size_t)r * E + c] = xn[(size_t)r * E + c] * (1.0f / 16) ;
}

int main() {
  hipStream_t st; hipStreamCreate(&st);
  const int E = 2048, R = 8, S = 16;
  float *part, *h, *h2, *xn; hipMalloc(&part, sizeof(float) * S * R * E); hipMalloc(&h, sizeof(float) * R * E); hipMalloc(&h2, sizeof(float) * R * E); hipMalloc(&xn, sizeof(float) * R * E);
  hipMemset(part, 0, sizeof(float) * S * R * E); hipMemset(h, 0, sizeof(float) * R * E);
  auto layer = [&](hipStream_t s, int mode) {
    // mode 0: a->b->c dependent chain; mode 1: only a; 2: only b; 3: only c
    if (mode == 0 || mode == 1) hipLaunchKernelGGL(k_a, dim3(E / 128, R), dim3(128), 0, s, part, S, (size_t)R * E, h, h2, E);
    if (mode == 0 || mode == 2) hipLaunchKernelGGL(k_b, dim3(R), dim3(256), 0, s, h2, xn, E);
    if (mode == 0 || mode == 3) hipLaunchKernelGGL(k_c, dim3(E / 512, S), dim3(512), 0, s, xn, part, E, S, (size_t)R * E);
  };
  for (int mode = 0; mode < 4; ++mode) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    const int reps = 100;
    for (int i = 0; i < reps; ++i) layer(st, mode);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 3; ++i) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < 20; ++i) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    auto t1 = std::chrono::high_resolution_clock::now();
    const int nk = mode == 0 ? 3 : 1;
    printf("mode %d: %.2f us per kernel (%d kernels per iteration)\n", mode, std::chrono::duration<double, std::micro>(t1 - t0).count() / (20.0 * reps * nk), nk);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  }
  return 0;
}
