// Micro-benchmark: cost of tiny dependent kernels on one stream and inside a hipGraph (MI355X).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty(int* p) {}
__global__ void k_touch(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = p[0] + 1; }
__global__ void k_chain(const int* idx, const float* a, float* out) {  // dependent load chain of depth 2 + store
  int i = idx[blockIdx.x];
  out[blockIdx.x * blockDim.x + threadIdx.x] = a[i + threadIdx.x] * 2.0f;
}
__global__ void k_stream(const float4* __restrict__ a, float4* __restrict__ out, size_t n) {  // pure streaming read+write
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { float4 v = a[i]; v.x += 1.0f; out[i] = v; }
}

template <typename F>
double time_stream(hipStream_t st, int n, F f) {
  for (int i = 0; i < 20; ++i) f(st);
  hipStreamSynchronize(st);
  auto t0 = std::chrono::high_resolution_clock::now();
  for (int i = 0; i < n; ++i) f(st);
  hipStreamSynchronize(st);
  auto t1 = std::chrono::high_resolution_clock::now();
  return std::chrono::duration<double, std::micro>(t1 - t0).count() / n;
}
template <typename F>
double time_graph(hipStream_t st, int per_graph, int reps, F f) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < per_graph; ++i) f(st);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int i = 0; i < 3; ++i) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  auto t0 = std::chrono::high_resolution_clock::now();
  for (int i = 0; i < reps; ++i) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  auto t1 = std::chrono::high_resolution_clock::now();
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return std::chrono::duration<double, std::micro>(t1 - t0).count() / (reps * (double)per_graph);
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  int* p; CK(hipMalloc(&p, 1 << 20)); CK(hipMemset(p, 0, 1 << 20));
  float *a, *out; CK(hipMalloc(&a, 64 << 20)); CK(hipMalloc(&out, 64 << 20)); CK(hipMemset(a, 0, 64 << 20));
  auto e1 = [&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, p); };
  auto e256 = [&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s, p); };
  auto e2048 = [&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(2048), dim3(256), 0, s, p); };
  auto t1 = [&](hipStream_t s) { hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, s, p); };
  auto c64 = [&](hipStream_t s) { hipLaunchKernelGGL(k_chain, dim3(64), dim3(256), 0, s, p, a, out); };
  auto s8 = [&](hipStream_t s) { hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, s, (const float4*)a, (float4*)out, (size_t)2048 * 256); };   // 8 MB read
  auto s32 = [&](hipStream_t s) { hipLaunchKernelGGL(k_stream, dim3(8192), dim3(256), 0, s, (const float4*)a, (float4*)out, (size_t)8192 * 256); };  // 32 MB read
  printf("%-28s %10s %10s\n", "kernel", "stream us", "graph us");
  printf("%-28s %10.2f %10.2f\n", "empty <<<1,64>>>", time_stream(st, 2000, e1), time_graph(st, 200, 20, e1));
  printf("%-28s %10.2f %10.2f\n", "empty <<<256,256>>>", time_stream(st, 2000, e256), time_graph(st, 200, 20, e256));
  printf("%-28s %10.2f %10.2f\n", "empty <<<2048,256>>>", time_stream(st, 2000, e2048), time_graph(st, 200, 20, e2048));
  printf("%-28s %10.2f %10.2f\n", "touch <<<1,64>>>", time_stream(st, 2000, t1), time_graph(st, 200, 20, t1));
  printf("%-28s %10.2f %10.2f\n", "chain <<<64,256>>>", time_stream(st, 2000, c64), time_graph(st, 200, 20, c64));
  printf("%-28s %10.2f %10.2f\n", "stream 8MB rd+wr", time_stream(st, 1000, s8), time_graph(st, 200, 10, s8));
  printf("%-28s %10.2f %10.2f\n", "stream 32MB rd+wr", time_stream(st, 1000, s32), time_graph(st, 200, 10, s32));
  return 0;
}
