// ixt_moe_kernels.h -- row kernels of the Mixtral block (layers/mixtral.py:52-170, layers/moe.py:170-204,
// :326-345, :420-484, :855-942).  Attention, norms and the per-expert MLP dots are the llama2 block's kernels; this file holds the router
// and the combination of the selected experts' outputs.  Canonical orders: gate logit = the vector order of the
// RMSNorm statistics (a lane's 8 consecutive k by fma, xor tree over the wave, blocks of 512 k in order; bf16 result); top_k = repeated
// first-index maximum; weights = bf16(exp*(v_j - v_0) / (e_0 + e_1 + ...)); mixture = fma chain over the k selected experts in top_k order.
#pragma once
#include "ixt_kernels.h"

#define IXT_MOE_MAX_EXPERTS 32
#define IXT_MOE_MAX_TOPK 8

// grid = R, block = 256 (4 waves, each takes every fourth expert's logit).  xn: the dot kernels' x layout (xidx); gate: row-major bf16 [E][n_exp].
__global__ __launch_bounds__(256) void k_moe_route(const bf16_t* xn, const bf16_t* gate, int E, int n_exp, int top_k, int32_t* idx_out /*[R][top_k]*/,
                                                  float* w_out /*[R][top_k]*/) {
  __shared__ float gl[IXT_MOE_MAX_EXPERTS];
  const int r = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = E / 8, nblk = (nch + 63) / 64;
  for (int e = wave; e < n_exp; e += 4) {
    float total = 0.0f;
    for (int b = 0; b < nblk; ++b) {
      const int ch = b * 64 + lane;
      float a = 0.0f;
      if (ch < nch) {
        const u32x4 xv = *reinterpret_cast<const u32x4*>(xn + xidx(r, ch * 8, E));
        const float x[8] = {bflo(xv.x), bfhi(xv.x), bflo(xv.y), bfhi(xv.y), bflo(xv.z), bfhi(xv.z), bflo(xv.w), bfhi(xv.w)};
#pragma unroll
        for (int j = 0; j < 8; ++j) a = __builtin_fmaf(x[j], bf2f(gate[(size_t)(ch * 8 + j) * n_exp + e]), a);
      }
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) a = a + __shfl_xor(a, off, 64);
      total = total + a;
    }
    if (lane == 0) gl[e] = rbf(total);
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  float v[IXT_MOE_MAX_TOPK], ex[IXT_MOE_MAX_TOPK];
  uint32_t taken = 0;
  for (int j = 0; j < top_k; ++j) {
    int best = -1;
    for (int e = 0; e < n_exp; ++e)
      if (!((taken >> e) & 1u) && (best < 0 || gl[e] > gl[best])) best = e;
    taken |= 1u << best;
    idx_out[(size_t)r * top_k + j] = best;
    v[j] = gl[best];
  }
  float s = 0.0f;
  for (int j = 0; j < top_k; ++j) { ex[j] = ixt_exp(v[j] - v[0]); s = s + ex[j]; }
  for (int j = 0; j < top_k; ++j) w_out[(size_t)r * top_k + j] = rbf(ex[j] / s);
}

// Expert e's down-projection slices -> bf16 -> the (row, j) slots that selected e.  grid = ((E + 255) / 256, R), block 256.
__global__ __launch_bounds__(256) void k_moe_scatter(const float* part, int slices, size_t slice_stride, int E, int e, int top_k, const int32_t* idx,
                                                     bf16_t* ybuf /*[R][top_k][E]*/) {
  const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
  if (c >= E) return;
  int j = -1;
  for (int t = 0; t < top_k; ++t)
    if (idx[(size_t)r * top_k + t] == e) j = t;
  if (j < 0) return;
  ybuf[((size_t)r * top_k + j) * E + c] = f2bf(sum_slices<16>(part + (size_t)r * E + c, slice_stride, slices));
}

// out[r][c] = bf16(fma chain over j of y_j * weight_j) as one f32 "slice" (the residual / norm kernels take it from there).
__global__ __launch_bounds__(256) void k_moe_combine(const bf16_t* ybuf, const float* w, int E, int top_k, float* out /*[R][E]*/, bf16_t* out16 = nullptr) {
  const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
  if (c >= E) return;
  float t = 0.0f;
  for (int j = 0; j < top_k; ++j) t = __builtin_fmaf(bf2f(ybuf[((size_t)r * top_k + j) * E + c]), w[(size_t)r * top_k + j], t);
  if (out != nullptr) out[(size_t)r * E + c] = rbf(t);
  if (out16 != nullptr) out16[(size_t)r * E + c] = f2bf(t);
}

// bf16(sum of slices) of a [R][N] projection (op-level entry points)
__global__ __launch_bounds__(256) void k_slices_to_bf16(const float* part, int slices, size_t slice_stride, size_t n, bf16_t* out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  out[i] = f2bf(sum_slices<16>(part + i, slice_stride, slices));
}
