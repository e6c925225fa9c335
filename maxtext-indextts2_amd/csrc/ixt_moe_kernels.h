// ixt_moe_kernels.h -- row kernels of the Mixtral block (layers/mixtral.py:52-170, layers/moe.py:170-204,
// :326-345, :420-484, :855-942).  Attention, norms and the per-expert MLP dots are the llama2 block's kernels; this file holds the router
// and the combination of the selected experts' outputs.  Canonical orders: gate logit = the vector order of the
// RMSNorm statistics (a lane's 8 consecutive k by fma, xor tree over the wave, blocks of 512 k in order; bf16 result); top_k = repeated
// first-index maximum; weights = bf16(exp*(v_j - v_0) / (e_0 + e_1 + ...)); mixture = fma chain over the k selected experts in top_k order.
#pragma once
#include "ixt_kernels.h"

#define IXT_MOE_MAX_EXPERTS 32
#define IXT_MOE_MAX_TOPK 8

// grid = R, block = 64 n_exp (one wave per expert: the n_exp dot products of a row run side by side, each wave's loads of a block of 512 k
// issued together).  xn: the dot kernels' x layout (xidx); gate: row-major bf16 [E][n_exp].
__global__ __launch_bounds__(1024) void k_moe_route(const bf16_t* xn, const bf16_t* gate, int E, int n_exp, int top_k, int32_t* idx_out /*[R][top_k]*/,
                                                   float* w_out /*[R][top_k]*/) {
  __shared__ float gl[IXT_MOE_MAX_EXPERTS];
  const int r = blockIdx.x, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nch = E / 8, nblk = (nch + 63) / 64;
  for (int e = wave; e < n_exp; e += nw) {
    float total = 0.0f;
    for (int b0 = 0; b0 < nblk; b0 += 4) {                 // four blocks of 512 k per round: 4 activation and 32 gate loads in flight per lane
      u32x4 xv[4];
      bf16_t gv[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ch = min((b0 + u) * 64 + lane, nch - 1);
        xv[u] = *reinterpret_cast<const u32x4*>(xn + xidx(r, ch * 8, E));
#pragma unroll
        for (int j = 0; j < 8; ++j) gv[u][j] = gate[(size_t)(ch * 8 + j) * n_exp + e];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (b0 + u < nblk) {
          const int ch = (b0 + u) * 64 + lane;
          float a = 0.0f;
          if (ch < nch) {
            const float x[8] = {bflo(xv[u].x), bfhi(xv[u].x), bflo(xv[u].y), bfhi(xv[u].y), bflo(xv[u].z), bfhi(xv[u].z), bflo(xv[u].w), bfhi(xv[u].w)};
#pragma unroll
            for (int j = 0; j < 8; ++j) a = __builtin_fmaf(x[j], bf2f(gv[u][j]), a);
          }
#pragma unroll
          for (int off = 1; off < 64; off <<= 1) a = a + __shfl_xor(a, off, 64);
          total = total + a;
        }
      }
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

// The same router with the work turned round (n_exp a multiple of 8): one wave per BLOCK of 512 k, all experts at once.  A lane's 8 k of the
// gate kernel are 8 n_exp consecutive bf16 (row-major [E][n_exp]): whole 16-byte loads, consecutive lanes on consecutive lines -- k_moe_route
// reads the same bytes two at a time, once per expert wave (21.5 us per launch at 4096 x 8; this form: see DESIGN.md section 10).  The order is the
// contract's: a lane's 8 k by fma, xor tree over the wave, then the blocks' sums added in block order by one thread per expert.
// grid = R, block = 64 min(blocks, 16).
template <int NG>   // NG = n_exp / 8
__global__ __launch_bounds__(1024) void k_moe_route_blocks(const bf16_t* xn, const bf16_t* gate, int E, int top_k, int32_t* idx_out, float* w_out) {
  constexpr int NEXP = NG * 8;
  __shared__ float part[64][NEXP];     // [block of 512 k][expert]
  __shared__ float gl[NEXP];
  const int r = blockIdx.x, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nch = E / 8, nblk = (nch + 63) / 64;
  for (int b = wave; b < nblk; b += nw) {
    const int ch = b * 64 + lane;
    const bool live = ch < nch;
    const int chc = min(ch, nch - 1);
    const u32x4 xv = *reinterpret_cast<const u32x4*>(xn + xidx(r, chc * 8, E));
    u32x4 gv[8][NG];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int gq = 0; gq < NG; ++gq) gv[j][gq] = *reinterpret_cast<const u32x4*>(gate + (size_t)(chc * 8 + j) * NEXP + gq * 8);
    const float x[8] = {bflo(xv.x), bfhi(xv.x), bflo(xv.y), bfhi(xv.y), bflo(xv.z), bfhi(xv.z), bflo(xv.w), bfhi(xv.w)};
    float a[NEXP];
#pragma unroll
    for (int e = 0; e < NEXP; ++e) a[e] = 0.0f;
    if (live) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) {
          const u32x4 gw = gv[j][gq];
          const float gf[8] = {bflo(gw.x), bfhi(gw.x), bflo(gw.y), bfhi(gw.y), bflo(gw.z), bfhi(gw.z), bflo(gw.w), bfhi(gw.w)};
#pragma unroll
          for (int e = 0; e < 8; ++e) a[gq * 8 + e] = __builtin_fmaf(x[j], gf[e], a[gq * 8 + e]);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < NEXP; ++e) {
      float t = a[e];
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) t = t + __shfl_xor(t, off, 64);
      if (lane == 0) part[b][e] = t;
    }
  }
  __syncthreads();
  if (threadIdx.x < NEXP) {
    float total = 0.0f;
    for (int b = 0; b < nblk; ++b) total = total + part[b][threadIdx.x];
    gl[threadIdx.x] = rbf(total);
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  float v[IXT_MOE_MAX_TOPK], ex[IXT_MOE_MAX_TOPK];
  uint32_t taken = 0;
  for (int j = 0; j < top_k; ++j) {
    int best = -1;
    for (int e = 0; e < NEXP; ++e)
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

// =====================================================================================================
// Routed ("grouped") evaluation of the experts (layers/moe.py:420-453 `permute`, :785-938 sparse_matmul; kernels/megablox/gmm.py:305): every
// (row, j) copy goes to the rows of ITS expert and each expert's kernels run over its own rows only.
//   k_moe_plan    pair p = r top_k + j of expert e gets destination row pos[p] = off[e] + (number of earlier pairs of e): the stable order of the
//                 reference's argsort.  off[e] is aligned to `align` rows (16: one row block of the dot kernels; 128: one row group of the ring
//                 GEMM), so no row block holds copies of two experts, and rb_exp[row block] = its expert (-1: nothing there -- the GEMM block
//                 leaves at once, the expert's kernels are not read).  One block of 1024 threads.
//   k_moe_gather  xg[pos[p]] = xn[r] in the dot kernels' x layout.
// The order of the copies inside an expert changes nothing: a row's result does not depend on its neighbours in a launch.
// =====================================================================================================
__global__ __launch_bounds__(1024) void k_moe_plan(const int32_t* idx /*[npairs]*/, int npairs, int e0, int n_exp, int align, int nrb_cap, int32_t* pos /*[npairs]*/,
                                                   int32_t* rb_exp /*[nrb_cap]*/, int32_t* cnt_off /*[2 n_exp]: counts, then offsets*/) {
  __shared__ int wsum[16];
  __shared__ int s_base, s_off;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < nrb_cap; i += 1024) rb_exp[i] = -1;
  for (int i = tid; i < npairs; i += 1024) pos[i] = -1;     // a copy of no resident expert (expert-parallel padding) stays without a row
  if (tid == 0) s_off = 0;
  __syncthreads();
  for (int e = 0; e < n_exp; ++e) {
    if (tid == 0) s_base = 0;
    __syncthreads();
    const int off = s_off;
    for (int p0 = 0; p0 < npairs; p0 += 1024) {
      const int p = p0 + tid;
      const bool mine = p < npairs && idx[p] - e0 == e;
      const unsigned long long bal = __ballot(mine);
      const int before = __popcll(bal & ((1ull << lane) - 1ull));
      if (lane == 0) wsum[wave] = __popcll(bal);
      __syncthreads();
      int wbefore = 0, total = 0;
#pragma unroll
      for (int w = 0; w < 16; ++w) { const int v = wsum[w]; if (w < wave) wbefore += v; total += v; }
      const int base = s_base;
      if (mine) pos[p] = off + base + wbefore + before;
      __syncthreads();
      if (tid == 0) s_base = base + total;
      __syncthreads();
    }
    const int n = s_base;
    const int nrb = (n + 15) >> 4;
    for (int i = tid; i < nrb; i += 1024) rb_exp[(off >> 4) + i] = e;
    if (tid == 0) {
      cnt_off[e] = n;
      cnt_off[n_exp + e] = off;
      s_off = off + ((n + align - 1) / align) * align;
    }
    __syncthreads();
  }
}

// grid = (ceil(E / 2048), npairs), block 256: 8 consecutive features per thread.
__global__ __launch_bounds__(256) void k_moe_gather(const bf16_t* xn /*xidx, [R][E]*/, const int32_t* pos, int top_k, int E, bf16_t* xg /*xidx*/) {
  const int k0 = (blockIdx.x * 256 + threadIdx.x) * 8, p = blockIdx.y;
  if (k0 >= E || pos[p] < 0) return;
  *reinterpret_cast<u32x4*>(xg + xidx(pos[p], k0, E)) = *reinterpret_cast<const u32x4*>(xn + xidx(p / top_k, k0, E));
}

// The mixture from the experts' down-projection slices in copy-row order: y_j = bf16(sum of slices of row pos[r top_k + j]),
// out[r][c] = bf16(fma chain over j of y_j weight_j) as one f32 slice (k_moe_combine's arithmetic).  grid = ((E + 255) / 256, R), block 256.
__global__ __launch_bounds__(256) void k_moe_combine_rows(const float* part, int slices, size_t slice_stride, const int32_t* pos, const float* w, int E, int top_k,
                                                          float* out /*[R][E]*/) {
  const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
  if (c >= E) return;
  float t = 0.0f;
  for (int j = 0; j < top_k; ++j) {
    const float y = rbf(sum_slices<16>(part + (size_t)pos[(size_t)r * top_k + j] * E + c, slice_stride, slices));
    t = __builtin_fmaf(y, w[(size_t)r * top_k + j], t);
  }
  out[(size_t)r * E + c] = rbf(t);
}

// op-level form of the routed evaluation (ixt_op_moe_experts): out[p] = bf16(sum of slices of row pos[p]).  grid = ((E + 255) / 256, npairs).
__global__ __launch_bounds__(256) void k_moe_rows_out(const float* part, int slices, size_t slice_stride, const int32_t* pos, int E, bf16_t* out /*[npairs][E]*/) {
  const int c = blockIdx.x * 256 + threadIdx.x, p = blockIdx.y;
  if (c >= E) return;
  out[(size_t)p * E + c] = pos[p] < 0 ? (bf16_t)0 : f2bf(sum_slices<16>(part + (size_t)pos[p] * E + c, slice_stride, slices));
}
