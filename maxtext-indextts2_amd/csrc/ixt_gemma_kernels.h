// ixt_gemma_kernels.h -- row kernels of the Gemma-2 block (layers/gemma2.py:69-236, decoders.py:552-565).
// The projections, RoPE, page writes and the attention kernels are the ones of the llama2 block (ixt_kernels.h); what differs is what
// happens between them: the projection output is normalised before the residual add, the gate is GELU(tanh), and the logits are
// divided by sqrt(emb) and soft-capped.  Same canonical orders as ixt_kernels.h (slices from 0; sum of squares: a lane's 8 features by
// fma, xor tree over the wave, blocks of 512 features in order).
#pragma once
#include "ixt_kernels.h"

// =====================================================================================================
// o = bf16(sum of slices)                                 the O- or down-projection of row r
// o = bf16(bf16(o * rsqrt(mean(o^2) + eps)) * post)       when post != nullptr   (gemma2.py:107-116, :135-143)
// h = bf16(h_in + o) -> h_out                             (gemma2.py:118, :146)
// xn = bf16(bf16(h * rsqrt(mean(h^2) + eps)) * next)      when next != nullptr (the following pre-norm), dot kernel's x layout
// grid = R, block 256, dynamic LDS = (E + 128) floats.
// =====================================================================================================
struct GNormArgs {
  const float* part; int slices; size_t slice_stride;
  const bf16_t* h_in; bf16_t* h_out;
  const bf16_t* post; const bf16_t* next;
  bf16_t* xn_out;
  int E; float eps;
};

__global__ __launch_bounds__(256) void k_gnorm(GNormArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* row = smem;             // [E] o, then h
  float* blk = smem + a.E;       // [<=64] block sums of o^2
  float* blk2 = blk + 64;        // [<=64] block sums of h^2
  const int r = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int E = a.E, nch = E / 8, nblk = (nch + 63) / 64;
  // pass 1: o and its sum of squares
  for (int it = 0; it * 4 < nblk; ++it) {
    const int b = it * 4 + wave;
    const int ch = b * 64 + lane;
    float v = 0.0f;
    if (b < nblk && ch < nch) {
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = 0.0f;
      for (int s = 0; s < a.slices; ++s) {
        const float* p = a.part + (size_t)s * a.slice_stride + (size_t)r * E + ch * 8;
        const float4 p0 = *reinterpret_cast<const float4*>(p), p1 = *reinterpret_cast<const float4*>(p + 4);
        o[0] = o[0] + p0.x; o[1] = o[1] + p0.y; o[2] = o[2] + p0.z; o[3] = o[3] + p0.w;
        o[4] = o[4] + p1.x; o[5] = o[5] + p1.y; o[6] = o[6] + p1.z; o[7] = o[7] + p1.w;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        o[j] = rbf(o[j]);
        row[ch * 8 + j] = o[j];
        v = __builtin_fmaf(o[j], o[j], v);
      }
    }
    if (b < nblk) {
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) v = v + __shfl_xor(v, off, 64);
      if (lane == 0) blk[b] = v;
    }
  }
  __syncthreads();
  float rs1 = 0.0f;
  if (a.post != nullptr) {
    float total = 0.0f;
    for (int b = 0; b < nblk; ++b) total = total + blk[b];
    rs1 = 1.0f / sqrtf(total / (float)E + a.eps);
  }
  // pass 2: post norm, residual, sum of squares of h (each thread revisits the chunks it wrote)
  for (int it = 0; it * 4 < nblk; ++it) {
    const int b = it * 4 + wave;
    const int ch = b * 64 + lane;
    float v = 0.0f;
    if (b < nblk && ch < nch) {
      const u32x4 hv = *reinterpret_cast<const u32x4*>(a.h_in + (size_t)r * E + ch * 8);
      const float hin[8] = {bflo(hv.x), bfhi(hv.x), bflo(hv.y), bfhi(hv.y), bflo(hv.z), bfhi(hv.z), bflo(hv.w), bfhi(hv.w)};
      float sc[8];
      if (a.post != nullptr) {
        const u32x4 sv = *reinterpret_cast<const u32x4*>(a.post + ch * 8);
        sc[0] = bflo(sv.x); sc[1] = bfhi(sv.x); sc[2] = bflo(sv.y); sc[3] = bfhi(sv.y);
        sc[4] = bflo(sv.z); sc[5] = bfhi(sv.z); sc[6] = bflo(sv.w); sc[7] = bfhi(sv.w);
      }
      float h[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float o = row[ch * 8 + j];
        if (a.post != nullptr) o = rbf(rbf(o * rs1) * sc[j]);
        h[j] = rbf(hin[j] + o);
        row[ch * 8 + j] = h[j];
        v = __builtin_fmaf(h[j], h[j], v);
      }
      u32x4 ov;
      ov.x = pack2(f2bf(h[0]), f2bf(h[1])); ov.y = pack2(f2bf(h[2]), f2bf(h[3]));
      ov.z = pack2(f2bf(h[4]), f2bf(h[5])); ov.w = pack2(f2bf(h[6]), f2bf(h[7]));
      *reinterpret_cast<u32x4*>(a.h_out + (size_t)r * E + ch * 8) = ov;
    }
    if (b < nblk) {
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) v = v + __shfl_xor(v, off, 64);
      if (lane == 0) blk2[b] = v;
    }
  }
  __syncthreads();
  if (a.next == nullptr) return;
  float total = 0.0f;
  for (int b = 0; b < nblk; ++b) total = total + blk2[b];
  const float rs = 1.0f / sqrtf(total / (float)E + a.eps);
  for (int ch = tid; ch < nch; ch += 256) {
    const u32x4 sv = *reinterpret_cast<const u32x4*>(a.next + ch * 8);
    const float sc[8] = {bflo(sv.x), bfhi(sv.x), bflo(sv.y), bfhi(sv.y), bflo(sv.z), bfhi(sv.z), bflo(sv.w), bfhi(sv.w)};
    bf16_t o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f2bf(rbf(row[ch * 8 + j] * rs) * sc[j]);
    u32x4 ov;
    ov.x = pack2(o[0], o[1]); ov.y = pack2(o[2], o[3]); ov.z = pack2(o[4], o[5]); ov.w = pack2(o[6], o[7]);
    *reinterpret_cast<u32x4*>(a.xn_out + xidx(r, ch * 8, E)) = ov;
  }
}

// =====================================================================================================
// GELU(tanh) gate from partial slices: k_swiglu_part with ixt_geglu (linears.py:449-464 with mlp_activations ["gelu", "linear"]).
// =====================================================================================================
__global__ __launch_bounds__(256) void k_geglu_part(const float* part, int slices, size_t slice_stride, int F, bf16_t* out /*[R][F] in xidx layout*/) {
  const int f0 = (blockIdx.x * 256 + threadIdx.x) * 8, r = blockIdx.y;
  if (f0 >= F) return;
  const float* p = part + (size_t)r * 2 * F;
  float a[8], b[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = 0.0f; b[j] = 0.0f; }
  for (int s = 0; s < slices; ++s) {
    const float* ps = p + (size_t)s * slice_stride;
    const int ca = (f0 >> 4) * 32 + (f0 & 15);   // wi_0 tile j = column tile 2 j, wi_1 tile j = column tile 2 j + 1 (k_tile_w mode 2)
    const float4 a0 = *reinterpret_cast<const float4*>(ps + ca), a1 = *reinterpret_cast<const float4*>(ps + ca + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(ps + ca + 16), b1 = *reinterpret_cast<const float4*>(ps + ca + 20);
    a[0] = a[0] + a0.x; a[1] = a[1] + a0.y; a[2] = a[2] + a0.z; a[3] = a[3] + a0.w;
    a[4] = a[4] + a1.x; a[5] = a[5] + a1.y; a[6] = a[6] + a1.z; a[7] = a[7] + a1.w;
    b[0] = b[0] + b0.x; b[1] = b[1] + b0.y; b[2] = b[2] + b0.z; b[3] = b[3] + b0.w;
    b[4] = b[4] + b1.x; b[5] = b[5] + b1.y; b[6] = b[6] + b1.z; b[7] = b[7] + b1.w;
  }
  bf16_t o[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = ixt_geglu(rbf(a[j]), rbf(b[j]));
  u32x4 ov;
  ov.x = pack2(o[0], o[1]); ov.y = pack2(o[2], o[3]); ov.z = pack2(o[4], o[5]); ov.w = pack2(o[6], o[7]);
  *reinterpret_cast<u32x4*>(out + xidx(r, f0, F)) = ov;
}

__global__ __launch_bounds__(256) void k_geglu(const bf16_t* a, const bf16_t* b, bf16_t* out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = ixt_geglu(bf2f(a[i]), bf2f(b[i]));
}

// =====================================================================================================
// Logits of the tied head (decoders.py:552-565), in place on bf16-valued f32 rows, then greedy sampling over the allowed columns
// (first index of the maximum; inference_utils.py:74-75).  normalize: v / sqrt(emb) in f32 (the divisor is an f32 array: the logits
// become f32) and the cap chain in f32; otherwise every primitive of the cap rounds to bf16.  grid = rows, block 256.
// =====================================================================================================
__device__ __forceinline__ float ixt_logit_post(float v, int normalize, float sqrt_emb, float cap) {
  if (normalize) {
    v = v / sqrt_emb;
    if (cap > 0.0f) v = ixt_tanh(v / cap) * cap;
    return v;
  }
  if (cap > 0.0f) v = rbf(rbf(ixt_tanh(rbf(v / cap))) * cap);
  return v;
}

// grid = (ceil(vocab / IXT_LPOST_COLS), rows), block 256: a block post-processes IXT_LPOST_COLS consecutive columns of one row and leaves its
// (maximum, first index) over the allowed columns in amax_val / amax_idx [row][block] for k_argmax_final (one block per row took 0.48 ms at
// 258 176 columns: 9 % of a Gemma-2 2B decode step).
#define IXT_LPOST_COLS 4096
__global__ __launch_bounds__(256) void k_logits_post(float* logits, int vocab, int normalize, float sqrt_emb, float cap, ColRanges cols, float* amax_val,
                                                     int32_t* amax_idx) {
  __shared__ float bv[256];
  __shared__ int bi[256];
  const int r = blockIdx.y, tid = threadIdx.x;
  float* row = logits + (size_t)r * vocab;
  float best = -INFINITY;
  int bidx = 0x7fffffff;
  const int n_end = min(vocab, (int)(blockIdx.x + 1) * IXT_LPOST_COLS);
  for (int n = blockIdx.x * IXT_LPOST_COLS + tid; n < n_end; n += 256) {
    const float v = ixt_logit_post(row[n], normalize, sqrt_emb, cap);
    row[n] = v;
    bool ok = cols.n == 0;
    for (int i = 0; i < cols.n; ++i) ok = ok || (n >= cols.lo[i] && n < cols.hi[i]);
    if (ok && (bidx == 0x7fffffff || v > best)) { best = v; bidx = n; }   // n ascends within a thread: the first maximum is kept
  }
  bv[tid] = best; bi[tid] = bidx;
  __syncthreads();
  for (int s2 = 128; s2 > 0; s2 >>= 1) {
    if (tid < s2) {
      const float v = bv[tid + s2]; const int i2 = bi[tid + s2];
      if (i2 != 0x7fffffff && (bi[tid] == 0x7fffffff || v > bv[tid] || (v == bv[tid] && i2 < bi[tid]))) { bv[tid] = v; bi[tid] = i2; }
    }
    __syncthreads();
  }
  if (tid == 0) {
    amax_val[(size_t)r * gridDim.x + blockIdx.x] = bi[0] == 0x7fffffff ? -INFINITY : bv[0];
    amax_idx[(size_t)r * gridDim.x + blockIdx.x] = bi[0];
  }
}

// embedding table [V][E] -> [E][V] (the logits kernel of the tied head before it is re-tiled).  grid = (ceil(V/32), ceil(E/32)), block (32, 8).
__global__ void k_transpose_ve(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int V, int E) {
  __shared__ bf16_t tile[32][33];
  const int v0 = blockIdx.x * 32, e0 = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int v = v0 + i, e = e0 + threadIdx.x;
    tile[i][threadIdx.x] = (v < V && e < E) ? src[(size_t)v * E + e] : (bf16_t)0;
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int e = e0 + i, v = v0 + threadIdx.x;
    if (e < E && v < V) dst[(size_t)e * V + v] = tile[threadIdx.x][i];
  }
}
