// ixt_kernels.h -- gfx950 kernels of the hot path.  Wave = 64 lanes everywhere.
//
// HBM layouts:
//   weights      bf16 row-major [K][N] (N contiguous): a wave reads 64 lanes x 16 B = 1 KiB of one k-row
//   K pages      bf16 [layer][page][kv_head][hd/8][tok][8]   (token-minor inside a d-chunk: a wave's
//                score load is tokens_per_page x 16 B contiguous per page)
//   V pages      bf16 [layer][page][kv_head][tok][hd]        (a wave's PV load is 256 B contiguous per page)
//   partials     f32  [slice][row][N]
#pragma once
#include "ixt_device.h"

// =====================================================================================================
// Canonical DOT (DenseGeneral, linears.py:188-232).
//   grid  = (ceil(N/512), nsplit, ceil(R/RB));  block = 512 threads = 8 waves.
//   wave w of the block owns chain w of each 128-k group: 16 k-rows x 512 columns, one 16 B load per lane
//   per k-row, 8 columns x RB rows of f32 accumulators per lane.  The 8 chain results of a group are added
//   in chain order through LDS (thread t owns column t).  SPLIT: every group sum is written to its own
//   partial slice (consumer adds slices in order).  !SPLIT: the block walks all groups and adds the group
//   sums in order itself (one slice).
// =====================================================================================================
struct DotArgs {
  const bf16_t* x;   // [R][K], row stride ldx
  int ldx;
  const bf16_t* w;   // [K][N]
  int K, N, R;
  float* out;        // [slices][R][N]
  int groups_per_block;  // SPLIT: groups handled by one block (blockIdx.y picks the range)
  float* amax_val = nullptr;     // EPI 1: [R][ntiles]
  int32_t* amax_idx = nullptr;   // EPI 1: [R][ntiles]
  int xmode = 0;
};

// Sum of partial slices in slice order, with the loads of up to U slices issued together (the adds stay
// sequential; padding with +0.0f is exact because a running sum that started at +0.0f is never -0.0f).
template <int U>
__device__ __forceinline__ float sum_slices(const float* p, size_t stride, int slices) {
  float t = 0.0f;
  for (int s0 = 0; s0 < slices; s0 += U) {
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = (s0 + u < slices) ? p[(size_t)(s0 + u) * stride] : 0.0f;
#pragma unroll
    for (int u = 0; u < U; ++u) t = t + v[u];
  }
  return t;
}
template <int U>
__device__ __forceinline__ float2 sum_slices2(const float* p, size_t stride, int slices) {
  float2 t = make_float2(0.0f, 0.0f);
  for (int s0 = 0; s0 < slices; s0 += U) {
    float2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = (s0 + u < slices) ? *reinterpret_cast<const float2*>(p + (size_t)(s0 + u) * stride) : make_float2(0.0f, 0.0f);
#pragma unroll
    for (int u = 0; u < U; ++u) { t.x = t.x + v[u].x; t.y = t.y + v[u].y; }
  }
  return t;
}

template <int LW> struct WVec;
template <> struct WVec<8> { typedef uint32_t type __attribute__((ext_vector_type(4))); };
template <> struct WVec<4> { typedef uint32_t type __attribute__((ext_vector_type(2))); };
template <> struct WVec<2> { typedef uint32_t type; };

template <int LW>
__device__ __forceinline__ void unpack_w(const typename WVec<LW>::type& v, float* wf) {
  if constexpr (LW == 8) {
    wf[0] = bflo(v.x); wf[1] = bfhi(v.x); wf[2] = bflo(v.y); wf[3] = bfhi(v.y);
    wf[4] = bflo(v.z); wf[5] = bfhi(v.z); wf[6] = bflo(v.w); wf[7] = bfhi(v.w);
  } else if constexpr (LW == 4) {
    wf[0] = bflo(v.x); wf[1] = bfhi(v.x); wf[2] = bflo(v.y); wf[3] = bfhi(v.y);
  } else {
    wf[0] = bflo(v); wf[1] = bfhi(v);
  }
}

// Index of activation element (row r, feature k) in the dot kernel's x layout: [row-block r/16][k/32][k%4][r%16][(k%32)/4].
// A lane (s = row-in-block, q = k%4) of the MFMA A fragment then finds its 8 values of a 32-k quarter-group in 16 contiguous bytes.
__host__ __device__ __forceinline__ size_t xidx(int r, int k, int K) {
  return (size_t)(r >> 4) * K * 16 + ((((size_t)(k >> 5) * 4 + (k & 3)) * 16 + (r & 15)) * 8 + ((k & 31) >> 2));
}

// =====================================================================================================
// Canonical DOT (DenseGeneral, linears.py:188-232), two kernels over one x layout and one operation order:
//   chain       16 consecutive k: a = 0; a = fma(x_k, w_kn, a)
//   group       8 chains (128 k) added in order from 0
//   super-group IXT_SG groups (512 k) added in order from 0        <- one f32 "slice" per super-group in split-K launches
//   total       super-groups added in order from 0
// v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate) is bit-for-bit a k-ordered fmaf chain (CDNA4 guide, "FP32-input MFMA"),
// so the matrix-core kernel and the vector-ALU kernel produce identical bits.
// =====================================================================================================
typedef float f32x4 __attribute__((ext_vector_type(4)));

// =====================================================================================================
// The same canonical DOT on the vector ALU (v_pk_fma_f32), for <= 8 activation rows per block (decode): at 8 rows the f32
// matrix pipe gives no advantage (f32 MFMA rate = vector rate, and a 16-row instruction is half empty), while this
// structure has every load of a block in flight at once and 8 waves per 128-k group.
//   grid  = (ceil(N/(64 LW)), nsplit, ceil(R/RB));  block = 512 threads = 8 waves = the 8 chains of one group.
//   wave w owns chain w: 16 k-rows x (64 LW) columns, one load per lane per k-row, LW columns x RB rows of f32 accumulators
//   per lane; x comes in as SGPR operands (scalar loads).  The 8 chain results are added in chain order through LDS.
//   SPLIT: every group sum goes to its own slice; !SPLIT: the block walks all groups and adds the group sums in order.
// =====================================================================================================
// LW = columns per lane (8: 16-B loads, tile 512 columns; 4: 8-B loads, tile 256; 2: 4-B loads, tile 128) -- a pure
// parallelisation choice for matrices with few column tiles; the numerics do not depend on it.
// EPI = 1 (only with !SPLIT): greedy-sampling epilogue (decoders.py:570 + inference_utils.py:74-75): the totals are rounded
// to bf16, optionally stored as f32 logits, and each block emits its (max, first index) pair per row for k_argmax_final.
template <int RB, int LW, bool SPLIT, int EPI, bool NT>
__global__ __launch_bounds__(512, 4) void k_dot(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, int K, int N, int R,
                                                float* __restrict__ out, int groups_per_block, float* __restrict__ amax_val,
                                                int32_t* __restrict__ amax_idx) {
  typedef typename WVec<LW>::type wvec_t;
  constexpr int RH = (LW == 8 && RB > 4) ? 4 : RB;  // rows per accumulate/exchange pass (8 rows x 8 columns would not fit 128 VGPRs)
  constexpr int TW = 64 * LW;          // tile width
  constexpr int NP = 512 / TW;         // threads per column in the exchange
  __shared__ __attribute__((aligned(16))) float red[8 * RH * TW];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int n0 = blockIdx.x * TW;
  const int nl = n0 + lane * LW;
  const bool col_ok = nl < N;
  const int r0 = blockIdx.z * RB;
  const int G = K / IXT_GL;
  int g_begin = 0, g_end = G;
  if (SPLIT) {   // one super-group per block
    g_begin = blockIdx.y * IXT_SG;
    g_end = min(G, g_begin + IXT_SG);
  }
  const int ocol = tid % TW, opart = tid / TW;
  float tot[RB], sgs[RB];   // total over super-groups, running super-group sum
#pragma unroll
  for (int r = 0; r < RB; ++r) { tot[r] = 0.0f; sgs[r] = 0.0f; }

  for (int g = g_begin; g < g_end; ++g) {
    const int kbase = g * IXT_GL + wave * IXT_CL;
    wvec_t wv[IXT_CL];
    if (col_ok) {
      const bf16_t* wp = w + (size_t)kbase * N + nl;
      // NT: weights of a decode step are read exactly once per step -> non-temporal, so they do not evict the partial
      // slices and activations the next kernels re-read from L2 / Infinity Cache
#pragma unroll
      for (int kk = 0; kk < IXT_CL; ++kk) {
        const wvec_t* pw = reinterpret_cast<const wvec_t*>(wp + (size_t)kk * N);
        if constexpr (NT) wv[kk] = __builtin_nontemporal_load(pw);
        else wv[kk] = *pw;
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < IXT_CL; ++kk) wv[kk] = wvec_t(0);
    }
#pragma unroll
    for (int hh = 0; hh < RB / RH; ++hh) {
      // ---- x for this pass: per row 4 scalar 8-byte loads from the dot x layout (xidx): for q = k%4 the wave's chain holds
      //      k = kbase+q, +4, +8, +12 as 4 consecutive bf16 (rows past R re-read row R-1; never stored)
      uint32_t xw[RH][4][2];
#pragma unroll
      for (int r = 0; r < RH; ++r) {
        const int row = min(r0 + hh * RH + r, R - 1);
        const bf16_t* xr = x + (size_t)(row >> 4) * K * 16 + ((size_t)(kbase >> 5) * 64 + (row & 15)) * 8 + ((kbase >> 4) & 1) * 4;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const uint32_t* p2 = reinterpret_cast<const uint32_t*>(xr + (size_t)qq * 128);
          xw[r][qq][0] = p2[0];
          xw[r][qq][1] = p2[1];
        }
      }
      float acc[RH][LW];
#pragma unroll
      for (int r = 0; r < RH; ++r)
#pragma unroll
        for (int j = 0; j < LW; ++j) acc[r][j] = 0.0f;
#pragma unroll
      for (int kk = 0; kk < IXT_CL; ++kk) {
        float wf[LW];
        unpack_w<LW>(wv[kk], wf);
#pragma unroll
        for (int r = 0; r < RH; ++r) {
          const uint32_t xd = xw[r][kk & 3][kk >> 3];
          const float xv = ((kk >> 2) & 1) ? bfhi(xd) : bflo(xd);
#pragma unroll
          for (int j = 0; j < LW; ++j) acc[r][j] = __builtin_fmaf(xv, wf[j], acc[r][j]);
        }
      }
      if (hh + 1 < RB / RH) {
        // make the weight registers opaque so the next pass re-unpacks them instead of keeping 128 unpacked floats live
#pragma unroll
        for (int kk = 0; kk < IXT_CL; ++kk) asm volatile("" : "+v"(wv[kk]));
      }
      // ---- add the 8 chains of the group in chain order
      if (hh > 0 || g > g_begin) __syncthreads();
#pragma unroll
      for (int r = 0; r < RH; ++r) {
        float* dst = red + ((wave * RH + r) * TW) + lane * LW;
#pragma unroll
        for (int j = 0; j < LW; ++j) dst[j] = acc[r][j];
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < RH; ++r) {
        if (RH >= NP ? (r % NP == opart) : (opart == r)) {
          float s = 0.0f;
#pragma unroll
          for (int c = 0; c < IXT_NC; ++c) s = s + red[(c * RH + r) * TW + ocol];
          const int row = r0 + hh * RH + r;
          sgs[hh * RH + r] = sgs[hh * RH + r] + s;
          if (((g + 1) % IXT_SG) == 0 || g + 1 == g_end) {   // super-group complete
            if (SPLIT) {
              if (row < R && n0 + ocol < N) out[((size_t)blockIdx.y * R + row) * N + n0 + ocol] = sgs[hh * RH + r];
            } else {
              tot[hh * RH + r] = tot[hh * RH + r] + sgs[hh * RH + r];
            }
            sgs[hh * RH + r] = 0.0f;
          }
        }
      }
    }
  }
  if (!SPLIT && EPI == 0) {
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const bool mine = (RH >= NP) ? ((r % RH) % NP == opart) : (opart == (r % RH));
      if (mine && r0 + r < R && n0 + ocol < N) out[(size_t)(r0 + r) * N + n0 + ocol] = tot[r];
    }
  }
  if (!SPLIT && EPI == 1) {
    // tile of rounded logits in LDS: red[r][TW] (RB*TW <= 8*RH*TW floats)
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const bool mine = (RH >= NP) ? ((r % RH) % NP == opart) : (opart == (r % RH));
      if (mine) red[r * TW + ocol] = (n0 + ocol < N) ? rbf(tot[r]) : -INFINITY;
    }
    __syncthreads();
    if (out != nullptr) {
      for (int i = tid; i < RB * TW; i += 512) {
        const int r = i / TW, c = i % TW;
        if (r0 + r < R && n0 + c < N) out[(size_t)(r0 + r) * N + n0 + c] = red[i];
      }
    }
    if (wave < RB && r0 + wave < R) {
      float best = -INFINITY;
      int bidx = 0x7fffffff;
#pragma unroll
      for (int c = lane; c < TW; c += 64) {
        const float v = red[wave * TW + c];
        if (v > best) { best = v; bidx = n0 + c; }   // ascending c: first index wins ties
      }
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bidx, off, 64);
        if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
      }
      if (lane == 0) {
        amax_val[(size_t)(r0 + wave) * gridDim.x + blockIdx.x] = best;
        amax_idx[(size_t)(r0 + wave) * gridDim.x + blockIdx.x] = bidx;
      }
    }
  }
}

// =====================================================================================================
// Matrix-core DOT, one memory round trip per group: block = 4 waves = one 128-k group x 128 columns; wave v owns chains
// 2v, 2v+1 (32 k): its 8 weight loads + 1 x load are all issued before the first MFMA; the 8 chain results of the group are
// exchanged through LDS and added in chain order (thread t owns row t/16, columns 8(t%16)..+7).
// grid = (ceil(N/128), nsplit, ceil(R/16)), block 256, LDS 64 KiB.
// =====================================================================================================
// NRB = 16-row blocks handled by one thread block with the SAME loaded weights (1: <= 16 rows, 2: <= 32, 4: <= 64 per
// blockIdx.z): the weights of a group are loaded once and reused from registers for every row block.
template <bool SPLIT, int EPI, bool NT, int NRB>
__global__ __launch_bounds__(256, 2) void k_dotg(const bf16_t* __restrict__ xT, const bf16_t* __restrict__ w, int K, int N, int R,
                                                 float* __restrict__ out, float* __restrict__ amax_val, int32_t* __restrict__ amax_idx,
                                                 int ntiles) {
  // (an exchange of 8 rows at a time in 32 KiB -- four blocks per CU -- measured slower: the extra barriers cost more than the occupancy buys)
  __shared__ __attribute__((aligned(16))) float red[8 * 16 * 128];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int s = lane & 15, q = lane >> 4;
  const int tile = blockIdx.x;
  const int n0 = tile * 128, nl = n0 + s * 8;
  const bool col_ok = nl < N;
  const int rz0 = blockIdx.z * NRB;                       // first 16-row block of this thread block
  const int G = K / IXT_GL;
  int g_begin = 0, g_end = G;
  if (SPLIT) { g_begin = blockIdx.y * IXT_SG; g_end = min(G, g_begin + IXT_SG); }   // one super-group per block
  // reducer role
  const int orow = tid >> 4, ocol = (tid & 15) * 8;
  float tot[NRB][8], sgs[NRB][8];   // total over super-groups, running super-group sum
#pragma unroll
  for (int b = 0; b < NRB; ++b)
#pragma unroll
    for (int j = 0; j < 8; ++j) { tot[b][j] = 0.0f; sgs[b][j] = 0.0f; }

  u32x4 wq[8];
  u32x4 xq[NRB];
#define IXT_LOAD_G(G_)                                                                                    \
  do {                                                                                                    \
    const int qg_ = (G_) * 4 + wave;   /* quarter-group = this wave's 2 chains */                         \
    _Pragma("unroll") for (int b_ = 0; b_ < NRB; ++b_)                                                    \
      xq[b_] = ((rz0 + b_) * 16 < R) ? *reinterpret_cast<const u32x4*>(xT + (size_t)(rz0 + b_) * K * 16 + (((size_t)qg_ * 4 + q) * 16 + s) * 8) : u32x4(0); \
    const bf16_t* wrow_ = w + (size_t)(qg_ * 32 + q) * N + nl;                                            \
    _Pragma("unroll") for (int t_ = 0; t_ < 8; ++t_) {                                                    \
      const u32x4* pw_ = reinterpret_cast<const u32x4*>(wrow_ + (size_t)(4 * t_) * N);                    \
      if (!col_ok) wq[t_] = u32x4(0);                                                                     \
      else if constexpr (NT) wq[t_] = __builtin_nontemporal_load(pw_);                                    \
      else wq[t_] = *pw_;                                                                                 \
    }                                                                                                     \
  } while (0)
  IXT_LOAD_G(g_begin);
  for (int g = g_begin; g < g_end; ++g) {
    u32x4 wc[8];
    u32x4 xc[NRB];
#pragma unroll
    for (int t = 0; t < 8; ++t) wc[t] = wq[t];
#pragma unroll
    for (int b = 0; b < NRB; ++b) xc[b] = xq[b];
    if (g + 1 < g_end) IXT_LOAD_G(g + 1);   // next group's loads fly during this group's MFMAs and exchanges
#pragma unroll
    for (int b = 0; b < NRB; ++b) {
      if ((rz0 + b) * 16 < R) {
        if (b > 0 || g > g_begin) __syncthreads();   // previous readers of red are done
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          f32x4 acc[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = f32x4(0.0f);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int i = c * 4 + t;
            const uint32_t xd = xc[b][i >> 1];
            const float a = (i & 1) ? bfhi(xd) : bflo(xd);
            const u32x4 wv = wc[i];
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bflo(wv.x), acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bfhi(wv.x), acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bflo(wv.y), acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bfhi(wv.y), acc[3], 0, 0, 0);
            acc[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bflo(wv.z), acc[4], 0, 0, 0);
            acc[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bfhi(wv.z), acc[5], 0, 0, 0);
            acc[6] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bflo(wv.w), acc[6], 0, 0, 0);
            acc[7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bfhi(wv.w), acc[7], 0, 0, 0);
          }
          // chain (2*wave + c) -> red[chain][row][col]
          float* dst = red + ((size_t)(wave * 2 + c) * 16) * 128 + s * 8;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* d2 = dst + (q * 4 + r) * 128;
            *reinterpret_cast<float4*>(d2) = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
            *reinterpret_cast<float4*>(d2 + 4) = make_float4(acc[4][r], acc[5][r], acc[6][r], acc[7][r]);
          }
        }
        __syncthreads();
        float sum[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) sum[j] = 0.0f;
#pragma unroll
        for (int c = 0; c < IXT_NC; ++c) {
          const float* src = red + ((size_t)c * 16 + orow) * 128 + ocol;
          const float4 a0 = *reinterpret_cast<const float4*>(src);
          const float4 a1 = *reinterpret_cast<const float4*>(src + 4);
          sum[0] = sum[0] + a0.x; sum[1] = sum[1] + a0.y; sum[2] = sum[2] + a0.z; sum[3] = sum[3] + a0.w;
          sum[4] = sum[4] + a1.x; sum[5] = sum[5] + a1.y; sum[6] = sum[6] + a1.z; sum[7] = sum[7] + a1.w;
        }
        const int row = (rz0 + b) * 16 + orow;
#pragma unroll
        for (int j = 0; j < 8; ++j) sgs[b][j] = sgs[b][j] + sum[j];
        if (((g + 1) % IXT_SG) == 0 || g + 1 == g_end) {   // super-group complete
          if (SPLIT) {
            if (row < R && n0 + ocol < N) {
              float* dst = out + ((size_t)blockIdx.y * R + row) * N + n0 + ocol;
              *reinterpret_cast<float4*>(dst) = make_float4(sgs[b][0], sgs[b][1], sgs[b][2], sgs[b][3]);
              *reinterpret_cast<float4*>(dst + 4) = make_float4(sgs[b][4], sgs[b][5], sgs[b][6], sgs[b][7]);
            }
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) tot[b][j] = tot[b][j] + sgs[b][j];
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) sgs[b][j] = 0.0f;
        }
      }
    }
  }
#undef IXT_LOAD_G
  if (!SPLIT) {
#pragma unroll
    for (int b = 0; b < NRB; ++b) {
      const int row = (rz0 + b) * 16 + orow;
      const bool ok = row < R && n0 + ocol < N;
      if (EPI == 0) {
        if (ok) {
          float* dst = out + (size_t)row * N + n0 + ocol;
          *reinterpret_cast<float4*>(dst) = make_float4(tot[b][0], tot[b][1], tot[b][2], tot[b][3]);
          *reinterpret_cast<float4*>(dst + 4) = make_float4(tot[b][4], tot[b][5], tot[b][6], tot[b][7]);
        }
      } else {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = rbf(tot[b][j]);
        if (ok && out != nullptr) {
          float* dst = out + (size_t)row * N + n0 + ocol;
          *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
        float best = -INFINITY;
        int bidx = 0x7fffffff;
        if (n0 + ocol < N) {
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (v[j] > best) { best = v[j]; bidx = n0 + ocol + j; }
        }
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {   // the 16 threads of a row are 16 consecutive lanes
          const float ov = __shfl_xor(best, off, 64);
          const int oi = __shfl_xor(bidx, off, 64);
          if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
        }
        if ((tid & 15) == 0 && row < R) {
          amax_val[(size_t)row * ntiles + tile] = best;
          amax_idx[(size_t)row * ntiles + tile] = bidx;
        }
      }
    }
  }
}

// x [R][K] row-major bf16 -> the dot kernel's x layout (op-level entry point only; the model's producers write it directly)
__global__ void k_transpose_x(const bf16_t* x, bf16_t* xT, int R, int K) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (k < K) xT[xidx(r, k, K)] = x[(size_t)r * K + k];
}

// final (max, first index) over the per-tile pairs.  grid = R, block 256.
__global__ __launch_bounds__(256) void k_argmax_final(const float* amax_val, const int32_t* amax_idx, int ntiles, int32_t* out_idx) {
  __shared__ float bv[256];
  __shared__ int bi[256];
  const int r = blockIdx.x, tid = threadIdx.x;
  float best = -INFINITY;
  int bidx = 0x7fffffff;
  for (int t = tid; t < ntiles; t += 256) {
    const float v = amax_val[(size_t)r * ntiles + t];
    const int i2 = amax_idx[(size_t)r * ntiles + t];
    if (v > best || (v == best && i2 < bidx)) { best = v; bidx = i2; }
  }
  bv[tid] = best; bi[tid] = bidx;
  __syncthreads();
  for (int s2 = 128; s2 > 0; s2 >>= 1) {
    if (tid < s2) {
      const float v = bv[tid + s2]; const int i2 = bi[tid + s2];
      if (v > bv[tid] || (v == bv[tid] && i2 < bi[tid])) { bv[tid] = v; bi[tid] = i2; }
    }
    __syncthreads();
  }
  if (tid == 0) out_idx[r] = bi[0] == 0x7fffffff ? 0 : bi[0];
}

// h_out = bf16(h_in + bf16(sum of slices))  (llama2.py:127, :157).  grid = (ceil(E/256), R), block 128, 2 columns per thread.
__global__ __launch_bounds__(128) void k_sumres(const float* part, int slices, size_t slice_stride, const bf16_t* h_in, bf16_t* h_out, int E) {
  const int c = (blockIdx.x * 128 + threadIdx.x) * 2, r = blockIdx.y;
  if (c >= E) return;
  const float2 o = sum_slices2<16>(part + (size_t)r * E + c, slice_stride, slices);
  const uint32_t hv = *reinterpret_cast<const uint32_t*>(h_in + (size_t)r * E + c);
  const bf16_t a = f2bf(bflo(hv) + rbf(o.x)), b = f2bf(bfhi(hv) + rbf(o.y));
  *reinterpret_cast<uint32_t*>(h_out + (size_t)r * E + c) = pack2(a, b);
}

// h_out = bf16(h_in + bf16(sum of slices)) AND the per-chunk sums of squares of h_out (chunk = 8 consecutive elements,
// one fma chain -- the first level of the canonical RMSNorm reduction; the consumer finishes it).
// grid = (ceil(E/128), R), block 64: thread t owns 2 columns; chunk = 4 consecutive threads.
__global__ __launch_bounds__(64) void k_sumres_chunk(const float* part, int slices, size_t slice_stride, const bf16_t* h_in, bf16_t* h_out,
                                                     float* chunk, int E) {
  const int r = blockIdx.y, t = threadIdx.x;
  const int c = blockIdx.x * 128 + t * 2;
  float h0 = 0.0f, h1 = 0.0f;
  if (c < E) {
    const float2 o = sum_slices2<16>(part + (size_t)r * E + c, slice_stride, slices);
    const uint32_t hv = *reinterpret_cast<const uint32_t*>(h_in + (size_t)r * E + c);
    h0 = rbf(bflo(hv) + rbf(o.x));
    h1 = rbf(bfhi(hv) + rbf(o.y));
    *reinterpret_cast<uint32_t*>(h_out + (size_t)r * E + c) = pack2(f2bf(h0), f2bf(h1));
  }
  // chain over the chunk's 8 elements in order: thread q of the chunk continues the chain of thread q-1
  float a = 0.0f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float prev = __shfl_up(a, 1, 64);
    if ((t & 3) == q) {
      if (q > 0) a = prev;
      a = __builtin_fmaf(h0, h0, a);
      a = __builtin_fmaf(h1, h1, a);
    }
  }
  if ((t & 3) == 3 && c < E) chunk[(size_t)r * (E / 8) + c / 8] = a;
}

// =====================================================================================================
// Row kernel: (optional) sum of partial slices + residual, then RMSNorm (normalizations.py:57-70).
//   MODE 0: h = embedding[token[r]]                         (decoders.py:493-534, embeddings.py:154)
//   MODE 1: h = bf16(h_in + bf16(sum_slices part))          (llama2.py:127, :157)
//   MODE 2: h = h_in                                        (norm only)
//   xn = bf16(bf16(h * rsqrt(mean(h^2)+eps)) * scale) when scale != nullptr.
// grid = R, block = 256, dynamic LDS = E floats + 64 floats.
// =====================================================================================================
struct RowNormArgs {
  const int32_t* tokens;
  const bf16_t* embedding;
  const float* part; int slices; size_t slice_stride;  // floats between slices
  const bf16_t* h_in;
  bf16_t* h_out;
  const bf16_t* scale;
  bf16_t* xn_out;
  int E; float eps;
  int xn_kmajor;   // 1: write xn as [row-block][E][16] (the dot kernel's x layout), 0: [R][E]
};

template <int MODE>
__global__ __launch_bounds__(256) void k_rownorm(RowNormArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* hrow = smem;            // [E]
  float* blk = smem + a.E;       // [<=64]
  const int r = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int E = a.E, nch = E / 8, nblk = (nch + 63) / 64;
  const bf16_t* src = nullptr;
  if (MODE == 0) src = a.embedding + (size_t)a.tokens[r] * E;
  if (MODE == 1 || MODE == 2) src = a.h_in + (size_t)r * E;
  for (int it = 0; it * 4 < nblk; ++it) {
    const int b = it * 4 + wave;
    const int ch = b * 64 + lane;
    float v = 0.0f;
    if (b < nblk && ch < nch) {
      const u32x4 hv = *reinterpret_cast<const u32x4*>(src + ch * 8);
      float h[8] = {bflo(hv.x), bfhi(hv.x), bflo(hv.y), bfhi(hv.y), bflo(hv.z), bfhi(hv.z), bflo(hv.w), bfhi(hv.w)};
      if (MODE == 1) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = 0.0f;
        for (int s0 = 0; s0 < a.slices; s0 += 8) {
          float4 p0[8], p1[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const float* p = a.part + (size_t)(s0 + u) * a.slice_stride + (size_t)r * E + ch * 8;
            p0[u] = (s0 + u < a.slices) ? *reinterpret_cast<const float4*>(p) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            p1[u] = (s0 + u < a.slices) ? *reinterpret_cast<const float4*>(p + 4) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            o[0] = o[0] + p0[u].x; o[1] = o[1] + p0[u].y; o[2] = o[2] + p0[u].z; o[3] = o[3] + p0[u].w;
            o[4] = o[4] + p1[u].x; o[5] = o[5] + p1[u].y; o[6] = o[6] + p1[u].z; o[7] = o[7] + p1[u].w;
          }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) h[j] = rbf(h[j] + rbf(o[j]));
      }
      if (MODE != 2 || a.h_out != nullptr) {
        if (a.h_out != nullptr) {
          u32x4 ov;
          ov.x = pack2(f2bf(h[0]), f2bf(h[1])); ov.y = pack2(f2bf(h[2]), f2bf(h[3]));
          ov.z = pack2(f2bf(h[4]), f2bf(h[5])); ov.w = pack2(f2bf(h[6]), f2bf(h[7]));
          *reinterpret_cast<u32x4*>(a.h_out + (size_t)r * E + ch * 8) = ov;
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        hrow[ch * 8 + j] = h[j];
        v = __builtin_fmaf(h[j], h[j], v);
      }
    }
    if (b < nblk) {
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) v = v + __shfl_xor(v, off, 64);
      if (lane == 0) blk[b] = v;
    }
  }
  __syncthreads();
  if (a.scale == nullptr) return;
  float total = 0.0f;
  for (int b = 0; b < nblk; ++b) total = total + blk[b];
  const float mean = total / (float)E;
  const float rs = 1.0f / sqrtf(mean + a.eps);
  for (int ch = tid; ch < nch; ch += 256) {
    const u32x4 sv = *reinterpret_cast<const u32x4*>(a.scale + ch * 8);
    const float sc[8] = {bflo(sv.x), bfhi(sv.x), bflo(sv.y), bfhi(sv.y), bflo(sv.z), bfhi(sv.z), bflo(sv.w), bfhi(sv.w)};
    bf16_t o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f2bf(rbf(hrow[ch * 8 + j] * rs) * sc[j]);
    if (a.xn_kmajor) {
#pragma unroll
      for (int j = 0; j < 8; ++j) a.xn_out[xidx(r, ch * 8 + j, E)] = o[j];
    } else {
      u32x4 ov;
      ov.x = pack2(o[0], o[1]); ov.y = pack2(o[2], o[3]); ov.z = pack2(o[4], o[5]); ov.w = pack2(o[6], o[7]);
      *reinterpret_cast<u32x4*>(a.xn_out + (size_t)r * E + ch * 8) = ov;
    }
  }
}

// =====================================================================================================
// SwiGLU from partial slices (linears.py:449-464): part is [slices][R][2F] (wi_0 columns then wi_1).
// grid = (ceil(F/256), R), block 256.
// =====================================================================================================
__global__ __launch_bounds__(256) void k_swiglu_part(const float* part, int slices, size_t slice_stride, int F,
                                                      bf16_t* out /*[R][F]*/) {
  const int f = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
  if (f >= F) return;
  const float* p = part + (size_t)r * 2 * F;
  const float a = sum_slices<16>(p + f, slice_stride, slices);
  const float b = sum_slices<16>(p + F + f, slice_stride, slices);
  out[xidx(r, f, F)] = ixt_swiglu(rbf(a), rbf(b));   // x layout of the dot kernel
}

__global__ __launch_bounds__(256) void k_swiglu(const bf16_t* a, const bf16_t* b, bf16_t* out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = ixt_swiglu(bf2f(a[i]), bf2f(b[i]));
}

// =====================================================================================================
// Attention, four steps (ref_ragged_paged_attention, paged_attention_kernel_v2.py:71-109).
// =====================================================================================================
struct AttnGeom {
  int nq, nkv, hd, P, num_pages, mpg;   // mpg = max_pages_per_group
  int Tmax;                             // scores row pitch
  int nsb_max;                          // super-blocks per row (pvpart pitch)
};

__device__ __forceinline__ size_t kpage_off(const AttnGeom& g, int page, int kvh, int dchunk, int tok) {
  return ((((size_t)page * g.nkv + kvh) * (g.hd / 8) + dchunk) * g.P + tok) * 8;
}
__device__ __forceinline__ size_t vpage_off(const AttnGeom& g, int page, int kvh, int tok) {
  return (((size_t)page * g.nkv + kvh) * g.P + tok) * g.hd;
}

// ---- step 1: slice-sum of the QKV projection, bf16 round, RoPE (embeddings.py:277-315), q -> qbuf,
//      k/v -> pages (paged_attention.py:446-471).  grid = (R, nq + 2*nkv), block 64.
struct QkvArgs {
  const float* part; int slices; size_t slice_stride; int ncols;  // ncols = (nq + 2 nkv) * hd
  const int32_t* row_slot; const int32_t* row_pos; const int32_t* row_active;
  const int32_t* page_map;
  const bf16_t* rope_sin; const bf16_t* rope_cos;   // [Tmax][hd/2]
  bf16_t* qbuf;                                     // [R][nq][hd]
  bf16_t* kpages; bf16_t* vpages;                   // this layer's pools
  AttnGeom g;
};

__global__ __launch_bounds__(64) void k_qkv_finalize(QkvArgs a) {
  const int r = blockIdx.x, hh = blockIdx.y, i = threadIdx.x;
  const AttnGeom& g = a.g;
  const int half = g.hd / 2;
  if (i >= half) return;
  const int col0 = hh * g.hd;
  const float* pq = a.part + (size_t)r * a.ncols + col0;
  float v1 = rbf(sum_slices<16>(pq + i, a.slice_stride, a.slices));
  float v2 = rbf(sum_slices<16>(pq + half + i, a.slice_stride, a.slices));
  const int pos = a.row_pos[r];
  const bool is_v = hh >= g.nq + g.nkv;
  if (!is_v) {
    const float sn = bf2f(a.rope_sin[(size_t)pos * half + i]);
    const float cs = bf2f(a.rope_cos[(size_t)pos * half + i]);
    const float o1 = rbf(rbf(v1 * cs) - rbf(v2 * sn));
    const float o2 = rbf(rbf(v2 * cs) + rbf(v1 * sn));
    v1 = o1;
    v2 = o2;
  }
  if (hh < g.nq) {
    bf16_t* q = a.qbuf + ((size_t)r * g.nq + hh) * g.hd;
    q[i] = f2bf(v1);
    q[half + i] = f2bf(v2);
    return;
  }
  if (!a.row_active[r]) return;
  const int page = a.page_map[(size_t)a.row_slot[r] * g.mpg + pos / g.P];
  const int tok = pos % g.P;
  if (!is_v) {
    const int kvh = hh - g.nq;
    a.kpages[kpage_off(g, page, kvh, i / 8, tok) + (i & 7)] = f2bf(v1);
    a.kpages[kpage_off(g, page, kvh, (half + i) / 8, tok) + ((half + i) & 7)] = f2bf(v2);
  } else {
    const int kvh = hh - g.nq - g.nkv;
    bf16_t* v = a.vpages + vpage_off(g, page, kvh, tok);
    v[i] = f2bf(v1);
    v[half + i] = f2bf(v2);
  }
}

// ---- dense k/v bf16 [R][nkv][hd] -> pages (op-level entry point).  grid = (R, nkv), block = hd.
__global__ void k_kv_write(const bf16_t* k, const bf16_t* v, const int32_t* row_slot, const int32_t* row_pos,
                           const int32_t* row_active, const int32_t* page_map, bf16_t* kpages, bf16_t* vpages, AttnGeom g) {
  const int r = blockIdx.x, kvh = blockIdx.y, d = threadIdx.x;
  if (!row_active[r] || d >= g.hd) return;
  const int pos = row_pos[r];
  const int page = page_map[(size_t)row_slot[r] * g.mpg + pos / g.P];
  const int tok = pos % g.P;
  const size_t src = ((size_t)r * g.nkv + kvh) * g.hd + d;
  kpages[kpage_off(g, page, kvh, d / 8, tok) + (d & 7)] = k[src];
  vpages[vpage_off(g, page, kvh, tok) + d] = v[src];
}

// ---- pages -> dense [len][nkv][hd] (cache export).  grid = (len, nkv), block = hd.
__global__ void k_kv_gather(const bf16_t* kpages, const bf16_t* vpages, const int32_t* page_map, int slot, AttnGeom g,
                            bf16_t* k_out, bf16_t* v_out) {
  const int j = blockIdx.x, kvh = blockIdx.y, d = threadIdx.x;
  if (d >= g.hd) return;
  const int page = page_map[(size_t)slot * g.mpg + j / g.P];
  const int tok = j % g.P;
  const size_t dst = ((size_t)j * g.nkv + kvh) * g.hd + d;
  k_out[dst] = kpages[kpage_off(g, page, kvh, d / 8, tok) + (d & 7)];
  v_out[dst] = vpages[vpage_off(g, page, kvh, tok) + d];
}

// ---- step 2: scores s_j = chain over d of q_d * k_jd (one fma chain per key, lane = key).
//      grid = (R, nkv, nsplit), block 256; a wave covers 64/P pages per pass.
struct ScoreArgs {
  const bf16_t* qbuf; const bf16_t* kpages;
  const int32_t* row_slot; const int32_t* row_pos; const int32_t* row_active; const int32_t* page_map;
  float* scores;      // [R][nq][Tmax]
  AttnGeom g;
};

template <int HD, int GQ>
__global__ __launch_bounds__(256) void k_attn_scores(ScoreArgs a) {
  __shared__ __attribute__((aligned(16))) float qs[GQ * HD];
  const AttnGeom& g = a.g;
  const int r = blockIdx.x, kvh = blockIdx.y, sp = blockIdx.z;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int kv_len = a.row_active[r] ? a.row_pos[r] + 1 : 0;
  const int pages = (kv_len + g.P - 1) / g.P;
  const int ppw = 64 / g.P;  // pages per wave pass
  const int first = (sp * 4) * ppw;
  if (first >= pages) return;
  for (int i = tid; i < GQ * HD; i += 256) qs[i] = bf2f(a.qbuf[((size_t)r * g.nq + kvh * GQ) * HD + i]);
  __syncthreads();
  const int32_t* prow = a.page_map + (size_t)a.row_slot[r] * g.mpg;
  const int stride = gridDim.z * 4;
  for (int ps = sp * 4 + wave; ps * ppw < pages; ps += stride) {
    const int pidx = ps * ppw + lane / g.P;
    const int tok = lane % g.P;
    const int j = pidx * g.P + tok;
    if (j >= kv_len) continue;
    const int page = prow[pidx];
    const bf16_t* kp = a.kpages + kpage_off(g, page, kvh, 0, tok);
    u32x4 kv[HD / 8];
#pragma unroll
    for (int dc = 0; dc < HD / 8; ++dc) kv[dc] = *reinterpret_cast<const u32x4*>(kp + (size_t)dc * g.P * 8);
    float acc[GQ];
#pragma unroll
    for (int q = 0; q < GQ; ++q) acc[q] = 0.0f;
#pragma unroll
    for (int dc = 0; dc < HD / 8; ++dc) {
      const float kf[8] = {bflo(kv[dc].x), bfhi(kv[dc].x), bflo(kv[dc].y), bfhi(kv[dc].y),
                           bflo(kv[dc].z), bfhi(kv[dc].z), bflo(kv[dc].w), bfhi(kv[dc].w)};
#pragma unroll
      for (int q = 0; q < GQ; ++q) {
        const float4 qa = *reinterpret_cast<const float4*>(qs + q * HD + dc * 8);
        const float4 qb = *reinterpret_cast<const float4*>(qs + q * HD + dc * 8 + 4);
        float s = acc[q];
        s = __builtin_fmaf(qa.x, kf[0], s); s = __builtin_fmaf(qa.y, kf[1], s);
        s = __builtin_fmaf(qa.z, kf[2], s); s = __builtin_fmaf(qa.w, kf[3], s);
        s = __builtin_fmaf(qb.x, kf[4], s); s = __builtin_fmaf(qb.y, kf[5], s);
        s = __builtin_fmaf(qb.z, kf[6], s); s = __builtin_fmaf(qb.w, kf[7], s);
        acc[q] = s;
      }
    }
#pragma unroll
    for (int q = 0; q < GQ; ++q) a.scores[((size_t)r * g.nq + kvh * GQ + q) * g.Tmax + j] = acc[q];
  }
}

// ---- steps 1+2 fused for decode rows (one new token per row): each block sums the q (and k) slices it needs, applies RoPE,
//      and computes the scores of its key range; the key of the new token is taken from LDS, never from the page another
//      block is writing.  Block z == 0 also writes the row's K and V into its page.  grid = (R, nkv, nsplit), block 256.
struct FrontArgs {
  const float* part; int slices; size_t slice_stride; int ncols;
  const int32_t* row_slot; const int32_t* row_pos; const int32_t* row_active; const int32_t* page_map;
  const bf16_t* rope_sin; const bf16_t* rope_cos;
  bf16_t* kpages; bf16_t* vpages;
  float* scores;
  AttnGeom g;
};

template <int HD, int GQ>
__global__ __launch_bounds__(256) void k_attn_front_decode(FrontArgs a) {
  __shared__ __attribute__((aligned(16))) float qs[GQ * HD];
  __shared__ __attribute__((aligned(16))) float kn[HD];
  const AttnGeom& g = a.g;
  const int r = blockIdx.x, kvh = blockIdx.y, sp = blockIdx.z;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (!a.row_active[r]) return;
  const int pos = a.row_pos[r];
  const int kv_len = pos + 1;
  const int pages = (kv_len + g.P - 1) / g.P;
  const int ppw = 64 / g.P;
  if ((sp * 4) * ppw >= pages) return;
  constexpr int half = HD / 2;
  const int32_t* prow = a.page_map + (size_t)a.row_slot[r] * g.mpg;
  const int QN = g.nq * HD, KN = g.nkv * HD;
  const int npairs = (GQ + 1 + (sp == 0 ? 1 : 0)) * half;
  for (int idx = tid; idx < npairs; idx += 256) {
    const int which = idx / half, i = idx % half;   // 0..GQ-1: q heads, GQ: k, GQ+1: v
    int col0;
    if (which < GQ) col0 = (kvh * GQ + which) * HD;
    else if (which == GQ) col0 = QN + kvh * HD;
    else col0 = QN + KN + kvh * HD;
    const float* pq = a.part + (size_t)r * a.ncols + col0;
    float v1 = rbf(sum_slices<16>(pq + i, a.slice_stride, a.slices));
    float v2 = rbf(sum_slices<16>(pq + half + i, a.slice_stride, a.slices));
    if (which <= GQ) {
      const float sn = bf2f(a.rope_sin[(size_t)pos * half + i]);
      const float cs = bf2f(a.rope_cos[(size_t)pos * half + i]);
      const float o1 = rbf(rbf(v1 * cs) - rbf(v2 * sn));
      const float o2 = rbf(rbf(v2 * cs) + rbf(v1 * sn));
      v1 = o1; v2 = o2;
    }
    if (which < GQ) {
      qs[which * HD + i] = v1; qs[which * HD + half + i] = v2;
    } else if (which == GQ) {
      kn[i] = v1; kn[half + i] = v2;
      if (sp == 0) {
        const int page = prow[pos / g.P], tok = pos % g.P;
        a.kpages[kpage_off(g, page, kvh, i / 8, tok) + (i & 7)] = f2bf(v1);
        a.kpages[kpage_off(g, page, kvh, (half + i) / 8, tok) + ((half + i) & 7)] = f2bf(v2);
      }
    } else {
      const int page = prow[pos / g.P], tok = pos % g.P;
      bf16_t* vd = a.vpages + vpage_off(g, page, kvh, tok);
      vd[i] = f2bf(v1); vd[half + i] = f2bf(v2);
    }
  }
  __syncthreads();
  const int stride = gridDim.z * 4;
  for (int ps = sp * 4 + wave; ps * ppw < pages; ps += stride) {
    const int pidx = ps * ppw + lane / g.P;
    const int tok = lane % g.P;
    const int j = pidx * g.P + tok;
    if (j >= kv_len) continue;
    float acc[GQ];
#pragma unroll
    for (int q = 0; q < GQ; ++q) acc[q] = 0.0f;
    if (j == pos) {
#pragma unroll 8
      for (int d = 0; d < HD; ++d) {
#pragma unroll
        for (int q = 0; q < GQ; ++q) acc[q] = __builtin_fmaf(qs[q * HD + d], kn[d], acc[q]);
      }
    } else {
      const int page = prow[pidx];
      const bf16_t* kp = a.kpages + kpage_off(g, page, kvh, 0, tok);
      u32x4 kv[HD / 8];
#pragma unroll
      for (int dc = 0; dc < HD / 8; ++dc) kv[dc] = *reinterpret_cast<const u32x4*>(kp + (size_t)dc * g.P * 8);
#pragma unroll
      for (int dc = 0; dc < HD / 8; ++dc) {
        const float kf[8] = {bflo(kv[dc].x), bfhi(kv[dc].x), bflo(kv[dc].y), bfhi(kv[dc].y),
                             bflo(kv[dc].z), bfhi(kv[dc].z), bflo(kv[dc].w), bfhi(kv[dc].w)};
#pragma unroll
        for (int q = 0; q < GQ; ++q) {
          const float4 qa = *reinterpret_cast<const float4*>(qs + q * HD + dc * 8);
          const float4 qb = *reinterpret_cast<const float4*>(qs + q * HD + dc * 8 + 4);
          float sacc = acc[q];
          sacc = __builtin_fmaf(qa.x, kf[0], sacc); sacc = __builtin_fmaf(qa.y, kf[1], sacc);
          sacc = __builtin_fmaf(qa.z, kf[2], sacc); sacc = __builtin_fmaf(qa.w, kf[3], sacc);
          sacc = __builtin_fmaf(qb.x, kf[4], sacc); sacc = __builtin_fmaf(qb.y, kf[5], sacc);
          sacc = __builtin_fmaf(qb.z, kf[6], sacc); sacc = __builtin_fmaf(qb.w, kf[7], sacc);
          acc[q] = sacc;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < GQ; ++q) a.scores[((size_t)r * g.nq + kvh * GQ + q) * g.Tmax + j] = acc[q];
  }
}

// ---- step 3: softmax + PV per super-block.  grid = (R, nkv, ceil(nsb_max/4)), block 256 (4 waves,
//      one super-block each).  Every block recomputes the row's max / exp / page sums / l (cheap, exact
//      in the canonical order) and the P of its own 4 super-blocks; lane = (page-in-super-block, d-chunk).
struct PvArgs {
  const float* scores; const bf16_t* vpages;
  const int32_t* row_slot; const int32_t* row_pos; const int32_t* row_active; const int32_t* page_map;
  float* pvpart;      // [R][nq][nsb_max][hd]
  AttnGeom g;
};

template <int HD, int GQ>
__global__ __launch_bounds__(256) void k_attn_pv(PvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const AttnGeom& g = a.g;
  const int r = blockIdx.x, kvh = blockIdx.y, chunk = blockIdx.z;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int kv_len = a.row_active[r] ? a.row_pos[r] + 1 : 0;
  const int pages = (kv_len + g.P - 1) / g.P;
  const int nsb = (pages + IXT_SB - 1) / IXT_SB;
  if (chunk * 4 >= nsb) return;
  float* ebuf = sm;                       // [mpg * P] scores -> exps -> (own range) normalised P
  float* psum = sm + (size_t)g.mpg * g.P; // [mpg]
  float* wred = psum + g.mpg;             // [8]
  const int32_t* prow = a.page_map + (size_t)a.row_slot[r] * g.mpg;
  constexpr int LPP = HD / 8;             // lanes per page (d-chunks)
  constexpr int PPW = 64 / LPP;           // pages a wave covers at once (4 for hd 128, 8 for hd 64)
  const int j0 = chunk * 4 * IXT_SB * g.P;  // first key of this block's range
  const int j1 = min(kv_len, j0 + 4 * IXT_SB * g.P);

  for (int q = 0; q < GQ; ++q) {
    const int h = kvh * GQ + q;
    const float* srow = a.scores + ((size_t)r * g.nq + h) * g.Tmax;
    if (q > 0) __syncthreads();
    // scores -> LDS (coalesced), max (order independent)
    float m = -INFINITY;
    for (int j = tid; j < kv_len; j += 256) {
      const float sv = srow[j];
      ebuf[j] = sv;
      m = fmaxf(m, sv);
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if (lane == 0) wred[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]));
    for (int j = tid; j < kv_len; j += 256) ebuf[j] = ixt_exp(ebuf[j] - m);
    __syncthreads();
    // page sums in key order: thread p owns logical page p
    for (int p = tid; p < pages; p += 256) {
      float sacc = 0.0f;
      const int e1 = min((p + 1) * g.P, kv_len);
      for (int j = p * g.P; j < e1; ++j) sacc = sacc + ebuf[j];
      psum[p] = sacc;
    }
    __syncthreads();
    float l = 0.0f;
    for (int p = 0; p < pages; ++p) l = l + psum[p];
    for (int j = j0 + tid; j < j1; j += 256) ebuf[j] = rbf(ebuf[j] / l);
    __syncthreads();
    // PV: a wave covers PPW pages at once = PPW/4 super-blocks
    const int tq = lane / LPP, dc = lane % LPP;
    if (wave * PPW < 4 * IXT_SB) {
      const int pg_local = wave * PPW + tq;  // 0..15: page within this block's 16-page range
      const int p = chunk * 4 * IXT_SB + pg_local;
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
      if (p < pages) {
        const int page = prow[p];
        const bf16_t* vp = a.vpages + vpage_off(g, page, kvh, 0) + dc * 8;
        const int ntok = min(g.P, kv_len - p * g.P);
        const float* pj = ebuf + (size_t)p * g.P;
        for (int t0 = 0; t0 < ntok; t0 += 8) {
          u32x4 vv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u)
            vv[u] = (t0 + u < ntok) ? *reinterpret_cast<const u32x4*>(vp + (size_t)(t0 + u) * HD) : u32x4(0);
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            if (t0 + u < ntok) {
              const float pv = pj[t0 + u];
              acc[0] = __builtin_fmaf(pv, bflo(vv[u].x), acc[0]); acc[1] = __builtin_fmaf(pv, bfhi(vv[u].x), acc[1]);
              acc[2] = __builtin_fmaf(pv, bflo(vv[u].y), acc[2]); acc[3] = __builtin_fmaf(pv, bfhi(vv[u].y), acc[3]);
              acc[4] = __builtin_fmaf(pv, bflo(vv[u].z), acc[4]); acc[5] = __builtin_fmaf(pv, bfhi(vv[u].z), acc[5]);
              acc[6] = __builtin_fmaf(pv, bflo(vv[u].w), acc[6]); acc[7] = __builtin_fmaf(pv, bfhi(vv[u].w), acc[7]);
            }
          }
        }
      }
      // super-block sum: pages 4s..4s+3 live in lane groups tq = 4s'..4s'+3 of this wave
      const int sbl = tq / IXT_SB;            // super-block within the wave
      const int base_lane = sbl * IXT_SB * LPP + dc;
      float sbs[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) sbs[j] = 0.0f;
#pragma unroll
      for (int i = 0; i < IXT_SB; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) sbs[j] = sbs[j] + __shfl(acc[j], base_lane + i * LPP, 64);
      }
      if (tq % IXT_SB == 0) {
        const int sb = chunk * 4 + (wave * PPW) / IXT_SB + sbl;
        if (sb < nsb) {
          float* dst = a.pvpart + (((size_t)r * g.nq + h) * g.nsb_max + sb) * HD + dc * 8;
          *reinterpret_cast<float4*>(dst) = make_float4(sbs[0], sbs[1], sbs[2], sbs[3]);
          *reinterpret_cast<float4*>(dst + 4) = make_float4(sbs[4], sbs[5], sbs[6], sbs[7]);
        }
      }
    }
  }
}

// ---- step 4: out = bf16(sum over super-blocks, in order).  grid = R, block 256.
__global__ __launch_bounds__(256) void k_attn_combine(const float* pvpart, const int32_t* row_pos, const int32_t* row_active,
                                                      bf16_t* out, AttnGeom g, int kmajor) {
  const int r = blockIdx.x;
  const int kv_len = row_active[r] ? row_pos[r] + 1 : 0;
  const int pages = (kv_len + g.P - 1) / g.P;
  const int nsb = (pages + IXT_SB - 1) / IXT_SB;
  const int n = g.nq * g.hd;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int h = i / g.hd, d = i % g.hd;
    const float t = sum_slices<8>(pvpart + (((size_t)r * g.nq + h) * g.nsb_max) * g.hd + d, (size_t)g.hd, nsb);
    if (kmajor) out[xidx(r, i, n)] = f2bf(t);
    else out[(size_t)r * n + i] = f2bf(t);
  }
}

// =====================================================================================================
// RoPE on dense rows (op-level).  grid = (R, nheads), block = hd/2.
// =====================================================================================================
__global__ void k_rope(bf16_t* x, const int32_t* pos, const bf16_t* rope_sin, const bf16_t* rope_cos, int nheads, int hd) {
  const int r = blockIdx.x, h = blockIdx.y, i = threadIdx.x, half = hd / 2;
  if (i >= half) return;
  bf16_t* p = x + ((size_t)r * nheads + h) * hd;
  const float x1 = bf2f(p[i]), x2 = bf2f(p[half + i]);
  const float sn = bf2f(rope_sin[(size_t)pos[r] * half + i]), cs = bf2f(rope_cos[(size_t)pos[r] * half + i]);
  p[i] = f2bf(rbf(x1 * cs) - rbf(x2 * sn));
  p[half + i] = f2bf(rbf(x2 * cs) + rbf(x1 * sn));
}

// =====================================================================================================
// Greedy sampling (inference_utils.py:74-75): logits f32 [R][V] (unrounded canonical totals) are rounded to
// bf16 (logits_dot_in_fp32 = False, decoders.py:570) in place, argmax = FIRST index of the maximum.
// grid = R, block 1024.
// =====================================================================================================
__global__ __launch_bounds__(1024) void k_round_argmax(float* logits, int V, int32_t* out_idx, float* logits_copy) {
  __shared__ float bv[1024];
  __shared__ int bi[1024];
  const int r = blockIdx.x, tid = threadIdx.x;
  float* row = logits + (size_t)r * V;
  float best = -INFINITY;
  int idx = 0x7fffffff;
  for (int n = tid; n < V; n += 1024) {
    const float v = rbf(row[n]);
    if (logits_copy) logits_copy[(size_t)r * V + n] = v;
    if (v > best || (v == best && n < idx)) { best = v; idx = n; }
  }
  bv[tid] = best; bi[tid] = idx;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if (tid < s) {
      const float v = bv[tid + s]; const int i2 = bi[tid + s];
      if (v > bv[tid] || (v == bv[tid] && i2 < bi[tid])) { bv[tid] = v; bi[tid] = i2; }
    }
    __syncthreads();
  }
  if (tid == 0) out_idx[r] = bi[0] == 0x7fffffff ? 0 : bi[0];
}

// =====================================================================================================
// Row descriptors and state update.
// =====================================================================================================
__global__ void k_rows_prefill(int32_t* row_slot, int32_t* row_pos, int32_t* row_active, int slot, int start, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { row_slot[i] = slot; row_pos[i] = start + i; row_active[i] = 1; }
}
__global__ void k_rows_decode(int32_t* row_slot, int32_t* row_pos, int32_t* row_active, int32_t* row_tok,
                              const int32_t* tokens, const int32_t* pos, const int32_t* has_active, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) { row_slot[i] = i; row_pos[i] = pos[i]; row_active[i] = has_active[i]; row_tok[i] = tokens[i]; }
}
// maxengine.py:913-935
__global__ void k_generate_epilogue(const int32_t* new_tok, int32_t* tokens, int32_t* next_pos, int32_t* generated,
                                    int32_t* result, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) {
    tokens[i] = new_tok[i];
    next_pos[i] = next_pos[i] + 1;
    const int gnew = generated[i] + 1;
    generated[i] = gnew;
    result[i * 3 + 0] = new_tok[i];
    result[i * 3 + 1] = 1;
    result[i * 3 + 2] = gnew;
  }
}

// =====================================================================================================
// PageState kernels (inference/page_manager.py), one wave each, state in HBM.
// =====================================================================================================
struct PageStateDev {
  int32_t* page_status;          // [num_pages]
  int32_t* page_map;             // [groups][mpg]
  int32_t* num_pages_used;       // [groups]
  int32_t* sequence_lengths;     // [groups]
  int32_t* active_page;          // [groups]
  int32_t* has_active_page;      // [groups] (0/1)
  int32_t* active_page_position; // [groups]
  int num_pages, groups, mpg, P;
};

// lowest index >= 1 with status 0, else -1 (page_manager.py:129-156); whole wave cooperates.
// Every access is an agent-scope (L1-bypassing) access followed by a wave-level drain, because the
// same wave re-reads words that its lane 0 has just written.
__device__ __forceinline__ void wave_drain() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

__device__ int wave_find_free(const int32_t* status, int num_pages, int lane) {
  for (int base = 0; base < num_pages; base += 64) {
    const int i = base + lane;
    const bool free_ = (i >= 1 && i < num_pages) ? (ldg_i32(status + i) == 0) : false;
    const unsigned long long mask = __ballot(free_);
    if (mask) return base + __ffsll((long long)mask) - 1;
  }
  return -1;
}
__device__ int wave_count_free(const int32_t* status, int num_pages, int lane) {
  int c = 0;
  for (int base = 0; base < num_pages; base += 64) {
    const int i = base + lane;
    const bool free_ = i < num_pages ? (ldg_i32(status + i) == 0) : false;
    c += __popcll(__ballot(free_));
  }
  return c;
}

__device__ void wave_release_group(PageStateDev s, int g, int lane) {
  const int nvalid = ldg_i32(s.num_pages_used + g);
  for (int i = lane; i < s.mpg; i += 64) {
    const int idx = ldg_i32(s.page_map + (size_t)g * s.mpg + i);
    if (i < nvalid && idx > 0) stg_i32(s.page_status + idx, 0);
  }
  if (lane == 0) {
    stg_i32(s.num_pages_used + g, 0); stg_i32(s.sequence_lengths + g, 0); stg_i32(s.active_page + g, 0);
    stg_i32(s.has_active_page + g, 0); stg_i32(s.active_page_position + g, 0);
  }
  wave_drain();
}

__global__ __launch_bounds__(64) void k_pages_reset(PageStateDev s) {
  const int lane = threadIdx.x;
  for (int i = lane; i < s.num_pages; i += 64) s.page_status[i] = (i == 0) ? 1 : 0;
  for (int i = lane; i < s.groups * s.mpg; i += 64) s.page_map[i] = 0;
  for (int i = lane; i < s.groups; i += 64) {
    s.num_pages_used[i] = 0; s.sequence_lengths[i] = 0; s.active_page[i] = 0;
    s.has_active_page[i] = 0; s.active_page_position[i] = 0;
  }
}
__global__ __launch_bounds__(64) void k_pages_release(PageStateDev s, int g) { wave_release_group(s, g, threadIdx.x); }

// release + reserve (page_manager.py:208-328); extend != 0: keep pages, grow to `length` tokens (build extension)
__global__ __launch_bounds__(64) void k_pages_reserve(PageStateDev s, int g, int length, int extend) {
  const int lane = threadIdx.x;
  if (!extend) wave_release_group(s, g, lane);
  const int need = (length + s.P - 1) / s.P;
  const int have = extend ? ldg_i32(s.num_pages_used + g) : 0;
  const int extra = need > have ? need - have : 0;
  const int nfree = wave_count_free(s.page_status, s.num_pages, lane);
  if (!(nfree >= extra && need <= s.mpg)) return;
  for (int i = have; i < need; ++i) {
    const int nf = wave_find_free(s.page_status, s.num_pages, lane);
    if (nf >= 0 && lane == 0) {
      stg_i32(s.page_status + nf, 1);
      stg_i32(s.page_map + (size_t)g * s.mpg + i, nf);
      stg_i32(s.num_pages_used + g, ldg_i32(s.num_pages_used + g) + 1);
    }
    wave_drain();
  }
  if (lane == 0) {
    stg_i32(s.sequence_lengths + g, length);
    stg_i32(s.active_page + g, ldg_i32(s.page_map + (size_t)g * s.mpg + need - 1));
    stg_i32(s.has_active_page + g, 1);
    stg_i32(s.active_page_position + g, ((length - 1) % s.P + 1) % s.P);
  }
}

// page_manager.py:331-412
__global__ __launch_bounds__(64) void k_pages_advance(PageStateDev s) {
  const int lane = threadIdx.x;
  for (int base = 0; base < s.groups; base += 64) {
    const int g = base + lane;
    bool needs = false;
    if (g < s.groups) {
      const int act = s.has_active_page[g];
      const int nl = s.sequence_lengths[g] + (act ? 1 : 0);
      s.sequence_lengths[g] = nl;
      if (act) s.active_page_position[g] = (nl - 1) % s.P;
      const int required = (nl + s.P - 1) / s.P;
      needs = act && required > s.num_pages_used[g] && required <= s.mpg;
    }
    unsigned long long mask = __ballot(needs);
    while (mask) {
      const int gl = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      const int gg = base + gl;
      const int nf = wave_find_free(s.page_status, s.num_pages, lane);
      if (nf >= 0 && lane == 0) {
        stg_i32(s.page_status + nf, 1);
        const int used = s.num_pages_used[gg];
        s.page_map[(size_t)gg * s.mpg + used] = nf;
        s.num_pages_used[gg] = used + 1;
        s.active_page[gg] = nf;
      }
      wave_drain();
    }
  }
}
__global__ void k_pages_set_active(PageStateDev s, int g, int v) { s.has_active_page[g] = v; }

// =====================================================================================================
// load-time helpers
// =====================================================================================================
// dst[r*ldd + c] = bf16(src[r*cols + c])
template <typename T>
__global__ void k_copy_convert(const T* src, bf16_t* dst, int64_t rows, int64_t cols, int64_t ldd) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  const int64_t r = i / cols, c = i % cols;
  if constexpr (sizeof(T) == 4) dst[r * ldd + c] = f2bf(src[i]);
  else dst[r * ldd + c] = src[i];
}
