// ixt_train_kernels.h -- gfx950 kernels of the training step (MaxText/train.py:255-472): forward + backward of the
// llama2 block stack, cross entropy with its custom VJP (max_utils.py:563-654), global-norm clip (maxtext_utils.py:825-846) and AdamW
// (optimizers.py:28-38).  Precision: bf16 activations and matmul operands, f32 accumulation (v_mfma_f32_16x16x32_bf16), f32 master
// weights / gradients / optimizer state -- the reference's dtype=bfloat16, weight_dtype=float32 (base.yml:86, :130).
// The tolerance of this path is 1e-4 relative on the loss against the f32 PyTorch restatement the tests hold (tests/test_gpu_train.py); it is NOT the
// bit-exact contract of the decode path, so reductions here use whatever order is fastest unless stated.
#pragma once
#include "ixt_device.h"

typedef float tf32x4 __attribute__((ext_vector_type(4)));
typedef short tbf16x8 __attribute__((ext_vector_type(8)));

// ---------------------------------------------------------------------------------------------------------------- general GEMM
// C[m][n] (+)= sum_k A(m, k) B(k, n) for bf16 operands with arbitrary element strides (either index may be the contiguous one), f32
// accumulation, two-level batch (z = z1 * nb2 + z2).  Block = 128 x 128 x 64, 4 waves as 2 x 2, wave = 64 x 64 = 4 x 4 MFMA tiles.
// An operand tile lies in LDS the way it lies in memory, so every global 16-byte vector is one 16-byte LDS store:
//   k contiguous  (TR = false): image [row][k], pitch 72 elements; a lane's fragment = two 8-byte reads
//   row contiguous (TR = true):  image [k][row], pitch 144 elements; a lane's fragment = two ds_read_b64_tr_b16 (the LDS hands a 4 x 16 block
//                                to 16 lanes column-wise: the transpose costs nothing; pitch 288 B puts the 8 k-rows a 32-lane half touches on
//                                8 disjoint bank groups)
// Both forms use ONE assignment of the 32 k of an MFMA to the lanes: lane group q (= lane / 16) holds k = 4q .. 4q+3 and 16+4q .. 16+4q+3
// of the 32-block (the sum over k does not care which lane group carries which k, as long as both operands agree) -- with it the two
// transposed reads of a half-wave cover 8 consecutive k-rows.
// MFMA roles: A-operand = B tile rows (n), B-operand = A tile rows (m), so a lane ends with 4 consecutive n of one m.
struct TGemm {
  const bf16_t* A; long sam, sak, sab1, sab2;
  const bf16_t* B; long sbk, sbn, sbb1, sbb2;
  void* C; long scm, scb1, scb2;      // C row stride (n contiguous)
  int M, N, K, nb2;
  int out_bf16;                       // 1: C is bf16, else f32
  int accumulate;                     // f32 output only: C += result
};

#define TG_LD 72      // [row][k] image: 64 k + 8 elements of padding (144-byte rows)
#define TG_LDT 144    // [k][row] image: 128 rows + 16 elements of padding (288-byte rows)
#define TG_BK 64
#define TG_TILE (64 * TG_LDT)   // elements of one operand image (the larger of 128 x 72 and 64 x 144: equal)
// One operand tile = 128 rows x 64 k = 1024 vectors of 8 elements, 4 per thread: fetched into registers (tg_fetch) while the previous
// tile is being multiplied, written to LDS afterwards (tg_store) -- the global latency runs under the MFMAs.
template <bool TR>
__device__ __forceinline__ void tg_fetch(u32x4 (&r)[4], const bf16_t* __restrict__ g, long s_row, long s_k, int row0, int k0, int nrows, int K, int tid) {
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int v = tid + it * 256;
    r[it] = u32x4(0);
    if (!TR) {
      const int rr = v >> 3, kk = (v & 7) * 8;
      if (row0 + rr < nrows && k0 + kk < K) r[it] = *reinterpret_cast<const u32x4*>(g + (long)(row0 + rr) * s_row + (k0 + kk));
    } else {   // rows are the contiguous index (s_row == 1): 8 consecutive rows at one k per load
      const int rr = (v & 15) * 8, kk = v >> 4;
      if (row0 + rr < nrows && k0 + kk < K) r[it] = *reinterpret_cast<const u32x4*>(g + (long)(k0 + kk) * s_k + (row0 + rr));
    }
  }
}
template <bool TR>
__device__ __forceinline__ void tg_store(bf16_t* __restrict__ lds, const u32x4 (&r)[4], int tid) {
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int v = tid + it * 256;
    if (!TR) *reinterpret_cast<u32x4*>(lds + (v >> 3) * TG_LD + (v & 7) * 8) = r[it];
    else *reinterpret_cast<u32x4*>(lds + (v >> 4) * TG_LDT + (v & 15) * 8) = r[it];
  }
}
// the 8 k of lane group q for rows row0 .. row0 + 15 of a tile, k-block ks (see the k assignment above)
template <bool TR>
__device__ __forceinline__ void tg_frag_issue(uint2 (&f)[2], const bf16_t* lds, int row0, int ks, int lane) {
  const int r_in = lane & 15, q = lane >> 4;
  if constexpr (!TR) {
    const bf16_t* p = lds + (row0 + r_in) * TG_LD + ks * 32 + 4 * q;
    f[0] = *reinterpret_cast<const uint2*>(p);
    f[1] = *reinterpret_cast<const uint2*>(p + 16);
  } else {
    // lane 4a + p of a 16-lane group: row (k) a of the block, columns (tile rows) 4p .. 4p+3
    const bf16_t* p = lds + (ks * 32 + 4 * q + (r_in >> 2)) * TG_LDT + row0 + 4 * (r_in & 3);
    const uint32_t addr = (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) void*)(p));
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f[0]) : "v"(addr) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f[1]) : "v"(addr), "n"(16 * TG_LDT * 2) : "memory");
  }
}
__device__ __forceinline__ tbf16x8 tg_frag(const uint2 (&f)[2]) {
  u32x4 v; v.x = f[0].x; v.y = f[0].y; v.z = f[1].x; v.w = f[1].y;
  return __builtin_bit_cast(tbf16x8, v);
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void k_tgemm(TGemm g) {
  __shared__ __attribute__((aligned(16))) bf16_t As[TG_TILE];
  __shared__ __attribute__((aligned(16))) bf16_t Bs[TG_TILE];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int z1 = blockIdx.z / g.nb2, z2 = blockIdx.z % g.nb2;
  const bf16_t* A = g.A + z1 * g.sab1 + z2 * g.sab2;
  const bf16_t* B = g.B + z1 * g.sbb1 + z2 * g.sbb2;
  const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
  const int wr = wave >> 1, wc = wave & 1;
  const int r_in = lane & 15, q = lane >> 4;
  tf32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = tf32x4(0.0f);
  u32x4 ra[4], rbv[4];
  tg_fetch<TA>(ra, A, g.sam, g.sak, m0, 0, g.M, g.K, tid);
  tg_fetch<TB>(rbv, B, g.sbn, g.sbk, n0, 0, g.N, g.K, tid);
  for (int k0 = 0; k0 < g.K; k0 += TG_BK) {
    tg_store<TA>(As, ra, tid);
    tg_store<TB>(Bs, rbv, tid);
    __syncthreads();
    if (k0 + TG_BK < g.K) {                       // the next tile's loads fly while this one is multiplied
      tg_fetch<TA>(ra, A, g.sam, g.sak, m0, k0 + TG_BK, g.M, g.K, tid);
      tg_fetch<TB>(rbv, B, g.sbn, g.sbk, n0, k0 + TG_BK, g.N, g.K, tid);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint2 fa[4][2], fb[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i) tg_frag_issue<TA>(fa[i], As, wr * 64 + i * 16, ks, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) tg_frag_issue<TB>(fb[j], Bs, wc * 64 + j * 16, ks, lane);
      // the transposed reads are issued by hand: the compiler neither counts them nor knows that their registers fill late -- the wait
      // takes the fragment registers as operands, so nothing that uses them can be scheduled ahead of it
      if constexpr (TA) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[2][0]), "+v"(fa[2][1]), "+v"(fa[3][0]), "+v"(fa[3][1]) :: "memory");
      if constexpr (TB) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[1][0]), "+v"(fb[1][1]), "+v"(fb[2][0]), "+v"(fb[2][1]), "+v"(fb[3][0]), "+v"(fb[3][1]) :: "memory");
      tbf16x8 af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = tg_frag(fa[i]);
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j] = tg_frag(fb[j]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  // D[i = n row][j = m col]: lane holds m = .. + r_in, n = .. + q * 4 + reg
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wr * 64 + i * 16 + r_in;
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wc * 64 + j * 16 + q * 4;
      if (n >= g.N) continue;
      const long off = z1 * g.scb1 + z2 * g.scb2 + (long)m * g.scm + n;
      if (g.out_bf16) {
        uint2 o;
        o.x = pack2(f2bf(acc[i][j][0]), f2bf(acc[i][j][1])); o.y = pack2(f2bf(acc[i][j][2]), f2bf(acc[i][j][3]));
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(g.C) + off) = o;
      } else {
        float4* p = reinterpret_cast<float4*>(reinterpret_cast<float*>(g.C) + off);
        float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        if (g.accumulate) { const float4 o = *p; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        *p = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------- elementwise / rows
__global__ void k_t_cast_bf16(const float* __restrict__ src, bf16_t* __restrict__ dst, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = f2bf(src[i]);
}

__device__ __forceinline__ float t_block_sum(float v, float* red) {   // 256 threads
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// RMSNorm forward (normalizations.py:57-70): y = bf16(bf16(x r) scale), r = rsqrt(mean(x^2) + eps) saved per row.  grid = T, block 256.
__global__ __launch_bounds__(256) void k_t_rmsnorm_fwd(const bf16_t* __restrict__ x, const bf16_t* __restrict__ scale, bf16_t* __restrict__ y,
                                                       float* __restrict__ rstd, int E, float eps) {
  __shared__ float red[4];
  const long row = blockIdx.x;
  float s = 0.0f;
  for (int e = threadIdx.x; e < E; e += 256) { const float v = bf2f(x[row * E + e]); s += v * v; }
  const float tot = t_block_sum(s, red);
  const float r = 1.0f / sqrtf(tot / (float)E + eps);
  if (threadIdx.x == 0) rstd[row] = r;
  for (int e = threadIdx.x; e < E; e += 256) y[row * E + e] = f2bf(rbf(bf2f(x[row * E + e]) * r) * bf2f(scale[e]));
}
// backward: dy f32 [T][E] (the sum of the consumers' input gradients); dx (f32) is ADDED to dh (the residual-stream gradient);
// dscale: a block walks TN_RPB rows and keeps its features' partial sums in registers -- one f32 atomic per feature and BLOCK (one per feature
// and row took 466 us per launch at 16384 x 2048: 2048 addresses under 33 M atomics).  grid = ceil(T / TN_RPB), block 256, E <= 256 TN_FPT.
#define TN_RPB 16
#define TN_FPT 16
__global__ __launch_bounds__(256) void k_t_rmsnorm_bwd(const float* __restrict__ dy, const bf16_t* __restrict__ x, const bf16_t* __restrict__ scale,
                                                       const float* __restrict__ rstd, float* __restrict__ dh, float* __restrict__ dscale, int E, long T) {
  __shared__ float red[4];
  float ds[TN_FPT];
#pragma unroll
  for (int i = 0; i < TN_FPT; ++i) ds[i] = 0.0f;
  const long row0 = (long)blockIdx.x * TN_RPB;
  for (long row = row0; row < row0 + TN_RPB && row < T; ++row) {
    const float r = rstd[row];
    float s = 0.0f;
    for (int e = threadIdx.x; e < E; e += 256) {
      const float xh = bf2f(x[row * E + e]) * r;
      s += dy[row * E + e] * bf2f(scale[e]) * xh;
    }
    const float mean = t_block_sum(s, red) / (float)E;
#pragma unroll
    for (int i = 0; i < TN_FPT; ++i) {
      const int e = threadIdx.x + i * 256;
      if (e < E) {
        const float xh = bf2f(x[row * E + e]) * r, gy = dy[row * E + e];
        dh[row * E + e] += r * (gy * bf2f(scale[e]) - xh * mean);
        ds[i] += gy * xh;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < TN_FPT; ++i) {
    const int e = threadIdx.x + i * 256;
    if (e < E) atomicAdd(dscale + e, ds[i]);
  }
}
// out[i] = parts[0][i] + parts[1][i] + ... (split-K partial products, in order)
__global__ void k_t_sum_parts(const float* __restrict__ parts, int nparts, long n, float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float t = parts[i];
  for (int p = 1; p < nparts; ++p) t += parts[(long)p * n + i];
  out[i] = t;
}

// RoPE (embeddings.py:277-315) on x bf16 [T][H][D] with per-token positions; INV: the transposed rotation (backward).  grid = (T, H), block D/2.
template <bool INV>
__global__ void k_t_rope(bf16_t* __restrict__ x, const int32_t* __restrict__ pos, const bf16_t* __restrict__ sn, const bf16_t* __restrict__ cs, int H, int D) {
  const long t = blockIdx.x;
  const int h = blockIdx.y, i = threadIdx.x, half = D / 2;
  if (i >= half) return;
  bf16_t* p = x + (t * H + h) * D;
  const float x1 = bf2f(p[i]), x2 = bf2f(p[half + i]);
  const float s = bf2f(sn[(long)pos[t] * half + i]), c = bf2f(cs[(long)pos[t] * half + i]);
  if (!INV) { p[i] = f2bf(rbf(x1 * c) - rbf(x2 * s)); p[half + i] = f2bf(rbf(x2 * c) + rbf(x1 * s)); }
  else { p[i] = f2bf(x1 * c + x2 * s); p[half + i] = f2bf(x2 * c - x1 * s); }
}

// Attention softmax over scores f32 [BH][S][S] with the train-mode mask (attentions.py:586-618: same segment AND col <= row), P bf16 out.
// grid = (S, BH), block 256; one row per block.
__global__ __launch_bounds__(256) void k_t_softmax_fwd(const float* __restrict__ sc, const int32_t* __restrict__ seg, bf16_t* __restrict__ P, int S, int H) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  const long bh = blockIdx.y, b = bh / H;
  const float* s = sc + (bh * S + row) * S;
  bf16_t* p = P + (bh * S + row) * S;
  const int myseg = seg[b * S + row];
  float m = -INFINITY;
  for (int j = threadIdx.x; j <= row; j += 256)
    if (seg[b * S + j] == myseg) m = fmaxf(m, s[j]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float l = 0.0f;
  for (int j = threadIdx.x; j <= row; j += 256)
    if (seg[b * S + j] == myseg) l += ixt_exp(s[j] - m);
  l = t_block_sum(l, red);
  for (int j = threadIdx.x; j < S; j += 256) {
    const bool on = j <= row && seg[b * S + j] == myseg;
    p[j] = on ? f2bf(ixt_exp(s[j] - m) / l) : (bf16_t)0;
  }
}
// The same row in ONE pass over memory (S <= 256 TS_MAXV): a thread's scores and segment tests stay in registers, every exponential is taken once.
// Same values as the three-pass form bit for bit (same per-thread order of the partial sums); 1.70 -> see DESIGN.md section 9.
#define TS_MAXV 16
__global__ __launch_bounds__(256) void k_t_softmax_fwd_reg(const float* __restrict__ sc, const int32_t* __restrict__ seg, bf16_t* __restrict__ P, int S, int H) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  const long bh = blockIdx.y, b = bh / H;
  const float* s = sc + (bh * S + row) * S;
  bf16_t* p = P + (bh * S + row) * S;
  const int myseg = seg[b * S + row];
  float v[TS_MAXV];
  unsigned onmask = 0;
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < TS_MAXV; ++i) {
    const int j = threadIdx.x + i * 256;
    const bool on = j <= row && seg[b * S + min(j, S - 1)] == myseg;     // (row < S: j <= row is inside the row)
    v[i] = on ? s[j] : -INFINITY;
    if (on) { onmask |= 1u << i; m = fmaxf(m, v[i]); }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float l = 0.0f;
#pragma unroll
  for (int i = 0; i < TS_MAXV; ++i) {
    if ((onmask >> i) & 1u) { v[i] = ixt_exp(v[i] - m); l += v[i]; } else v[i] = 0.0f;
  }
  l = t_block_sum(l, red);
#pragma unroll
  for (int i = 0; i < TS_MAXV; ++i) {
    const int j = threadIdx.x + i * 256;
    if (j < S) p[j] = ((onmask >> i) & 1u) ? f2bf(v[i] / l) : (bf16_t)0;
  }
}
// dS = P o (dP - sum_j dP P), bf16 out (in place of nothing: separate buffer).  grid = (S, BH), block 256.
__global__ __launch_bounds__(256) void k_t_softmax_bwd(const float* __restrict__ dP, const bf16_t* __restrict__ P, bf16_t* __restrict__ dS, int S) {
  __shared__ float red[4];
  const long base = ((long)blockIdx.y * S + blockIdx.x) * S;
  float s = 0.0f;
  for (int j = threadIdx.x; j < S; j += 256) s += dP[base + j] * bf2f(P[base + j]);
  const float dot = t_block_sum(s, red);
  for (int j = threadIdx.x; j < S; j += 256) { const float p = bf2f(P[base + j]); dS[base + j] = f2bf(p * (dP[base + j] - dot)); }
}
// one pass (S <= 256 TS_MAXV); columns beyond the row hold P = 0 (the causal mask): their dP is not read
__global__ __launch_bounds__(256) void k_t_softmax_bwd_reg(const float* __restrict__ dP, const bf16_t* __restrict__ P, bf16_t* __restrict__ dS, int S) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  const long base = ((long)blockIdx.y * S + row) * S;
  float g[TS_MAXV], pv[TS_MAXV];
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < TS_MAXV; ++i) {
    const int j = threadIdx.x + i * 256;
    const bool in = j <= row;
    pv[i] = in ? bf2f(P[base + j]) : 0.0f;
    g[i] = in ? dP[base + j] : 0.0f;
    s += g[i] * pv[i];
  }
  const float dot = t_block_sum(s, red);
#pragma unroll
  for (int i = 0; i < TS_MAXV; ++i) {
    const int j = threadIdx.x + i * 256;
    if (j < S) dS[base + j] = f2bf(pv[i] * (g[i] - dot));
  }
}

// SwiGLU (linears.py:449-464) forward: act = bf16(bf16(a sg) b), sg = bf16(sigmoid(a)); backward through the unrounded formula.
__global__ void k_t_swiglu_fwd(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ act, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) act[i] = ixt_swiglu(bf2f(a[i]), bf2f(b[i]));
}
__global__ void k_t_swiglu_bwd(const bf16_t* __restrict__ dact, const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ da,
                               bf16_t* __restrict__ db, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float av = bf2f(a[i]), bv = bf2f(b[i]), g = bf2f(dact[i]);
  const float sg = 1.0f / (1.0f + ixt_exp(-av));
  da[i] = f2bf(g * bv * sg * (1.0f + av * (1.0f - sg)));
  db[i] = f2bf(g * av * sg);
}

// residual stream: h_out = bf16(h + y) (llama2.py:127, :157); gradient helpers
__global__ void k_t_add_bf16(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = f2bf(bf2f(a[i]) + bf2f(b[i]));
}
__global__ void k_t_f32_to_bf16(const float* __restrict__ a, bf16_t* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = f2bf(a[i]);
}
__global__ void k_t_bf16_to_f32(const bf16_t* __restrict__ a, float* __restrict__ out, long n, int add) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (add ? out[i] : 0.0f) + bf2f(a[i]);
}

// embedding: forward gather (embeddings.py:154) = k_embed of ixt_kernels.h; backward scatter-add of dh f32 [T][E] into dE f32 [V][E]
__global__ void k_t_embed_bwd(const float* __restrict__ dh, const int32_t* __restrict__ ids, float* __restrict__ dE, int E, int V) {
  const long t = blockIdx.y;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  int id = ids[t];
  id = id < 0 ? 0 : (id >= V ? V - 1 : id);
  atomicAdd(dE + (long)id * E + e, dh[t * E + e]);
}

// cross entropy with the reference's VJP (max_utils.py:563-654, z_loss = 0) on a chunk of f32 logits [Tc][V] (already bf16-rounded values,
// decoders.py:570,586): loss_row = log(sum exp(l - max)) + max - l[target]; dlogits = (softmax - onehot) * weight, bf16.  grid = Tc, block 256.
__global__ __launch_bounds__(256) void k_t_xent(const float* __restrict__ logits, const int32_t* __restrict__ targets, const int32_t* __restrict__ tseg,
                                                bf16_t* __restrict__ dlogits, float* __restrict__ row_loss, int V) {
  __shared__ float red[4];
  const long row = blockIdx.x;
  const float* l = logits + row * V;
  float m = -INFINITY;
  for (int j = threadIdx.x; j < V; j += 256) m = fmaxf(m, rbf(l[j]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float s = 0.0f;
  for (int j = threadIdx.x; j < V; j += 256) s += ixt_exp(rbf(l[j]) - m);
  s = t_block_sum(s, red);
  const float w = tseg[row] != 0 ? 1.0f : 0.0f;
  const int tg = targets[row];
  if (threadIdx.x == 0) row_loss[row] = w * (logf(s) + m - rbf(l[tg]));
  const float inv = w / s;
  for (int j = threadIdx.x; j < V; j += 256) dlogits[row * V + j] = f2bf(ixt_exp(rbf(l[j]) - m) * inv - (j == tg ? w : 0.0f));
}

// out[i] = 1 where s[i] != 0 (the weight of a target position, train.py:308-309)
__global__ void k_t_nonzero(const int32_t* __restrict__ s, float* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = s[i] != 0 ? 1.0f : 0.0f;
}

// fixed-order sum of n floats by one block (deterministic), optionally of squares; out[0] (+)= sum
__global__ __launch_bounds__(256) void k_t_reduce(const float* __restrict__ x, long n, int squares, float* __restrict__ out, int accumulate) {
  __shared__ float red[4];
  double s = 0.0;
  for (long i = threadIdx.x; i < n; i += 256) { const float v = x[i]; s += squares ? (double)v * v : (double)v; }
  float f = (float)s;
  f = t_block_sum(f, red);
  if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.0f) + f;
}
// sum of squares of a large buffer: one partial per block, then k_t_reduce over the partials.  grid = nblk, block 256.
__global__ __launch_bounds__(256) void k_t_sumsq_partial(const float* __restrict__ x, long n, float* __restrict__ part) {
  __shared__ float red[4];
  float s = 0.0f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { const float v = x[i]; s += v * v; }
  s = t_block_sum(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// optax.clip_by_global_norm + optax.adamw (optimizers.py:28-38): g <- g * gscale (1 / total_weights and the clip factor), mu, nu, bias
// correction with `count` (1 at the first step), decoupled weight decay, p -= lr * update; the bf16 copy of the parameter is refreshed.
struct AdamArgs { float lr, b1, b2, eps, eps_root, wd, bc1, bc2; };
__global__ void k_t_adamw(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ mu, float* __restrict__ nu, bf16_t* __restrict__ pb,
                          const float* __restrict__ gscale, long n, AdamArgs a) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gv = g[i] * gscale[0];
  const float m = a.b1 * mu[i] + (1.0f - a.b1) * gv;
  const float v = a.b2 * nu[i] + (1.0f - a.b2) * gv * gv;
  mu[i] = m; nu[i] = v;
  const float upd = (m / a.bc1) / (sqrtf(v / a.bc2 + a.eps_root) + a.eps) + a.wd * p[i];
  const float np_ = p[i] - a.lr * upd;
  p[i] = np_;
  pb[i] = f2bf(np_);
}
// gscale[0] = (1 / denom) * min(1, clip / (norm / denom)) with norm = sqrt(sumsq[0]); norms_out = {raw grad norm, clipped grad norm}
__global__ void k_t_clip_scale(const float* __restrict__ sumsq, float denom, float clip, float* __restrict__ gscale, float* __restrict__ norms_out) {
  const float raw = sqrtf(sumsq[0]) / denom;
  const float f = (clip > 0.0f && raw >= clip) ? clip / raw : 1.0f;
  gscale[0] = f / denom;
  norms_out[0] = raw;
  norms_out[1] = raw * f;
}
