// ixt_fp8.h -- the projections of a decoder layer with FP8 weights on the gfx950 fp8 matrix cores (BASELINE.json configs[3]: "fp8 MFMA
// weights"; included by ixt_api.hip).
//
// The reference has no fp8 inference behaviour to be equal to: `quantization: fp8 / nanoo_fp8` installs AQT hooks for TRAINING
// (layers/quantizations.py:289-298) and decode refuses them (decode.py:127-128).  This build therefore DEFINES the scheme ("IXT-FP8-v1",
// DESIGN.md section 3b) and states its error against the bf16 path instead of claiming bit equality:
//   weights      W[k][n] (bf16 values) -> e4m3 (OCP: 4 exponent bits, 3 mantissa bits, max 448, no infinities) with one f32 scale per output
//                column: s_n = amax_k |W[k][n]| / 448 (1 for an all-zero column), q = e4m3_rne(W / s_n)
//   activations  x[r][k] (bf16 values) -> e4m3 with one f32 scale per row and 512-k block (the canonical dot's chain):
//                a_rc = amax over the block / 448 (1 for an all-zero block), q = e4m3_rne(x / a_rc)
//   dot          chain c: v_mfma_f32_16x16x32_fp8_fp8 over the block's k into one f32 accumulator; t = sum_c acc_c * (a_rc * s_n), c ascending
// Everything after the dot (rounding to bf16, RoPE, norms, gates, residuals, attention) is the bf16 path's, unchanged: the dot kernels here
// write the same f32 slices [slice][row][N] the split-K bf16 kernels write, with the scales already applied.
//
// Fragment layout (both operands): 16 rows (columns) x 64 k = 1 KiB, lane (n = lane % 16, kq = lane / 16) holds 16 bytes:
//   bytes 0..7 = k 64 s + 8 kq + 0..7 (operand of the first MFMA of the pair), bytes 8..15 = k 64 s + 32 + 8 kq + 0..7 (the second).
// A 64-k fragment is one LDS-DMA / one 16-byte load per lane, exactly like a 32-k bf16 fragment, so the ring GEMM moves data as before.
#pragma once

__host__ __device__ __forceinline__ size_t f8idx(int r, int k, int K) {   // byte index of element (r, k) of an operand with K columns
  return (size_t)(r >> 4) * K * 16 + ((size_t)(k >> 6) * 64 + ((k & 31) >> 3) * 16 + (r & 15)) * 16 + ((k & 63) >> 5) * 8 + (k & 7);
}

// two f32 -> two e4m3 (round to nearest even, saturating at +-448): the pair lands in byte pair `hi` of `old`
__device__ __forceinline__ uint32_t f8pack(float a, float b, uint32_t old, bool hi) {
  a = fminf(fmaxf(a, -448.0f), 448.0f);
  b = fminf(fmaxf(b, -448.0f), 448.0f);
  return hi ? (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, (int)old, true) : (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, (int)old, false);
}

// ---- weights: bf16 kernel in the dot kernels' fragment order (widx, final column order) -> e4m3 fragments + one scale per column.
//      grid = N (one block per column), block 256.
__global__ __launch_bounds__(256) void k_f8_quant_w(const bf16_t* __restrict__ w /*widx*/, int K, int N, uint8_t* __restrict__ w8, float* __restrict__ wscale) {
  __shared__ float red[4];
  const int n = blockIdx.x, tid = threadIdx.x;
  float m = 0.0f;
  for (int k = tid; k < K; k += 256) m = fmaxf(m, fabsf(bf2f(w[widx(k, n, K)])));
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float s = m > 0.0f ? m / 448.0f : 1.0f;
  if (tid == 0) wscale[n] = s;
  for (int k = tid * 2; k < K; k += 512) {     // pairs (k, k + 1) are neighbouring bytes
    const uint32_t p = f8pack(bf2f(w[widx(k, n, K)]) / s, bf2f(w[widx(k + 1, n, K)]) / s, 0, false);
    *reinterpret_cast<uint16_t*>(w8 + f8idx(n, k, K)) = (uint16_t)(p & 0xffffu);
  }
}

// ---- activations: bf16 rows in the dot kernels' x layout (xidx) -> e4m3 fragments + one scale per (row, 512-k block).
//      grid = (chains, row blocks), block 64: lane (r = lane % 16, kq = lane / 16) holds its row's 8 k of every 32-k step of the block.
__global__ __launch_bounds__(64) void k_f8_quant_x(const bf16_t* __restrict__ x /*xidx*/, int K, int R, uint8_t* __restrict__ x8, float* __restrict__ xscale /*[rows16][chains]*/) {
  const int c = blockIdx.x, rb = blockIdx.y, lane = threadIdx.x;
  const int nsteps = K >> 5, chains = gridDim.x;
  const int s0 = c * 16, cs = min(16, nsteps - s0);
  const u32x4* src = reinterpret_cast<const u32x4*>(x) + ((size_t)rb * nsteps + s0) * 64 + lane;
  u32x4 v[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) v[s] = s < cs ? src[(size_t)s * 64] : u32x4(0);
  float m = 0.0f;
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    m = fmaxf(m, fmaxf(fmaxf(fabsf(bflo(v[s].x)), fabsf(bfhi(v[s].x))), fmaxf(fabsf(bflo(v[s].y)), fabsf(bfhi(v[s].y)))));
    m = fmaxf(m, fmaxf(fmaxf(fabsf(bflo(v[s].z)), fabsf(bfhi(v[s].z))), fmaxf(fabsf(bflo(v[s].w)), fabsf(bfhi(v[s].w)))));
  }
  m = fmaxf(m, __shfl_xor(m, 16, 64));
  m = fmaxf(m, __shfl_xor(m, 32, 64));           // over the four lanes that hold this row
  if (!(m <= 3.0e38f)) m = 3.0e38f;              // a row with Inf / NaN: keep the scale finite (the values saturate)
  const float a = m > 0.0f ? m / 448.0f : 1.0f;
  const int row = rb * 16 + (lane & 15);
  if ((lane >> 4) == 0) xscale[(size_t)row * chains + c] = a;
  u32x4* dst = reinterpret_cast<u32x4*>(x8) + ((size_t)rb * (K >> 6) + (size_t)c * 8) * 64 + lane;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (2 * j < cs) {
      u32x4 o;
      const u32x4 p = v[2 * j], q = v[2 * j + 1];
      o.x = f8pack(bflo(p.x) / a, bfhi(p.x) / a, 0, false); o.x = f8pack(bflo(p.y) / a, bfhi(p.y) / a, o.x, true);
      o.y = f8pack(bflo(p.z) / a, bfhi(p.z) / a, 0, false); o.y = f8pack(bflo(p.w) / a, bfhi(p.w) / a, o.y, true);
      o.z = f8pack(bflo(q.x) / a, bfhi(q.x) / a, 0, false); o.z = f8pack(bflo(q.y) / a, bfhi(q.y) / a, o.z, true);
      o.w = f8pack(bflo(q.z) / a, bfhi(q.z) / a, 0, false); o.w = f8pack(bflo(q.w) / a, bfhi(q.w) / a, o.w, true);
      dst[(size_t)j * 64] = o;
    }
  }
}

// ---- the dot below 128 rows: k_dotb's split-K form on e4m3 fragments.  grid = (tile groups x row-block groups, chains), block 256 (4 waves =
//      4 adjacent 16-column tiles), LDS = NRB x 8 KiB.  Every wave issues the 8 weight fragments of its chain (8 KiB in flight per wave), the block
//      copies the chain's activation fragments to LDS once; 16 x NRB MFMAs; the accumulator times (a_rc s_n) is slice c.
//      Routed experts (rb_exp != nullptr; NRB = 1), as in k_dotb: row block rb belongs to expert rb_exp[rb], whose e4m3 kernel starts wstride 16-byte
//      units after w8 and whose column scales start sstride floats after wscale; a row block of no expert leaves at once.
template <int NRB>
__global__ __launch_bounds__(256, 2) void k_dotb8(const u32x4* __restrict__ x8, const u32x4* __restrict__ w8, const float* __restrict__ xscale,
                                                  const float* __restrict__ wscale, int K, int N, int R, float* __restrict__ out, int nwaves, int zblocks,
                                                  const int32_t* __restrict__ rb_exp = nullptr, size_t wstride = 0, size_t sstride = 0) {
  extern __shared__ __attribute__((aligned(16))) u32x4 xs[];   // [NRB][8 fragments][64 lanes]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bx = blockIdx.x, c = blockIdx.y;
  const int tile = ((bx / (8 * zblocks)) * 8 + (bx & 7)) * 4 + wave;
  const int nfr = K >> 6, nchains = gridDim.y;                 // 64-k fragments per row block / tile
  const int nrbk = (R + 15) >> 4;
  const int rb0 = ((bx >> 3) % zblocks) * NRB;
  const int f0 = c * 8, cf = min(8, nfr - f0);                 // (K % 64 == 0: a last chain of 256 k has 4 fragments)
  const int r_in = lane & 15, cq = lane >> 4;
  const bool tile_ok = tile < nwaves;
  if (rb_exp != nullptr) {
    const int ex = rb_exp[rb0];
    if (ex < 0) return;                                        // the whole block: no barrier is skipped
    w8 += (size_t)ex * wstride;
    wscale += (size_t)ex * sstride;
  }
  // activation fragments of this chain first (L2), then the weights (HBM): loads return in order
  u32x4 xq[NRB * 2];
#pragma unroll
  for (int j = 0; j < NRB * 2; ++j) {
    const int e = tid + j * 256, b = e / (cf * 64), rem = e - b * (cf * 64);
    xq[j] = (e < NRB * cf * 64 && rb0 + b < nrbk) ? x8[((size_t)(rb0 + b) * nfr + f0) * 64 + rem] : u32x4(0);
  }
  const u32x4* wbase = w8 + ((size_t)(tile_ok ? tile : 0) * nfr + f0) * 64 + lane;
  u32x4 wv[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) wv[s] = __builtin_nontemporal_load(wbase + (size_t)min(s, cf - 1) * 64);
#pragma unroll
  for (int j = 0; j < NRB * 2; ++j) {
    const int e = tid + j * 256, b = e / (cf * 64), rem = e - b * (cf * 64);
    if (e < NRB * cf * 64) xs[b * 512 + rem] = xq[j];
  }
  __syncthreads();
  f32x4 acc[NRB];
#pragma unroll
  for (int b = 0; b < NRB; ++b) acc[b] = f32x4(0.0f);
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    if (s < cf) {
      const long wlo = (long)(((unsigned long long)wv[s].y << 32) | wv[s].x), whi = (long)(((unsigned long long)wv[s].w << 32) | wv[s].z);
#pragma unroll
      for (int b = 0; b < NRB; ++b) {
        const u32x4 xv = xs[b * 512 + s * 64 + lane];
        const long xlo = (long)(((unsigned long long)xv.y << 32) | xv.x), xhi = (long)(((unsigned long long)xv.w << 32) | xv.z);
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wlo, xlo, acc[b], 0, 0, 0);
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(whi, xhi, acc[b], 0, 0, 0);
      }
    }
  }
  if (!tile_ok) return;
  const float4 sn = *reinterpret_cast<const float4*>(wscale + tile * 16 + cq * 4);
#pragma unroll
  for (int b = 0; b < NRB; ++b) {
    const int row = (rb0 + b) * 16 + r_in;
    if (row < R) {
      const float a = xscale[(size_t)row * nchains + c];
      *reinterpret_cast<float4*>(out + ((size_t)c * R + row) * N + tile * 16 + cq * 4) =
          make_float4(acc[b][0] * (a * sn.x), acc[b][1] * (a * sn.y), acc[b][2] * (a * sn.z), acc[b][3] * (a * sn.w));
    }
  }
}

// f32 -> e4m3 -> f32 (tests: the conversion against the numpy restatement of OCP e4m3 rounding)
__global__ void k_f8_roundtrip(const float* in, float* out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t p = f8pack(in[i], 0.0f, 0, false);
  out[i] = __builtin_amdgcn_cvt_f32_fp8((int)p, 0);
}
