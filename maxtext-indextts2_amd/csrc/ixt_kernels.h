// ixt_kernels.h -- gfx950 kernels of the hot path.  Wave = 64 lanes everywhere.
//
// HBM layouts:
//   weights      bf16, MFMA-fragment tiled [N/16][K/32][(k%32)/8][n%16][k%8]: a wave reads 1 KiB per k-step, contiguous along k
//   K pages      bf16 [layer][page][kv_head][hd/8][tok][8]   (token-minor inside a d-chunk: a wave's
//                score load is tokens_per_page x 16 B contiguous per page)
//   V pages      bf16 [layer][page][kv_head][tok][hd]        (a wave's PV load is 256 B contiguous per page)
//   partials     f32  [slice][row][N]
#pragma once
#include "ixt_device.h"

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

// =====================================================================================================
// Canonical DOT (DenseGeneral, linears.py:188-232) on the bf16 matrix cores.
//   One v_mfma_f32_16x16x32_bf16 accumulates, per output element, four steps of 8 consecutive k in ascending order; a step
//   is an aligned, truncating integer add (measured on MI355X and restated bit for bit by the CPU checker used in tests,
//   orc_mfma_step8).  The canonical order is therefore fixed by the instruction sequence:
//     chain c : acc = 0; MFMAs over k in [512c, min(K, 512c+512)), ascending, all into the same accumulator
//     total   : t = 0; t = t + chain(c), c ascending (plain f32 adds)   <- one f32 "slice" per chain in split-K launches
//   Which wave runs which chain, how many rows share a weight fragment, split-K or not: free, the bits do not change.
//
// Fragment layouts (both operands are stored in HBM exactly as a lane reads them: 16 contiguous bytes per lane, 1 KiB per
// wave-instruction, consecutive k-steps of one tile contiguous):
//   activations x(r, k): [r/16][k/32][(k%32)/8][r%16][k%8]      (xidx; written by the producer kernels)
//   weights     w(k, n): [n/16][k/32][(k%32)/8][n%16][k%8]      (widx; written once by ixt_load_weight)
// MFMA roles: A = weights (A row i = column n of the layer), B = activations (B column j = activation row r), so a lane
// ends up with 4 consecutive output columns of one row: one 16-byte store.
// =====================================================================================================
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__host__ __device__ __forceinline__ size_t xidx(int r, int k, int K) {
  return (size_t)(r >> 4) * K * 16 + (((size_t)(k >> 5) * 4 + ((k & 31) >> 3)) * 16 + (r & 15)) * 8 + (k & 7);
}
__host__ __device__ __forceinline__ size_t widx(int k, int n, int K) { return xidx(n, k, K); }
// Ring layout of a weight matrix (the LDS-ring GEMM's operand order, ixt_gemm.h): [n/128][k/64][(n%128)/16][(k%64)/32] fragments of 1 KiB,
// i.e. the 16 KiB a 128-column block needs for one 64-k chunk are contiguous and a block reads its whole operand as ONE sequential
// stream (DRAM row locality); columns are padded to a multiple of 128.
__host__ __device__ __forceinline__ size_t r8frag(int tile, int step, int nsteps) {
  return ((size_t)(tile >> 3) * (nsteps >> 1) + (step >> 1)) * 16 + (tile & 7) * 2 + (step & 1);
}
__host__ __device__ __forceinline__ size_t r8idx(int k, int n, int K) {
  return r8frag(n >> 4, k >> 5, K >> 5) * 512 + ((size_t)((k & 31) >> 3) * 16 + (n & 15)) * 8 + (k & 7);
}
// Column of feature i (0..hd-1) of a Q or K head inside the head's hd columns of the fused QKV kernel: 16-column tile j holds
// features 8j..8j+7 and half+8j..half+8j+7, so the two RoPE partners (i, i + half) of a feature share a tile (ixt_gemm.h, GEPI_QKV).
__host__ __device__ __forceinline__ int qk_col(int i, int half) {
  const int h = i >= half ? 1 : 0, ii = i - h * half;
  return (ii >> 3) * 16 + h * 8 + (ii & 7);
}

//   grid = (ceil(waves/WPB) * ceil(rowblocks/NRB), SPLIT ? chains : 1), block = WPB waves (blockDim.x / 64), LDS = NRB x 16 KiB.
//   A block = WPB adjacent 16-column tiles x one chain at a time x NRB 16-row blocks.  Per chain: every wave issues the loads
//   of its 16 weight fragments (16 x 1 KiB, all in flight together, HBM), the block copies the chain's activation slab
//   (NRB x 16 fragments, L2) into LDS once for all its waves, and each wave runs 16 x NRB MFMAs reading the activation
//   fragments from LDS.  Without the LDS copy every wave pulls the whole slab through L2 itself: 4x the weight bytes at 64 rows.
//   SPLIT (one chain per block, 256 threads): the slab loads are issued BEFORE the weight loads -- loads return in order, so
//   the L2-resident slab lands and is written to LDS while the weight fragments are still on their way from HBM.
//   Routed experts (rb_exp != nullptr; NRB = 1): row block rb belongs to expert rb_exp[rb], whose kernels start wstride 16-byte units after
//   wt; a row block of no expert (-1) leaves at once (k_moe_plan, ixt_moe_kernels.h).
//   EPI = 1 (whole-K blocks only): the columns are interleaved (wi_0, wi_1) tiles (k_tile_w mode 2) and `out` is the bf16 activation buffer of the
//   next dot: the odd wave of a pair hands its totals to the even one through LDS, which writes bf16(bf16(a * bf16(sigmoid a)) * b) in the x layout
//   (linears.py:331-361) -- no f32 slice leaves the chip (routed experts at decode: 132 MB of slices per layer and the kernel that re-read them).
template <int NRB, bool SPLIT, bool NT, int EPI = 0>
__global__ __launch_bounds__(256, 2) void k_dotb(const u32x4* __restrict__ xt, const u32x4* __restrict__ wt, int K, int N, int R,
                                                 float* __restrict__ out, int nwaves, int zblocks, const int32_t* __restrict__ rb_exp = nullptr,
                                                 size_t wstride = 0) {
  extern __shared__ __attribute__((aligned(16))) u32x4 xs[];   // [NRB][16 steps][64 lanes]
  const int tid = threadIdx.x, nthr = blockDim.x, wpb = nthr >> 6;
  const int wave = tid >> 6, lane = tid & 63;
  // Blocks that stream the same weights (the row-block groups of one tile group) must share an L2: consecutive block ids are
  // dealt round-robin over the 8 XCDs, so the row-block group is taken from (blockIdx.x / 8) and ids b, b + 8, .. are partners.
  const int bx = blockIdx.x;
  const int tile = ((bx / (8 * zblocks)) * 8 + (bx & 7)) * wpb + wave;
  const int nsteps = K >> 5;
  const int nchains = (nsteps + 15) >> 4;
  const int nrbk = (R + 15) >> 4;
  const int rb0 = ((bx >> 3) % zblocks) * NRB;
  const int c_begin = SPLIT ? (int)blockIdx.y : 0, c_end = SPLIT ? (int)blockIdx.y + 1 : nchains;
  const int r_in = lane & 15, cq = lane >> 4;
  const bool tile_ok = tile < nwaves;
  const u32x4* wbase = wt + (size_t)(tile_ok ? tile : 0) * nsteps * 64 + lane;
  if (rb_exp != nullptr) {
    const int ex = rb_exp[rb0];
    if (ex < 0) return;                                  // the whole block: no barrier is skipped
    wbase += (size_t)ex * wstride;
  }
  f32x4 tot[NRB];
#pragma unroll
  for (int b = 0; b < NRB; ++b) tot[b] = f32x4(0.0f);
  u32x4 wv[16];
#define IXT_WLOAD(S_, STEP_)                                                                     \
  do {                                                                                           \
    if (tile_ok && (STEP_) < nsteps) {                                                           \
      if constexpr (NT) wv[S_] = __builtin_nontemporal_load(wbase + (size_t)(STEP_) * 64);       \
      else wv[S_] = wbase[(size_t)(STEP_) * 64];                                                 \
    }                                                                                            \
  } while (0)
  u32x4 xq[SPLIT ? NRB * 4 : 1];                        // split-K: the one slab of this block, fetched into registers first
  if constexpr (SPLIT) {
    const int cs_ = min(16, nsteps - c_begin * 16);
#pragma unroll
    for (int j = 0; j < NRB * 4; ++j) {
      const int e = tid + j * nthr, b = e / (cs_ * 64), rem = e - b * (cs_ * 64);
      xq[j] = (e < NRB * cs_ * 64 && rb0 + b < nrbk) ? xt[((size_t)(rb0 + b) * nsteps + c_begin * 16) * 64 + rem] : u32x4(0);
    }
  }
#pragma unroll
  for (int s = 0; s < 16; ++s) IXT_WLOAD(s, c_begin * 16 + s);
  for (int c = c_begin; c < c_end; ++c) {
    const int s0 = c * 16;
    const int cs = min(16, nsteps - s0);              // steps of this chain
    // activation slab of this chain -> LDS (row block b at xs[b*1024 ...]: cs*64 consecutive 16-byte elements)
    if constexpr (SPLIT) {   // (split-K launches use 256-thread blocks: NRB * 4 slab elements per thread)
#pragma unroll
      for (int j = 0; j < NRB * 4; ++j) {
        const int e = tid + j * nthr, b = e / (cs * 64), rem = e - b * (cs * 64);
        if (e < NRB * cs * 64) xs[b * 1024 + rem] = xq[j];
      }
    } else {
      if (c > c_begin) __syncthreads();               // every wave is done reading the previous slab
      for (int e = tid; e < NRB * cs * 64; e += nthr) {
        const int b = e / (cs * 64), rem = e - b * (cs * 64);
        xs[b * 1024 + rem] = (rb0 + b < nrbk) ? xt[((size_t)(rb0 + b) * nsteps + s0) * 64 + rem] : u32x4(0);
      }
    }
    __syncthreads();
    f32x4 acc[NRB];
#pragma unroll
    for (int b = 0; b < NRB; ++b) acc[b] = f32x4(0.0f);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (s < cs) {
        const bf16x8 wf = __builtin_bit_cast(bf16x8, wv[s]);
#pragma unroll
        for (int b = 0; b < NRB; ++b)
          acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(bf16x8, xs[b * 1024 + s * 64 + lane]), acc[b], 0, 0, 0);
        if (!SPLIT && c + 1 < c_end) IXT_WLOAD(s, s0 + 16 + s);   // refill with the next chain's fragment
      }
    }
    if (SPLIT) {
#pragma unroll
      for (int b = 0; b < NRB; ++b) {
        const int row = (rb0 + b) * 16 + r_in;
        if (tile_ok && row < R)
          *reinterpret_cast<float4*>(out + ((size_t)c * R + row) * N + tile * 16 + cq * 4) = make_float4(acc[b][0], acc[b][1], acc[b][2], acc[b][3]);
      }
    } else {
#pragma unroll
      for (int b = 0; b < NRB; ++b) tot[b] = tot[b] + acc[b];
    }
  }
#undef IXT_WLOAD
  if constexpr (!SPLIT && EPI == 1) {
    static_assert(EPI == 0 || !SPLIT, "the gate needs the whole-K totals");
    __syncthreads();                                   // every wave is done with the last slab
    f32x4* ex = reinterpret_cast<f32x4*>(xs);          // [wave pair][NRB][64 lanes]
    if (wave & 1) {
#pragma unroll
      for (int b = 0; b < NRB; ++b) ex[((wave >> 1) * NRB + b) * 64 + lane] = tot[b];
    }
    __syncthreads();
    if (!(wave & 1) && tile_ok && tile + 1 < nwaves) {
      bf16_t* act = reinterpret_cast<bf16_t*>(out);
      const int F = N >> 1, f0 = (tile >> 1) * 16 + cq * 4;
#pragma unroll
      for (int b = 0; b < NRB; ++b) {
        const int row = (rb0 + b) * 16 + r_in;
        if (row < R) {
          const f32x4 up = ex[((wave >> 1) * NRB + b) * 64 + lane];
          bf16_t o[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = ixt_swiglu(rbf(tot[b][j]), rbf(up[j]));
          uint2 ov;
          ov.x = pack2(o[0], o[1]); ov.y = pack2(o[2], o[3]);
          *reinterpret_cast<uint2*>(act + xidx(row, f0, F)) = ov;
        }
      }
    }
  } else if (!SPLIT && tile_ok) {
#pragma unroll
    for (int b = 0; b < NRB; ++b) {
      const int row = (rb0 + b) * 16 + r_in;
      if (row < R)
        *reinterpret_cast<float4*>(out + (size_t)row * N + tile * 16 + cq * 4) = make_float4(tot[b][0], tot[b][1], tot[b][2], tot[b][3]);
    }
  }
}

// =====================================================================================================
// Unembedding + greedy sampling (decoders.py:566-587, inference_utils.py:74-75): the canonical dot over the [E][V] logits
// kernel with the argmax epilogue, organised for a matrix that is far larger than any other (V = 264 192 columns):
//   block = 4 waves x TS(4) tiles each = 256 adjacent columns, all chains, NRB 16-row blocks.
//   chain-outer / tile-inner: the activation slab of a chain is copied to LDS ONCE per chain and reused for the block's 16
//   (wave, tile) pairs; weight fragments stream through two 8-fragment register buffers, a quarter chain ahead of the MFMAs; the per-tile
//   totals stay in registers across the chains (t = t + chain(c), c ascending: the canonical order).
//   Each wave emits one (max, first index) pair per row over its 64 columns for k_argmax_final; logits (rounded to bf16,
//   stored as f32) are written only when `out` is given.   grid = ceil(V/256) * ceil(rowblocks/NRB), block 256,
//   LDS = NRB x 16 KiB.
// =====================================================================================================
// Columns the greedy sampler may pick: the union of up to 4 half-open ranges (n == 0: every column).  The expanded vocabulary
// keeps the audio codes in two runs of rows (6..104 and 256001..264093, audio_token_mapping_adjusted.json); restricting the
// argmax to them makes every generated id an audio id.  The logits themselves are not touched.
struct ColRanges { int n; int lo[4]; int hi[4]; };
__device__ __forceinline__ bool col_allowed(const ColRanges& cr, int col) {
  if (cr.n == 0) return true;
  bool ok = false;
#pragma unroll
  for (int i = 0; i < 4; ++i) ok = ok || (i < cr.n && col >= cr.lo[i] && col < cr.hi[i]);
  return ok;
}

// EPI = 1, TS = 4: the unembedding.  EPI = 0, TS = 2: the same streaming structure for a layer kernel at many rows (block = 128
// columns x all chains): the chain results are added in the block, so ONE f32 slice leaves the chip instead of `chains`
// (at 256 rows the slices of a split-K launch are as large as the weights).  Same bits as k_dotb: t = t + chain(c), c ascending.
template <int NRB, bool NT, int TS, int EPI>
__global__ __launch_bounds__(256, 2) void k_unembed(const u32x4* __restrict__ xt, const u32x4* __restrict__ wt, int K, int N, int R,
                                                    float* __restrict__ out, float* __restrict__ amax_val, int32_t* __restrict__ amax_idx,
                                                    int nwaves, int zblocks, ColRanges cr) {
  static_assert(TS == 2 || TS == 4, "one or two tile duos per wave");
  extern __shared__ __attribute__((aligned(16))) u32x4 xs[];   // [NRB][16 steps][64 lanes]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int bx = blockIdx.x;                                   // partners (same columns, other rows) are ids b, b + 8, ..: same XCD, same L2
  const int wid = ((bx / (8 * zblocks)) * 8 + (bx & 7)) * 4 + wave;
  const int nsteps = K >> 5, ntile = N >> 4;
  const int nchains = (nsteps + 15) >> 4;
  const int nrbk = (R + 15) >> 4;
  const int rb0 = ((bx >> 3) % zblocks) * NRB;
  const int r_in = lane & 15, cq = lane >> 4;
  const int tile0 = wid * TS;
  f32x4 tot[TS][NRB];
#pragma unroll
  for (int t = 0; t < TS; ++t)
#pragma unroll
    for (int b = 0; b < NRB; ++b) tot[t][b] = f32x4(0.0f);
  u32x4 wa[8], wb[8];
  // A wave runs its 4 tiles as 2 duos: both tiles of a duo consume the SAME activation fragments, read from LDS once per
  // step (the LDS port, shared by the 4 SIMDs, is the scarce resource here: one 1 KiB read per MFMA would saturate it).
  // quarter Q_ (steps 4 Q_ .. 4 Q_ + 3) of duo D_ in chain C_: 2 tiles x 4 fragments; a tile past the end loads nothing
#define IXT_ULOADQ(BUF_, C_, D_, Q_)                                                                   \
  do {                                                                                                 \
    if ((C_) * 16 + (Q_) * 4 < nsteps) {   /* feature sizes are multiples of 128: a quarter is whole or absent */ \
      _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                  \
        const int tl_ = min(tile0 + (D_) * 2 + u, ntile - 1);   /* past the end: a valid tile, result dropped */ \
        const u32x4* wp_ = wt + ((size_t)tl_ * nsteps + (C_) * 16 + (Q_) * 4) * 64 + lane;             \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                \
          if constexpr (NT) BUF_[u * 4 + s] = __builtin_nontemporal_load(wp_ + (size_t)s * 64);        \
          else BUF_[u * 4 + s] = wp_[(size_t)s * 64];                                                  \
        }                                                                                              \
      }                                                                                                \
    }                                                                                                  \
  } while (0)
#define IXT_UMMAQ(BUF_, Q_)                                                                            \
  do {                                                                                                 \
    if ((Q_) * 4 < cs) {                                                                               \
      _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                  \
        bf16x8 xf_[NRB];                                                                               \
        _Pragma("unroll") for (int b = 0; b < NRB; ++b) xf_[b] = __builtin_bit_cast(bf16x8, xs[b * 1024 + ((Q_) * 4 + s) * 64 + lane]); \
        _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                \
          const bf16x8 wf_ = __builtin_bit_cast(bf16x8, BUF_[u * 4 + s]);                              \
          _Pragma("unroll") for (int b = 0; b < NRB; ++b)                                              \
            acc[u][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf_, xf_[b], acc[u][b], 0, 0, 0);      \
        }                                                                                              \
      }                                                                                                \
    }                                                                                                  \
  } while (0)
  // duo D_ of chain c: quarters alternate between the two buffers; the next quarter (of this duo, of the next duo, or of the
  // next chain's first duo: (CN_, DN_)) is always in flight while the current one runs
#define IXT_UDUO(D_, CN_, DN_)                                                                         \
  do {                                                                                                 \
    f32x4 acc[2][NRB];                                                                                 \
    _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                      \
      _Pragma("unroll") for (int b = 0; b < NRB; ++b) acc[u][b] = f32x4(0.0f);                         \
    IXT_ULOADQ(wb, c, D_, 1); IXT_UMMAQ(wa, 0);                                                        \
    IXT_ULOADQ(wa, c, D_, 2); IXT_UMMAQ(wb, 1);                                                        \
    IXT_ULOADQ(wb, c, D_, 3); IXT_UMMAQ(wa, 2);                                                        \
    IXT_ULOADQ(wa, CN_, DN_, 0); IXT_UMMAQ(wb, 3);                                                     \
    _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                      \
      _Pragma("unroll") for (int b = 0; b < NRB; ++b) tot[(D_) * 2 + u][b] = tot[(D_) * 2 + u][b] + acc[u][b]; \
  } while (0)
  IXT_ULOADQ(wa, 0, 0, 0);
  for (int c = 0; c < nchains; ++c) {
    const int s0 = c * 16;
    const int cs = min(16, nsteps - s0);
    if (c > 0) __syncthreads();                          // every wave is done with the previous slab
    for (int e = tid; e < NRB * cs * 64; e += 256) {
      const int b = e / (cs * 64), rem = e - b * (cs * 64);
      xs[b * 1024 + rem] = (rb0 + b < nrbk) ? xt[((size_t)(rb0 + b) * nsteps + s0) * 64 + rem] : u32x4(0);
    }
    __syncthreads();
    if constexpr (TS == 4) {
      IXT_UDUO(0, c, 1);
      IXT_UDUO(1, c + 1, 0);
    } else {
      IXT_UDUO(0, c + 1, 0);
    }
  }
#undef IXT_ULOADQ
#undef IXT_UMMAQ
#undef IXT_UDUO
  if (wid >= nwaves) return;
  if constexpr (EPI == 0) {
#pragma unroll
    for (int b = 0; b < NRB; ++b) {
      const int row = (rb0 + b) * 16 + r_in;
#pragma unroll
      for (int t = 0; t < TS; ++t)
        if (tile0 + t < ntile && row < R)
          *reinterpret_cast<float4*>(out + (size_t)row * N + (tile0 + t) * 16 + cq * 4) = make_float4(tot[t][b][0], tot[t][b][1], tot[t][b][2], tot[t][b][3]);
    }
    return;
  }
#pragma unroll
  for (int b = 0; b < NRB; ++b) {
    const int row = (rb0 + b) * 16 + r_in;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int t = 0; t < TS; ++t) {
      if (tile0 + t < ntile) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = rbf(tot[t][b][j]);
        if (out != nullptr && row < R) *reinterpret_cast<float4*>(out + (size_t)row * N + (tile0 + t) * 16 + cq * 4) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (v[j] > bv && col_allowed(cr, (tile0 + t) * 16 + cq * 4 + j)) { bv = v[j]; bi = (tile0 + t) * 16 + cq * 4 + j; }   // ascending columns: a later equal value never replaces
      }
    }
#pragma unroll
    for (int off = 16; off < 64; off <<= 1) {   // the 4 lanes of a row are lanes r, r+16, r+32, r+48
      const float ov = __shfl_xor(bv, off, 64);
      const int oi = __shfl_xor(bi, off, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (cq == 0 && row < R) {
      amax_val[(size_t)row * nwaves + wid] = bv;
      amax_idx[(size_t)row * nwaves + wid] = bi;
    }
  }
}

// =====================================================================================================
// Canonical DOT as a tiled GEMM, for many rows (>= 128) x many columns: block = 4 waves (2 x 2) = 128 rows x 128 columns x
// ALL of K, wave = 64 x 64 (4 x 4 MFMA tiles: every weight fragment meets 4 activation fragments and vice versa, so a k-step
// costs 8 LDS fragment reads for 16 MFMAs and the L2 -> CU traffic per MAC is a quarter of k_dotb's).  K is walked in chunks of
// 64 (2 k-steps): the next chunk's 32 KiB (16 weight + 16 activation fragments) are fetched into registers while the current
// one is consumed from LDS (2 stages, one barrier per chunk).  Every 512 k the accumulators are added to the totals
// (t = t + chain(c), c ascending: the canonical order), so ONE f32 slice leaves the chip instead of `chains`.
//   grid = ceil(N/128) * ceil(R/128) (row groups of the same columns are ids b, b + 8, ..: same XCD), block 256, LDS 64 KiB.
// =====================================================================================================
//   CT = column tiles per wave: 4 (block 128 x 128, LDS 64 KiB) or 2 (block 128 rows x 64 columns, LDS 48 KiB: twice the blocks,
//   three per CU -- for matrices whose 128-column groups alone would leave CUs with a single block and one chunk in flight).
//   EPI = 1: the greedy-sampling epilogue of k_unembed (rounded logits on request, one (max, first index) pair per row and per
//   wave column of 16 CT columns into amax_*[row][2 ncolg]) -- the unembedding at >= 128 rows.
//   SPLIT: blockIdx.y = chain: the block walks only the 8 chunks of its chain and writes slice [chain][R][N] (split-K for
//   matrices with too few 128 x 128 blocks to fill the chip; still half of k_dotb's L2 traffic per MAC).
//   (Fetching 3 chunks ahead into registers -- one wave per SIMD, 512 registers -- was measured slower: 5.68 vs 5.25 ms per step.)
//   SPLIT = 2 (prefix halves): blockIdx.y = 0 walks the first half of the chains and writes their running total as slice 0;
//   blockIdx.y = 1 walks the second half and writes every chain as its own slice (1, 2, ..): summed in slice order this is still
//   ((0 + c0) + c1) + c2 + c3, twice the blocks of the full-K form for one or two extra slices.
template <bool NT, int CT, int EPI, int SPLIT>
__global__ __launch_bounds__(256, 2) void k_gemm(const u32x4* __restrict__ xt, const u32x4* __restrict__ wt, int K, int N, int R,
                                                 float* __restrict__ out, int ncolg, int zblocks, float* __restrict__ amax_val,
                                                 int32_t* __restrict__ amax_idx, ColRanges cr) {
  constexpr int NWF = 4 * CT;          // weight fragments per chunk (2 CT tiles x 2 k-steps)
  constexpr int NF = NWF + 16;         // + 8 row blocks x 2 k-steps of activations
  constexpr int NJ = NF / 4;           // fragments staged per thread and chunk
  extern __shared__ __attribute__((aligned(16))) u32x4 st[];   // [2 stages][NF fragments][64 lanes]; f < NWF: weights (tile f>>1, k-step f&1), else activations
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int bx = blockIdx.x;
  const int colg = (bx / (8 * zblocks)) * 8 + (bx & 7), rowg = (bx >> 3) % zblocks;
  if (colg >= ncolg) return;                                    // whole block: no barrier is skipped
  const int nsteps = K >> 5, ntile = N >> 4, nrbk = (R + 15) >> 4;
  const int nchunks = nsteps >> 1;
  const int wr = wave >> 1, wc = wave & 1;
  const int r_in = lane & 15, cq = lane >> 4;
  const u32x4* src[NJ];
  bool ok[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int f = wave + 4 * j;
    if (f < NWF) {
      const int tile = colg * 2 * CT + (f >> 1);
      ok[j] = tile < ntile;
      src[j] = wt + ((size_t)(ok[j] ? tile : 0) * nsteps + (f & 1)) * 64 + lane;
    } else {
      const int rb = rowg * 8 + ((f - NWF) >> 1);
      ok[j] = rb < nrbk;
      src[j] = xt + ((size_t)(ok[j] ? rb : 0) * nsteps + (f & 1)) * 64 + lane;
    }
  }
  u32x4 pre[NJ];
#define IXT_GFETCH(CH_)                                                                              \
  do {                                                                                               \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                 \
      if (!ok[j]) pre[j] = u32x4(0);                                                                 \
      else if (NT && wave + 4 * j < NWF) pre[j] = __builtin_nontemporal_load(src[j] + (size_t)(CH_) * 128); \
      else pre[j] = src[j][(size_t)(CH_) * 128];                                                     \
    }                                                                                                \
  } while (0)
#define IXT_GSTORE(STAGE_)                                                                           \
  do {                                                                                               \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j) st[((STAGE_) * NF + wave + 4 * j) * 64 + lane] = pre[j]; \
  } while (0)
  f32x4 acc[CT][4], tot[CT][4];
#pragma unroll
  for (int i = 0; i < CT; ++i)
#pragma unroll
    for (int b = 0; b < 4; ++b) { acc[i][b] = f32x4(0.0f); tot[i][b] = f32x4(0.0f); }
  const int hchains = ((nchunks + 7) >> 3) >> 1;                // SPLIT 2: chains of the first half
  const int chb = SPLIT == 1 ? (int)blockIdx.y * 8 : (SPLIT == 2 && blockIdx.y == 1) ? hchains * 8 : 0;   // chunk range of this block
  const int che = SPLIT == 1 ? min(nchunks, chb + 8) : (SPLIT == 2 && blockIdx.y == 0) ? hchains * 8 : nchunks;
  if (SPLIT == 1) out += (size_t)blockIdx.y * R * N;
  const bool per_chain = SPLIT == 2 && blockIdx.y == 1;         // every chain of this block is its own slice
  IXT_GFETCH(chb);
  IXT_GSTORE(0);
  __syncthreads();
  for (int ch = chb; ch < che; ++ch) {
    if (ch + 1 < che) IXT_GFETCH(ch + 1);
    const u32x4* sb = st + (size_t)((ch - chb) & 1) * NF * 64;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 wf[CT], xf[4];
#pragma unroll
      for (int i = 0; i < CT; ++i) wf[i] = __builtin_bit_cast(bf16x8, sb[((wc * CT + i) * 2 + ks) * 64 + lane]);
#pragma unroll
      for (int b = 0; b < 4; ++b) xf[b] = __builtin_bit_cast(bf16x8, sb[(NWF + (wr * 4 + b) * 2 + ks) * 64 + lane]);
#pragma unroll
      for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[i][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[b], acc[i][b], 0, 0, 0);
    }
    if ((ch & 7) == 7 || ch + 1 == che) {   // a 512-k chain is complete
      if (SPLIT == 2 && per_chain) {
        float* so = out + (size_t)(1 + (ch >> 3) - hchains) * R * N;
#pragma unroll
        for (int i = 0; i < CT; ++i) {
          const int tile = colg * 2 * CT + wc * CT + i;
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const int row = (rowg * 8 + wr * 4 + b) * 16 + r_in;
            if (tile < ntile && row < R)
              *reinterpret_cast<float4*>(so + (size_t)row * N + tile * 16 + cq * 4) = make_float4(acc[i][b][0], acc[i][b][1], acc[i][b][2], acc[i][b][3]);
            acc[i][b] = f32x4(0.0f);
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
          for (int b = 0; b < 4; ++b) { tot[i][b] = tot[i][b] + acc[i][b]; acc[i][b] = f32x4(0.0f); }
      }
    }
    if (ch + 1 < che) IXT_GSTORE((ch + 1 - chb) & 1);
    __syncthreads();
  }
#undef IXT_GFETCH
#undef IXT_GSTORE
  if (SPLIT == 2 && per_chain) return;
  if constexpr (EPI == 1) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int row = (rowg * 8 + wr * 4 + b) * 16 + r_in;
      float bv = -INFINITY;
      int bi = 0x7fffffff;
#pragma unroll
      for (int i = 0; i < CT; ++i) {
        const int tile = colg * 2 * CT + wc * CT + i;
        if (tile < ntile) {
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = rbf(tot[i][b][j]);
          if (out != nullptr && row < R) *reinterpret_cast<float4*>(out + (size_t)row * N + tile * 16 + cq * 4) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (v[j] > bv && col_allowed(cr, tile * 16 + cq * 4 + j)) { bv = v[j]; bi = tile * 16 + cq * 4 + j; }   // ascending columns
        }
      }
#pragma unroll
      for (int off = 16; off < 64; off <<= 1) {
        const float ov = __shfl_xor(bv, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
      }
      if (cq == 0 && row < R) {
        amax_val[(size_t)row * (2 * ncolg) + colg * 2 + wc] = bv;
        amax_idx[(size_t)row * (2 * ncolg) + colg * 2 + wc] = bi;
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < CT; ++i) {
    const int tile = colg * 2 * CT + wc * CT + i;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int row = (rowg * 8 + wr * 4 + b) * 16 + r_in;
      if (tile < ntile && row < R)
        *reinterpret_cast<float4*>(out + (size_t)row * N + tile * 16 + cq * 4) = make_float4(tot[i][b][0], tot[i][b][1], tot[i][b][2], tot[i][b][3]);
    }
  }
}

// weights: row-major [K][ncols] (bf16 or f32) -> the tiled fragment layout of a [K][*] matrix, columns n_off..n_off+ncols
//   mode 0: column n -> n_off + n
//   mode 1: Q / K heads of `hd` columns, pair-interleaved (qk_col): feature i of a head -> the column that keeps both RoPE partners of
//           a feature in one 16-column tile (ixt_gemm.h, GEPI_QKV)
//   mode 2: the fused MLP gate kernel: tile j of this source becomes column tile 2 j + n_off (n_off = 0: wi_0, 1: wi_1), so that a
//           wave holds both gates of a feature (ixt_gemm.h, GEPI_SWIGLU; k_swiglu_part reads the same layout)
template <typename T>
__global__ void k_tile_w(const T* __restrict__ src, bf16_t* __restrict__ dst, int K, int ncols, int n_off, int mode, int hd, int ring) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)K * ncols) return;
  const int k = (int)(i / ncols), n = (int)(i % ncols);
  bf16_t v;
  if constexpr (sizeof(T) == 2) v = src[i]; else v = f2bf(src[i]);
  int col;
  if (mode == 1) col = n_off + (n / hd) * hd + qk_col(n % hd, hd >> 1);
  else if (mode == 3) { const int f = n % hd; col = n_off + (n / hd) * hd + qk_col((f & 1) ? (hd >> 1) + (f >> 1) : (f >> 1), hd >> 1); }   // llama3.1: de-interleave, then as mode 1
  else if (mode == 2) col = ((n >> 4) * 2 + n_off) * 16 + (n & 15);
  else col = n_off + n;
  dst[ring ? r8idx(k, col, K) : widx(k, col, K)] = v;
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
  const int32_t* row_index;   // MODE 2: output row r normalises h_in row row_index[r] (nullptr: row r)
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
  if (MODE == 1) src = a.h_in + (size_t)r * E;
  if (MODE == 2) src = a.h_in + (size_t)(a.row_index != nullptr ? a.row_index[r] : r) * E;
  // this thread's first scale chunk: fetched now, so that it is not one more memory round trip after the barrier
  const u32x4 sv_first = (a.scale != nullptr && tid < nch) ? *reinterpret_cast<const u32x4*>(a.scale + tid * 8) : u32x4(0);
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
    const u32x4 sv = ch == tid ? sv_first : *reinterpret_cast<const u32x4*>(a.scale + ch * 8);
    const float sc[8] = {bflo(sv.x), bfhi(sv.x), bflo(sv.y), bfhi(sv.y), bflo(sv.z), bfhi(sv.z), bflo(sv.w), bfhi(sv.w)};
    bf16_t o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f2bf(rbf(hrow[ch * 8 + j] * rs) * sc[j]);
    u32x4 ov;
    ov.x = pack2(o[0], o[1]); ov.y = pack2(o[2], o[3]); ov.z = pack2(o[4], o[5]); ov.w = pack2(o[6], o[7]);
    // in the dot kernel's x layout 8 consecutive features of a row are one lane's 16 bytes
    *reinterpret_cast<u32x4*>(a.xn_out + (a.xn_kmajor ? xidx(r, ch * 8, E) : (size_t)r * E + ch * 8)) = ov;
  }
}

// =====================================================================================================
// SwiGLU from partial slices (linears.py:449-464): part is [slices][R][2F] (column tiles alternate wi_0 / wi_1: k_tile_w mode 2).
// grid = (ceil(F/2048), R), block 256: a thread owns 8 consecutive features (two 16-byte loads per slice and operand, one
// 16-byte store: 8 consecutive features of a row are one lane's fragment bytes in the dot kernel's x layout).
// =====================================================================================================
__global__ __launch_bounds__(256) void k_swiglu_part(const float* part, int slices, size_t slice_stride, int F,
                                                      bf16_t* out /*[R][F] in xidx layout*/, const int32_t* rb_exp = nullptr) {
  const int f0 = (blockIdx.x * 256 + threadIdx.x) * 8, r = blockIdx.y;
  if (f0 >= F) return;
  if (rb_exp != nullptr && rb_exp[r >> 4] < 0) return;      // routed experts: a row block nobody's copies live in
  const float* p = part + (size_t)r * 2 * F;
  float a[8], b[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = 0.0f; b[j] = 0.0f; }
  for (int s = 0; s < slices; ++s) {   // slices added in order from 0 (sum_slices semantics)
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
  for (int j = 0; j < 8; ++j) o[j] = ixt_swiglu(rbf(a[j]), rbf(b[j]));
  u32x4 ov;
  ov.x = pack2(o[0], o[1]); ov.y = pack2(o[2], o[3]); ov.z = pack2(o[4], o[5]); ov.w = pack2(o[6], o[7]);
  *reinterpret_cast<u32x4*>(out + xidx(r, f0, F)) = ov;
}

// op-level helpers of the layers.* module surface: residual add bf16(a + b) (llama2.py:127, :157) and the embedding gather
// (embeddings.py:154).  Elementwise, 8 elements per thread where the size allows.
__global__ __launch_bounds__(256) void k_residual(const bf16_t* a, const bf16_t* b, bf16_t* out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = f2bf(bf2f(a[i]) + bf2f(b[i]));
}
__global__ __launch_bounds__(256) void k_embed(const bf16_t* table, const int32_t* ids, bf16_t* out, int E, int V) {
  const int r = blockIdx.y, c = (blockIdx.x * 256 + threadIdx.x) * 8;
  if (c >= E) return;
  int id = ids[r];
  id = id < 0 ? 0 : (id >= V ? V - 1 : id);
  *reinterpret_cast<u32x4*>(out + (size_t)r * E + c) = *reinterpret_cast<const u32x4*>(table + (size_t)id * E + c);
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
  int rope32;                           // 1: rope_type "llama3.1" (embeddings.py:354-488): the tables are f32, one rounding at the end (ixt_rope_pair)
};

// Where feature f of the kernels' de-interleaved Q / K order goes in the q buffer and the K pages: its place in the MODEL's order (llama3.1:
// position i of the first half is feature 2 i, of the second half feature 2 i + 1), so that every score is the same fma chain over d as the
// oracle's; the default form keeps f.
__device__ __forceinline__ int ixt_qk_feat(int rope32, int f, int half) { return rope32 ? (f < half ? 2 * f : 2 * (f - half) + 1) : f; }

// One RoPE pair (x1, x2) at table index idx.  Default RotaryEmbedding (embeddings.py:277-315): bf16 tables, every product and the final sum round
// to bf16.  LLaMARotaryEmbedding (embeddings.py:354-488, rope_type "llama3.1"): `inputs * cos + inputs_shifted * sin * sign` with f32 sin / cos,
// evaluated in f32 and cast to bf16 once.  The llama3.1 pairs are the NEIGHBOURING features (2i, 2i + 1); the engine stores the Q / K kernels'
// columns de-interleaved (even features first: k_tile_w mode 3), so that the pair sits at (i, half + i) here as for the default form; the
// rotated values go back to their places in the model's feature order (ixt_qk_feat).
__device__ __forceinline__ void ixt_rope_pair(int rope32, const bf16_t* sn, const bf16_t* cs, size_t idx, float x1, float x2, float& o1, float& o2) {
  if (rope32) {
    const float s = reinterpret_cast<const float*>(sn)[idx], c = reinterpret_cast<const float*>(cs)[idx];
    o1 = rbf(x1 * c - x2 * s);
    o2 = rbf(x2 * c + x1 * s);
  } else {
    const float s = bf2f(sn[idx]), c = bf2f(cs[idx]);
    o1 = rbf(rbf(x1 * c) - rbf(x2 * s));
    o2 = rbf(rbf(x2 * c) + rbf(x1 * s));
  }
}

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
  // Gemma-3 (attentions.py:2246-2248, :2262-2264): RMSNorm over head_dim of every query / key head before RoPE (scales [hd]; nullptr:
  // none) and q * q_scalar (a bf16 value) after it (0: none)
  const bf16_t* qnorm = nullptr; const bf16_t* knorm = nullptr;
  float eps = 0.0f, q_scalar = 0.0f;
};

__global__ __launch_bounds__(64) void k_qkv_finalize(QkvArgs a) {
  const int r = blockIdx.x, hh = blockIdx.y;
  const AttnGeom& g = a.g;
  const int half = g.hd / 2;
  const int col0 = hh * g.hd;
  const float* pq = a.part + (size_t)r * a.ncols + col0;
  const bool is_v = hh >= g.nq + g.nkv;
  const int pos = a.row_pos[r];
  // head norm (block-uniform): the head's bf16 values in feature order -> LDS; sum of squares in the RMSNorm order (lane = chunk of 8
  // features by fma, xor tree over the wave; head_dim <= 512 is one block)
  const bf16_t* nscale = is_v ? nullptr : (hh < g.nq ? a.qnorm : a.knorm);
  __shared__ float hv[512];
  float nr = 0.0f;
  if (nscale != nullptr) {
    for (int i = threadIdx.x; i < half; i += 64) {
      hv[i] = rbf(sum_slices<16>(pq + qk_col(i, half), a.slice_stride, a.slices));
      hv[half + i] = rbf(sum_slices<16>(pq + qk_col(half + i, half), a.slice_stride, a.slices));
    }
    __syncthreads();
    float ss = 0.0f;
    if ((int)threadIdx.x < g.hd / 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float x = hv[threadIdx.x * 8 + j];
        ss = __builtin_fmaf(x, x, ss);
      }
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) ss = ss + __shfl_xor(ss, off, 64);
    const float total = 0.0f + ss;
    nr = 1.0f / sqrtf(total / (float)g.hd + a.eps);
  }
  const float qs = (hh < g.nq && a.q_scalar != 0.0f) ? a.q_scalar : 0.0f;
  for (int i = threadIdx.x; i < half; i += 64) {   // head_dim 256: two RoPE pairs per lane
    float v1, v2;
    if (nscale != nullptr) {
      v1 = rbf(rbf(hv[i] * nr) * bf2f(nscale[i]));
      v2 = rbf(rbf(hv[half + i] * nr) * bf2f(nscale[half + i]));
    } else {
      v1 = rbf(sum_slices<16>(pq + (is_v ? i : qk_col(i, half)), a.slice_stride, a.slices));
      v2 = rbf(sum_slices<16>(pq + (is_v ? half + i : qk_col(half + i, half)), a.slice_stride, a.slices));
    }
    if (!is_v) {
      float o1, o2;
      ixt_rope_pair(g.rope32, a.rope_sin, a.rope_cos, (size_t)pos * half + i, v1, v2, o1, o2);
      v1 = o1;
      v2 = o2;
      if (qs != 0.0f) {
        v1 = rbf(v1 * qs);
        v2 = rbf(v2 * qs);
      }
    }
    const int f1 = is_v ? i : ixt_qk_feat(g.rope32, i, half), f2 = is_v ? half + i : ixt_qk_feat(g.rope32, half + i, half);
    if (hh < g.nq) {
      bf16_t* q = a.qbuf + ((size_t)r * g.nq + hh) * g.hd;
      q[f1] = f2bf(v1);
      q[f2] = f2bf(v2);
      continue;
    }
    if (!a.row_active[r]) continue;
    const int page = a.page_map[(size_t)a.row_slot[r] * g.mpg + pos / g.P];
    const int tok = pos % g.P;
    if (!is_v) {
      const int kvh = hh - g.nq;
      a.kpages[kpage_off(g, page, kvh, f1 / 8, tok) + (f1 & 7)] = f2bf(v1);
      a.kpages[kpage_off(g, page, kvh, f2 / 8, tok) + (f2 & 7)] = f2bf(v2);
    } else {
      const int kvh = hh - g.nq - g.nkv;
      bf16_t* v = a.vpages + vpage_off(g, page, kvh, tok);
      v[i] = f2bf(v1);
      v[half + i] = f2bf(v2);
    }
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
    const bool rope_ = which <= GQ;   // q and k heads are stored pair-interleaved (qk_col), v heads in feature order
    float v1 = rbf(sum_slices<16>(pq + (rope_ ? qk_col(i, half) : i), a.slice_stride, a.slices));
    float v2 = rbf(sum_slices<16>(pq + (rope_ ? qk_col(half + i, half) : half + i), a.slice_stride, a.slices));
    if (which <= GQ) {
      float o1, o2;
      ixt_rope_pair(g.rope32, a.rope_sin, a.rope_cos, (size_t)pos * half + i, v1, v2, o1, o2);
      v1 = o1; v2 = o2;
    }
    const int f1 = ixt_qk_feat(g.rope32, i, half), f2 = ixt_qk_feat(g.rope32, half + i, half);   // (query / key rows only)
    if (which < GQ) {
      qs[which * HD + f1] = v1; qs[which * HD + f2] = v2;
    } else if (which == GQ) {
      kn[f1] = v1; kn[f2] = v2;
      if (sp == 0) {
        const int page = prow[pos / g.P], tok = pos % g.P;
        a.kpages[kpage_off(g, page, kvh, f1 / 8, tok) + (f1 & 7)] = f2bf(v1);
        a.kpages[kpage_off(g, page, kvh, f2 / 8, tok) + (f2 & 7)] = f2bf(v2);
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
  float cap;          // > 0: score = ixt_tanh(score / cap) * cap before the max (attn_logits_soft_cap, attentions.py:1257-1259)
  int window;         // > 0: keys j <= pos - window are invisible (LOCAL_SLIDING, attentions.py:625-631): e_j = p_j = +0, every sum unchanged
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
  constexpr int FPL = HD > 128 ? 16 : 8;  // features per lane: FPL/8 consecutive 16-byte d-chunks
  constexpr int NCH = FPL / 8;
  constexpr int LPP = HD / FPL;           // lanes per page
  constexpr int PPW = 64 / LPP;           // pages a wave covers at once (4 for hd 128 and 256, 8 for hd 64)
  const int j0 = chunk * 4 * IXT_SB * g.P;  // first key of this block's range
  const int j1 = min(kv_len, j0 + 4 * IXT_SB * g.P);
  const int jvis = (a.window > 0 && kv_len > a.window) ? kv_len - a.window : 0;   // first visible key

  for (int q = 0; q < GQ; ++q) {
    const int h = kvh * GQ + q;
    const float* srow = a.scores + ((size_t)r * g.nq + h) * g.Tmax;
    if (q > 0) __syncthreads();
    // scores -> LDS (coalesced), max (order independent)
    float m = -INFINITY;
    for (int j = tid; j < kv_len; j += 256) {
      float sv = srow[j];
      if (a.cap > 0.0f) sv = ixt_tanh(sv / a.cap) * a.cap;
      ebuf[j] = sv;
      if (j >= jvis) m = fmaxf(m, sv);
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if (lane == 0) wred[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]));
    for (int j = tid; j < kv_len; j += 256) ebuf[j] = j >= jvis ? ixt_exp(ebuf[j] - m) : 0.0f;
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
      float acc[FPL];
#pragma unroll
      for (int j = 0; j < FPL; ++j) acc[j] = 0.0f;
      if (p < pages) {
        const int page = prow[p];
        const bf16_t* vp = a.vpages + vpage_off(g, page, kvh, 0) + dc * FPL;
        const int ntok = min(g.P, kv_len - p * g.P);
        const float* pj = ebuf + (size_t)p * g.P;
        for (int t0 = 0; t0 < ntok; t0 += 8) {
          u32x4 vv[8][NCH];
#pragma unroll
          for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int cch = 0; cch < NCH; ++cch)
              vv[u][cch] = (t0 + u < ntok) ? *reinterpret_cast<const u32x4*>(vp + (size_t)(t0 + u) * HD + cch * 8) : u32x4(0);
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            if (t0 + u < ntok) {
              const float pv = pj[t0 + u];
#pragma unroll
              for (int cch = 0; cch < NCH; ++cch) {
                float* ac = acc + cch * 8;
                const u32x4 w = vv[u][cch];
                ac[0] = __builtin_fmaf(pv, bflo(w.x), ac[0]); ac[1] = __builtin_fmaf(pv, bfhi(w.x), ac[1]);
                ac[2] = __builtin_fmaf(pv, bflo(w.y), ac[2]); ac[3] = __builtin_fmaf(pv, bfhi(w.y), ac[3]);
                ac[4] = __builtin_fmaf(pv, bflo(w.z), ac[4]); ac[5] = __builtin_fmaf(pv, bfhi(w.z), ac[5]);
                ac[6] = __builtin_fmaf(pv, bflo(w.w), ac[6]); ac[7] = __builtin_fmaf(pv, bfhi(w.w), ac[7]);
              }
            }
          }
        }
      }
      // super-block sum: pages 4s..4s+3 live in lane groups tq = 4s'..4s'+3 of this wave
      const int sbl = tq / IXT_SB;            // super-block within the wave
      const int base_lane = sbl * IXT_SB * LPP + dc;
      float sbs[FPL];
#pragma unroll
      for (int j = 0; j < FPL; ++j) sbs[j] = 0.0f;
#pragma unroll
      for (int i = 0; i < IXT_SB; ++i) {
#pragma unroll
        for (int j = 0; j < FPL; ++j) sbs[j] = sbs[j] + __shfl(acc[j], base_lane + i * LPP, 64);
      }
      if (tq % IXT_SB == 0) {
        const int sb = chunk * 4 + (wave * PPW) / IXT_SB + sbl;
        if (sb < nsb) {
          float* dst = a.pvpart + (((size_t)r * g.nq + h) * g.nsb_max + sb) * HD + dc * FPL;
#pragma unroll
          for (int j = 0; j < FPL; j += 4) *reinterpret_cast<float4*>(dst + j) = make_float4(sbs[j], sbs[j + 1], sbs[j + 2], sbs[j + 3]);
        }
      }
    }
  }
}

// ---- steps 1-4 in ONE launch for decode rows: block = (row, kv head) owns the whole key range of its row.
//      QKV slice sums + RoPE + K/V page write (as k_attn_front_decode), scores into LDS, softmax statistics once, PV per
//      super-block, the in-order sum over super-blocks through LDS, bf16 output in the dot kernel's x layout.  No score /
//      partial round trip through HBM, no per-split recomputation of the softmax, and the first key tile is already in
//      flight while the prologue runs.  The new token's K and V are taken from LDS, never from the page being written.
//      Same operation order as the split kernels (scores: one fma chain over d; l: page sums in key order, pages in order;
//      PV: page chains, 4 pages per super-block in order, super-blocks in order).  grid = (nkv, R), block 256.
struct DecAttnArgs {
  const float* part; int slices; size_t slice_stride; int ncols;
  const int32_t* row_slot; const int32_t* row_pos; const int32_t* row_active; const int32_t* page_map;
  const bf16_t* rope_sin; const bf16_t* rope_cos;
  bf16_t* kpages; bf16_t* vpages;
  bf16_t* out;        // [R][nq*hd] in xidx layout
  AttnGeom g;
  const bf16_t* qbuf; // NEWTOK = false: [R][nq][hd] finalised queries
  int rows;           // k_attn_small: blocks with blockIdx.y >= rows are prefetch blocks
  const u32x4* pf; size_t pf_n;
};

// NEWTOK = true: decode rows (the row's own new token is finalised here: QKV slices -> RoPE -> pages, and read back from LDS).
// NEWTOK = false: prefill rows: q comes from qbuf and every key / value from the pages (k_qkv_finalize ran before this launch).
template <int HD, int GQ, bool NEWTOK>
__global__ __launch_bounds__(256) void k_attn_decode(DecAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const AttnGeom& g = a.g;
  const int kvh = blockIdx.x, r = blockIdx.y;   // heads fastest: the 16 blocks of a row read 16 adjacent 8 KiB pieces of each page
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n_out = g.nq * HD;
  if (!a.row_active[r]) {
    for (int i = tid; i < GQ * HD; i += 256) a.out[xidx(r, kvh * GQ * HD + i, n_out)] = 0;
    return;
  }
  float* qs = sm;                               // [GQ][HD]
  float* kn = qs + GQ * HD;                     // [HD] new key (bf16-rounded, as f32)
  float* vn = kn + HD;                          // [HD] new value
  float* ebuf = vn + HD;                        // [GQ][Tmax] scores -> exps -> P
  const int tp = (g.Tmax + 3) & ~3;             // row pitch of ebuf: 16-byte aligned rows, and psum after them
  float* psum = ebuf + (size_t)GQ * tp;         // [mpg]
  float* wred = psum + ((g.mpg + 3) & ~3);        // [8]
  float* sbs = wred + 8;                        // [nsb_max][HD] super-block sums of the head being finished
  const int pos = a.row_pos[r];
  const int kv_len = pos + 1;
  const int pages = (kv_len + g.P - 1) / g.P;
  const int nsb = (pages + IXT_SB - 1) / IXT_SB;
  const int ppw = 64 / g.P;
  constexpr int half = HD / 2;
  const int32_t* prow = a.page_map + (size_t)a.row_slot[r] * g.mpg;
  const int QN = g.nq * HD, KN = g.nkv * HD;

  // Score work of this wave as a flat list of half tiles: tile i = keys of pages (wave + 4 i) ppw .., half h = d-chunks
  // h HD/16 .. (h + 1) HD/16 - 1 of the lane's key.  Two register buffers of HD/16 x 16 bytes; the first half tile is in flight
  // before the prologue, and every later one before the fma chains of its predecessor.
  constexpr int KH = HD / 16;             // d-chunks per half tile
  u32x4 ka[KH], kb[KH];
#define IXT_KLOAD(DST_, F_)                                                                                       \
  do {                                                                                                            \
    const int ps_ = wave + 4 * ((F_) >> 1), h_ = (F_) & 1;                                                        \
    const int pidx_ = ps_ * ppw + lane / g.P, j_ = pidx_ * g.P + lane % g.P;                                      \
    if (ps_ * ppw < pages && j_ < kv_len && (!NEWTOK || j_ != pos)) {                                             \
      const bf16_t* kp_ = a.kpages + kpage_off(g, prow[pidx_], kvh, h_ * KH, lane % g.P);                         \
      _Pragma("unroll") for (int dc = 0; dc < KH; ++dc) DST_[dc] = *reinterpret_cast<const u32x4*>(kp_ + (size_t)dc * g.P * 8); \
    }                                                                                                             \
  } while (0)
  IXT_KLOAD(ka, 0);

  if constexpr (NEWTOK) {
  // prologue: q heads, new k, new v of this kv head from the QKV slices (+ RoPE), K/V page write
  for (int idx = tid; idx < (GQ + 2) * half; idx += 256) {
    const int which = idx / half, i = idx % half;   // 0..GQ-1: q heads, GQ: k, GQ+1: v
    int col0;
    if (which < GQ) col0 = (kvh * GQ + which) * HD;
    else if (which == GQ) col0 = QN + kvh * HD;
    else col0 = QN + KN + kvh * HD;
    const float* pq = a.part + (size_t)r * a.ncols + col0;
    const bool rope_ = which <= GQ;   // q and k heads are stored pair-interleaved (qk_col), v heads in feature order
    float v1 = rbf(sum_slices<16>(pq + (rope_ ? qk_col(i, half) : i), a.slice_stride, a.slices));
    float v2 = rbf(sum_slices<16>(pq + (rope_ ? qk_col(half + i, half) : half + i), a.slice_stride, a.slices));
    if (which <= GQ) {
      float o1, o2;
      ixt_rope_pair(g.rope32, a.rope_sin, a.rope_cos, (size_t)pos * half + i, v1, v2, o1, o2);
      v1 = o1; v2 = o2;
    }
    const int page = prow[pos / g.P], tok = pos % g.P;
    const int f1 = ixt_qk_feat(g.rope32, i, half), f2 = ixt_qk_feat(g.rope32, half + i, half);   // (query / key rows only)
    if (which < GQ) {
      qs[which * HD + f1] = v1; qs[which * HD + f2] = v2;
    } else if (which == GQ) {
      kn[f1] = v1; kn[f2] = v2;
      a.kpages[kpage_off(g, page, kvh, f1 / 8, tok) + (f1 & 7)] = f2bf(v1);
      a.kpages[kpage_off(g, page, kvh, f2 / 8, tok) + (f2 & 7)] = f2bf(v2);
    } else {
      vn[i] = v1; vn[half + i] = v2;
      bf16_t* vd = a.vpages + vpage_off(g, page, kvh, tok);
      vd[i] = f2bf(v1); vd[half + i] = f2bf(v2);
    }
  }
  } else {
    const bf16_t* qsrc = a.qbuf + ((size_t)r * g.nq + kvh * GQ) * HD;
    for (int i = tid; i < GQ * HD; i += 256) qs[i] = bf2f(qsrc[i]);
  }
  __syncthreads();

  // scores: lane = key; one fma chain over d per (key, q head), continued across the two halves
#define IXT_KHALF(BUF_, F_)                                                                                       \
  do {                                                                                                            \
    const int ps_ = wave + 4 * ((F_) >> 1), h_ = (F_) & 1;                                                        \
    const int j_ = (ps_ * ppw + lane / g.P) * g.P + lane % g.P;                                                   \
    if (h_ == 0) { _Pragma("unroll") for (int q = 0; q < GQ; ++q) acc[q] = 0.0f; }                                \
    if (j_ < kv_len) {                                                                                            \
      if (NEWTOK && j_ == pos) {   /* the key this block has just written: from LDS */                            \
        for (int d = h_ * (HD / 2); d < (h_ + 1) * (HD / 2); ++d) {                                               \
          _Pragma("unroll") for (int q = 0; q < GQ; ++q) acc[q] = __builtin_fmaf(qs[q * HD + d], kn[d], acc[q]);  \
        }                                                                                                         \
      } else {                                                                                                    \
        _Pragma("unroll") for (int dc = 0; dc < KH; ++dc) {                                                       \
          const float kf[8] = {bflo(BUF_[dc].x), bfhi(BUF_[dc].x), bflo(BUF_[dc].y), bfhi(BUF_[dc].y),            \
                               bflo(BUF_[dc].z), bfhi(BUF_[dc].z), bflo(BUF_[dc].w), bfhi(BUF_[dc].w)};           \
          _Pragma("unroll") for (int q = 0; q < GQ; ++q) {                                                        \
            const float4 qa = *reinterpret_cast<const float4*>(qs + q * HD + (h_ * KH + dc) * 8);                 \
            const float4 qb = *reinterpret_cast<const float4*>(qs + q * HD + (h_ * KH + dc) * 8 + 4);             \
            float sacc = acc[q];                                                                                  \
            sacc = __builtin_fmaf(qa.x, kf[0], sacc); sacc = __builtin_fmaf(qa.y, kf[1], sacc);                   \
            sacc = __builtin_fmaf(qa.z, kf[2], sacc); sacc = __builtin_fmaf(qa.w, kf[3], sacc);                   \
            sacc = __builtin_fmaf(qb.x, kf[4], sacc); sacc = __builtin_fmaf(qb.y, kf[5], sacc);                   \
            sacc = __builtin_fmaf(qb.z, kf[6], sacc); sacc = __builtin_fmaf(qb.w, kf[7], sacc);                   \
            acc[q] = sacc;                                                                                        \
          }                                                                                                       \
        }                                                                                                         \
      }                                                                                                           \
      if (h_ == 1) { _Pragma("unroll") for (int q = 0; q < GQ; ++q) ebuf[(size_t)q * tp + j_] = acc[q]; }     \
    }                                                                                                             \
  } while (0)
  {
    const int ntile = (pages > wave * ppw) ? (pages - wave * ppw + 4 * ppw - 1) / (4 * ppw) : 0;
    float acc[GQ];
    for (int f = 0; f < 2 * ntile; f += 2) {
      IXT_KLOAD(kb, f + 1);
      IXT_KHALF(ka, f);
      IXT_KLOAD(ka, f + 2);
      IXT_KHALF(kb, f + 1);
    }
  }
#undef IXT_KHALF
#undef IXT_KLOAD
  __syncthreads();

  constexpr int LPP = HD / 8;             // lanes per page (d-chunks)
  constexpr int PPW = 64 / LPP;           // pages a wave covers at once (4 for hd 128, 8 for hd 64)
  const int tq = lane / LPP, dc = lane % LPP;
  // PV work of this wave as a flat list of batches: group i = pages pg0 = (wave + 4 i) PPW .. + PPW - 1 (lane: page pg0 + tq,
  // d-chunk dc), batch bt = tokens 8 bt .. 8 bt + 7 of the lane's page.  The loads of batch f + 1 are issued before the fma
  // chains of batch f, and batch 0 before the softmax of the head: V does not depend on it.
  const int nb = g.P / 8;                                              // batches per page
  const int ngrp = (pages > wave * PPW) ? (pages - wave * PPW + 4 * PPW - 1) / (4 * PPW) : 0;
  const int nf = ngrp * nb;
#define IXT_VLOAD(BUF_, F_)                                                                                       \
  do {                                                                                                            \
    const int f_ = (F_);                                                                                          \
    if (f_ < nf) {                                                                                                \
      const int p_ = (wave + 4 * (f_ / nb)) * PPW + tq, t0_ = (f_ % nb) * 8;                                      \
      if (p_ < pages) {                                                                                           \
        const bf16_t* vp_ = a.vpages + vpage_off(g, prow[p_], kvh, t0_) + dc * 8;                                 \
        const int nt_ = min(g.P, kv_len - p_ * g.P) - t0_;                                                        \
        _Pragma("unroll") for (int u = 0; u < 8; ++u)                                                             \
          if (u < nt_) BUF_[u] = *reinterpret_cast<const u32x4*>(vp_ + (size_t)u * HD);                           \
      }                                                                                                           \
    }                                                                                                             \
  } while (0)
#define IXT_VBATCH(BUF_, F_)                                                                                      \
  do {                                                                                                            \
    const int f_ = (F_);                                                                                          \
    const int pg0_ = (wave + 4 * (f_ / nb)) * PPW, p_ = pg0_ + tq, bt_ = f_ % nb, t0_ = bt_ * 8;                  \
    if (bt_ == 0) { _Pragma("unroll") for (int j = 0; j < 8; ++j) acc[j] = 0.0f; }                                \
    if (p_ < pages) {                                                                                             \
      const int nt_ = min(g.P, kv_len - p_ * g.P) - t0_;                                                          \
      const float* pj_ = e + (size_t)p_ * g.P + t0_;                                                              \
      _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                             \
        if (u < nt_) {                                                                                            \
          float vf[8] = {bflo(BUF_[u].x), bfhi(BUF_[u].x), bflo(BUF_[u].y), bfhi(BUF_[u].y),                      \
                         bflo(BUF_[u].z), bfhi(BUF_[u].z), bflo(BUF_[u].w), bfhi(BUF_[u].w)};                     \
          if (NEWTOK && p_ * g.P + t0_ + u == pos) {   /* the token this block has just written: its value comes from LDS */ \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) vf[j] = vn[dc * 8 + j];                                 \
          }                                                                                                       \
          const float pv = pj_[u];                                                                                \
          _Pragma("unroll") for (int j = 0; j < 8; ++j) acc[j] = __builtin_fmaf(pv, vf[j], acc[j]);               \
        }                                                                                                         \
      }                                                                                                           \
    }                                                                                                             \
    if (bt_ == nb - 1) {   /* group complete: super-block sums (pages 4s..4s+3 live in lane groups tq = 4s..4s+3) */ \
      const int sbl = tq / IXT_SB, base_lane = sbl * IXT_SB * LPP + dc;                                           \
      float ss[8];                                                                                                \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) ss[j] = 0.0f;                                                 \
      _Pragma("unroll") for (int i = 0; i < IXT_SB; ++i) {                                                        \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) ss[j] = ss[j] + __shfl(acc[j], base_lane + i * LPP, 64);    \
      }                                                                                                           \
      const int sb = pg0_ / IXT_SB + sbl;                                                                         \
      if (tq % IXT_SB == 0 && sb < nsb) {                                                                         \
        float* dst = sbs + (size_t)sb * HD + dc * 8;                                                              \
        *reinterpret_cast<float4*>(dst) = make_float4(ss[0], ss[1], ss[2], ss[3]);                                \
        *reinterpret_cast<float4*>(dst + 4) = make_float4(ss[4], ss[5], ss[6], ss[7]);                            \
      }                                                                                                           \
    }                                                                                                             \
  } while (0)
  for (int q = 0; q < GQ; ++q) {
    float* e = ebuf + (size_t)q * tp;
    u32x4 va[8], vb[8];
    IXT_VLOAD(va, 0);
    // max (order independent), exp, page sums in key order, l in page order, P
    float m = -INFINITY;
    for (int j = tid; j < kv_len; j += 256) m = fmaxf(m, e[j]);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if (lane == 0) wred[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]));
    for (int j = tid; j < kv_len; j += 256) e[j] = ixt_exp(e[j] - m);
    __syncthreads();
    // page sums in key order by the page's owner thread: the page's values are fetched 8 at a time (two 16-byte LDS reads in
    // flight) and added one after the other -- same order as a scalar loop, a tenth of its LDS round trips
    for (int p = tid; p < pages; p += 256) {
      float sacc = 0.0f;
      const float* ep = e + (size_t)p * g.P;
      const int n = min(g.P, kv_len - p * g.P);
      int t0 = 0;
      for (; t0 + 8 <= n; t0 += 8) {
        const float4 x0 = *reinterpret_cast<const float4*>(ep + t0), x1 = *reinterpret_cast<const float4*>(ep + t0 + 4);
        sacc = sacc + x0.x; sacc = sacc + x0.y; sacc = sacc + x0.z; sacc = sacc + x0.w;
        sacc = sacc + x1.x; sacc = sacc + x1.y; sacc = sacc + x1.z; sacc = sacc + x1.w;
      }
      for (; t0 < n; ++t0) sacc = sacc + ep[t0];
      psum[p] = sacc;
    }
    __syncthreads();
    float l = 0.0f;
    {
      int p = 0;
      for (; p + 4 <= pages; p += 4) {
        const float4 x0 = *reinterpret_cast<const float4*>(psum + p);
        l = l + x0.x; l = l + x0.y; l = l + x0.z; l = l + x0.w;
      }
      for (; p < pages; ++p) l = l + psum[p];
    }
    for (int j = tid; j < kv_len; j += 256) e[j] = rbf(e[j] / l);
    __syncthreads();
    float acc[8];
    for (int f = 0; f < nf; f += 2) {
      IXT_VLOAD(vb, f + 1);
      IXT_VBATCH(va, f);
      if (f + 1 < nf) {
        IXT_VLOAD(va, f + 2);
        IXT_VBATCH(vb, f + 1);
      }
    }
    __syncthreads();
    if (tid < HD) {
      float t = 0.0f;
      for (int sb = 0; sb < nsb; ++sb) t = t + sbs[(size_t)sb * HD + tid];
      a.out[xidx(r, (kvh * GQ + q) * HD + tid, n_out)] = f2bf(t);
    }
    __syncthreads();
  }
#undef IXT_VLOAD
#undef IXT_VBATCH
}

// ---- the same one-launch decode attention with ONE WAVE per (row, kv head) pair and no block barrier at all: a block is just
//      four independent waves.  Every phase is wave-local (max by shuffles, page sums by the page's owner lane, the in-order
//      super-block sum kept in registers because one wave meets the super-blocks in order), so waves drift apart instead of
//      marching through load-free phases in lockstep, and 16 pairs per CU are resident instead of 4.
//      grid = ceil(R * nkv / 4), block 256; LDS per wave: (GQ + 2) * HD + GQ * Tmax + mpg floats.
template <int HD, int GQ>
__global__ __launch_bounds__(256) void k_attn_decode_w(DecAttnArgs a, int npairs) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const AttnGeom& g = a.g;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int pair = blockIdx.x * 4 + wave;
  if (pair >= npairs) return;
  const int r = pair / g.nkv, kvh = pair % g.nkv;
  const int n_out = g.nq * HD;
  constexpr int LPP = HD / 8;             // lanes per page in the PV phase (d-chunks)
  constexpr int PPW = 64 / LPP;           // pages per PV group (4 for hd 128, 8 for hd 64)
  const int tq = lane / LPP, dc = lane % LPP;
  if (!a.row_active[r]) {
    if (tq == 0) {
#pragma unroll
      for (int q = 0; q < GQ; ++q) *reinterpret_cast<u32x4*>(a.out + xidx(r, (kvh * GQ + q) * HD + dc * 8, n_out)) = u32x4(0);
    }
    return;
  }
  const size_t per_wave = (size_t)(GQ + 2) * HD + (size_t)GQ * g.Tmax + g.mpg;
  float* qs = sm + (size_t)wave * per_wave;     // [GQ][HD]
  float* kn = qs + GQ * HD;                     // [HD] new key (bf16-rounded, as f32)
  float* vn = kn + HD;                          // [HD] new value
  float* ebuf = vn + HD;                        // [GQ][Tmax] scores -> exps -> P
  float* psum = ebuf + (size_t)GQ * g.Tmax;     // [mpg]
  const int pos = a.row_pos[r];
  const int kv_len = pos + 1;
  const int pages = (kv_len + g.P - 1) / g.P;
  const int ppw = 64 / g.P;                     // pages per score tile (lane = key)
  constexpr int half = HD / 2;
  const int32_t* prow = a.page_map + (size_t)a.row_slot[r] * g.mpg;
  const int QN = g.nq * HD, KN = g.nkv * HD;

  constexpr int KH = HD / 16;             // d-chunks per half tile
  u32x4 ka[KH], kb[KH];
#define IXT_KLOAD(DST_, F_)                                                                                       \
  do {                                                                                                            \
    const int ps_ = (F_) >> 1, h_ = (F_) & 1;                                                                     \
    const int pidx_ = ps_ * ppw + lane / g.P, j_ = pidx_ * g.P + lane % g.P;                                      \
    if (ps_ * ppw < pages && j_ < kv_len && j_ != pos) {                                                          \
      const bf16_t* kp_ = a.kpages + kpage_off(g, prow[pidx_], kvh, h_ * KH, lane % g.P);                         \
      _Pragma("unroll") for (int d8 = 0; d8 < KH; ++d8) DST_[d8] = *reinterpret_cast<const u32x4*>(kp_ + (size_t)d8 * g.P * 8); \
    }                                                                                                             \
  } while (0)
  IXT_KLOAD(ka, 0);

  // prologue: q heads, new k, new v of this kv head from the QKV slices (+ RoPE), K/V page write
  {
    const int page = prow[pos / g.P], tok = pos % g.P;
    for (int idx = lane; idx < (GQ + 2) * half; idx += 64) {
      const int which = idx / half, i = idx % half;   // 0..GQ-1: q heads, GQ: k, GQ+1: v
      int col0;
      if (which < GQ) col0 = (kvh * GQ + which) * HD;
      else if (which == GQ) col0 = QN + kvh * HD;
      else col0 = QN + KN + kvh * HD;
      const float* pq = a.part + (size_t)r * a.ncols + col0;
      const bool rope_ = which <= GQ;
      float v1 = rbf(sum_slices<16>(pq + (rope_ ? qk_col(i, half) : i), a.slice_stride, a.slices));
      float v2 = rbf(sum_slices<16>(pq + (rope_ ? qk_col(half + i, half) : half + i), a.slice_stride, a.slices));
      if (which <= GQ) {
        float o1, o2;
        ixt_rope_pair(g.rope32, a.rope_sin, a.rope_cos, (size_t)pos * half + i, v1, v2, o1, o2);
        v1 = o1; v2 = o2;
      }
      const int f1 = ixt_qk_feat(g.rope32, i, half), f2 = ixt_qk_feat(g.rope32, half + i, half);   // (query / key rows only)
      if (which < GQ) {
        qs[which * HD + f1] = v1; qs[which * HD + f2] = v2;
      } else if (which == GQ) {
        kn[f1] = v1; kn[f2] = v2;
        a.kpages[kpage_off(g, page, kvh, f1 / 8, tok) + (f1 & 7)] = f2bf(v1);
        a.kpages[kpage_off(g, page, kvh, f2 / 8, tok) + (f2 & 7)] = f2bf(v2);
      } else {
        vn[i] = v1; vn[half + i] = v2;
        bf16_t* vd = a.vpages + vpage_off(g, page, kvh, tok);
        vd[i] = f2bf(v1); vd[half + i] = f2bf(v2);
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // LDS writes of this wave before its own cross-lane reads
  __builtin_amdgcn_wave_barrier();

  // scores: lane = key; one fma chain over d per (key, q head), continued across the two halves; running max per lane
  float mx[GQ];
#pragma unroll
  for (int q = 0; q < GQ; ++q) mx[q] = -INFINITY;
#define IXT_KHALF(BUF_, F_)                                                                                       \
  do {                                                                                                            \
    const int ps_ = (F_) >> 1, h_ = (F_) & 1;                                                                     \
    const int j_ = (ps_ * ppw + lane / g.P) * g.P + lane % g.P;                                                   \
    if (h_ == 0) { _Pragma("unroll") for (int q = 0; q < GQ; ++q) acc[q] = 0.0f; }                                \
    if (j_ < kv_len) {                                                                                            \
      if (j_ == pos) {   /* the key this wave has just written: from LDS */                                       \
        for (int d = h_ * (HD / 2); d < (h_ + 1) * (HD / 2); ++d) {                                               \
          _Pragma("unroll") for (int q = 0; q < GQ; ++q) acc[q] = __builtin_fmaf(qs[q * HD + d], kn[d], acc[q]);  \
        }                                                                                                         \
      } else {                                                                                                    \
        _Pragma("unroll") for (int d8 = 0; d8 < KH; ++d8) {                                                       \
          const float kf[8] = {bflo(BUF_[d8].x), bfhi(BUF_[d8].x), bflo(BUF_[d8].y), bfhi(BUF_[d8].y),            \
                               bflo(BUF_[d8].z), bfhi(BUF_[d8].z), bflo(BUF_[d8].w), bfhi(BUF_[d8].w)};           \
          _Pragma("unroll") for (int q = 0; q < GQ; ++q) {                                                        \
            const float4 qa = *reinterpret_cast<const float4*>(qs + q * HD + (h_ * KH + d8) * 8);                 \
            const float4 qb = *reinterpret_cast<const float4*>(qs + q * HD + (h_ * KH + d8) * 8 + 4);             \
            float sacc = acc[q];                                                                                  \
            sacc = __builtin_fmaf(qa.x, kf[0], sacc); sacc = __builtin_fmaf(qa.y, kf[1], sacc);                   \
            sacc = __builtin_fmaf(qa.z, kf[2], sacc); sacc = __builtin_fmaf(qa.w, kf[3], sacc);                   \
            sacc = __builtin_fmaf(qb.x, kf[4], sacc); sacc = __builtin_fmaf(qb.y, kf[5], sacc);                   \
            sacc = __builtin_fmaf(qb.z, kf[6], sacc); sacc = __builtin_fmaf(qb.w, kf[7], sacc);                   \
            acc[q] = sacc;                                                                                        \
          }                                                                                                       \
        }                                                                                                         \
      }                                                                                                           \
      if (h_ == 1) {                                                                                              \
        _Pragma("unroll") for (int q = 0; q < GQ; ++q) { ebuf[(size_t)q * g.Tmax + j_] = acc[q]; mx[q] = fmaxf(mx[q], acc[q]); } \
      }                                                                                                           \
    }                                                                                                             \
  } while (0)
  {
    const int ntile = (pages + ppw - 1) / ppw;
    float acc[GQ];
    for (int f = 0; f < 2 * ntile; f += 2) {
      IXT_KLOAD(kb, f + 1);
      IXT_KHALF(ka, f);
      IXT_KLOAD(ka, f + 2);
      IXT_KHALF(kb, f + 1);
    }
  }
#undef IXT_KHALF
#undef IXT_KLOAD

  const int nb = g.P / 8;                                              // batches per page
  const int ngrp = (pages + PPW - 1) / PPW;
  const int nf = ngrp * nb;
#define IXT_VLOAD(BUF_, F_)                                                                                       \
  do {                                                                                                            \
    const int f_ = (F_);                                                                                          \
    if (f_ < nf) {                                                                                                \
      const int p_ = (f_ / nb) * PPW + tq, t0_ = (f_ % nb) * 8;                                                   \
      if (p_ < pages) {                                                                                           \
        const bf16_t* vp_ = a.vpages + vpage_off(g, prow[p_], kvh, t0_) + dc * 8;                                 \
        const int nt_ = min(g.P, kv_len - p_ * g.P) - t0_;                                                        \
        _Pragma("unroll") for (int u = 0; u < 8; ++u)                                                             \
          if (u < nt_) BUF_[u] = *reinterpret_cast<const u32x4*>(vp_ + (size_t)u * HD);                           \
      }                                                                                                           \
    }                                                                                                             \
  } while (0)
#define IXT_VBATCH(BUF_, F_)                                                                                      \
  do {                                                                                                            \
    const int f_ = (F_);                                                                                          \
    const int p_ = (f_ / nb) * PPW + tq, bt_ = f_ % nb, t0_ = bt_ * 8;                                            \
    if (bt_ == 0) { _Pragma("unroll") for (int j = 0; j < 8; ++j) acc[j] = 0.0f; }                                \
    if (p_ < pages) {                                                                                             \
      const int nt_ = min(g.P, kv_len - p_ * g.P) - t0_;                                                          \
      const float* pj_ = e + (size_t)p_ * g.P + t0_;                                                              \
      _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                             \
        if (u < nt_) {                                                                                            \
          float vf[8] = {bflo(BUF_[u].x), bfhi(BUF_[u].x), bflo(BUF_[u].y), bfhi(BUF_[u].y),                      \
                         bflo(BUF_[u].z), bfhi(BUF_[u].z), bflo(BUF_[u].w), bfhi(BUF_[u].w)};                     \
          if (p_ * g.P + t0_ + u == pos) {   /* the token this wave has just written: its value comes from LDS */ \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) vf[j] = vn[dc * 8 + j];                                 \
          }                                                                                                       \
          const float pv = pj_[u];                                                                                \
          _Pragma("unroll") for (int j = 0; j < 8; ++j) acc[j] = __builtin_fmaf(pv, vf[j], acc[j]);               \
        }                                                                                                         \
      }                                                                                                           \
    }                                                                                                             \
    if (bt_ == nb - 1) {   /* group complete: its PPW/4 super-blocks, in order, onto the running total */         \
      _Pragma("unroll") for (int sbl = 0; sbl < PPW / IXT_SB; ++sbl) {                                            \
        float ss[8];                                                                                              \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) ss[j] = 0.0f;                                               \
        _Pragma("unroll") for (int i = 0; i < IXT_SB; ++i) {                                                      \
          _Pragma("unroll") for (int j = 0; j < 8; ++j) ss[j] = ss[j] + __shfl(acc[j], (sbl * IXT_SB + i) * LPP + dc, 64); \
        }                                                                                                         \
        if ((f_ / nb) * (PPW / IXT_SB) + sbl < nsb) {                                                             \
          _Pragma("unroll") for (int j = 0; j < 8; ++j) total[j] = total[j] + ss[j];                              \
        }                                                                                                         \
      }                                                                                                           \
    }                                                                                                             \
  } while (0)
  const int nsb = (pages + IXT_SB - 1) / IXT_SB;
  for (int q = 0; q < GQ; ++q) {
    float* e = ebuf + (size_t)q * g.Tmax;
    u32x4 va[8], vb[8];
    IXT_VLOAD(va, 0);
    // max (order independent), exp, page sums in key order by the page's owner lane, l in page order, P
    float m = mx[q];
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    for (int j = lane; j < kv_len; j += 64) e[j] = ixt_exp(e[j] - m);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int p = lane; p < pages; p += 64) {
      float sacc = 0.0f;
      const int e1 = min((p + 1) * g.P, kv_len);
      for (int j = p * g.P; j < e1; ++j) sacc = sacc + e[j];
      psum[p] = sacc;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    float l = 0.0f;
    for (int p = 0; p < pages; ++p) l = l + psum[p];
    for (int j = lane; j < kv_len; j += 64) e[j] = rbf(e[j] / l);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    float acc[8], total[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) total[j] = 0.0f;
    for (int f = 0; f < nf; f += 2) {
      IXT_VLOAD(vb, f + 1);
      IXT_VBATCH(va, f);
      if (f + 1 < nf) {
        IXT_VLOAD(va, f + 2);
        IXT_VBATCH(vb, f + 1);
      }
    }
    if (tq == 0) {
      u32x4 ov;
      ov.x = pack2(f2bf(total[0]), f2bf(total[1])); ov.y = pack2(f2bf(total[2]), f2bf(total[3]));
      ov.z = pack2(f2bf(total[4]), f2bf(total[5])); ov.w = pack2(f2bf(total[6]), f2bf(total[7]));
      *reinterpret_cast<u32x4*>(a.out + xidx(r, (kvh * GQ + q) * HD + dc * 8, n_out)) = ov;
    }
    __builtin_amdgcn_wave_barrier();   // e / psum are reused by the next q head
  }
#undef IXT_VLOAD
#undef IXT_VBATCH
}

// ---- prefill rows, tiled: block = (kv head, NQ / GQ consecutive rows).  Consecutive prefill rows belong to one page group, so their
//      keys and values are the same pages: a tile loads every K row once (lane = key, the whole key unpacked to f32 registers) and runs
//      the NQ query vectors over it, and every V batch once for NQ accumulator sets -- the per-row kernel above re-reads the group's
//      pages for every row.  Same operation order per query vector as k_attn_decode: scores one fma chain over d, max, exp, page sums in
//      key order, l in page order, P = bf16(e / l), PV page chains -> super-blocks of 4 pages in order -> super-blocks in order.
//      Scores: the query operand of the chains is wave-uniform, so it comes by scalar loads (bf16 pairs expanded on the scalar unit, the
//      next 32 d of both queries in flight while the current 32 are multiplied) -- no LDS traffic, no vector registers -- and two query
//      vectors run as the two halves of packed f32 fmas.  Softmax: one wave per four query vectors, their phases interleaved.
//      PV: lane = (page, 4 d), page chains for all NQ vectors in registers; the page sums meet in LDS, where the thread that owns
//      (vector, d) adds them up four pages at a time and in order.  Rows of a tile that sit in different page groups (a tile across the
//      border of two packed prompts) are worked off group by group.  grid = (nkv, ceil(R / TQ)), block 256;
//      LDS = NQ x lmax scores + 4 x mpg page sums + (256 / (HD / 4)) x NQ / 2 x HD page chains.
struct TileAttnArgs {
  const int32_t* row_slot; const int32_t* row_pos; const int32_t* row_active; const int32_t* page_map;
  const bf16_t* qbuf;                 // [R][nq][hd] finalised queries
  const bf16_t* kpages; const bf16_t* vpages;
  bf16_t* out;                        // [R][nq*hd] in xidx layout
  AttnGeom g;
  int R;
  int lmax;                           // keys one score row holds (multiple of 4); a group reaching further is left to the caller's guard
  float cap;                          // > 0: score = ixt_tanh(score / cap) * cap before the max (attn_logits_soft_cap, attentions.py:1257-1259), as k_attn_pv
  int window;                         // > 0: keys j <= pos - window are invisible (LOCAL_SLIDING, attentions.py:625-631): e_j = p_j = +0, every sum unchanged
};
typedef uint32_t ixt_u32x16 __attribute__((ext_vector_type(16)));
typedef float ixt_f32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) ixt_u32x16* ixt_cq16p;   // constant address space: uniform reads become scalar loads

template <int HD, int GQ, int NQ>
__global__ __launch_bounds__(256, 2) void k_attn_tile(TileAttnArgs a) {
  constexpr int TQ = NQ / GQ;             // rows of a tile
  constexpr int LPG = HD / 4;             // PV: lanes per page (each 4 d)
  constexpr int PPW = 64 / LPG;           // PV: pages a wave covers at once
  constexpr int PPB = 4 * PPW;            // pages of a pass
  constexpr int SBB = PPB / IXT_SB;       // super-blocks of a pass
  constexpr int NQH = NQ / 2;             // the page chains go through LDS in two halves of the query vectors
  constexpr int NT = NQH * HD / 256;      // (query vector, d) totals per thread and half
  constexpr int NH = HD > 128 ? 2 : 1;    // head_dim 256: the key goes through the registers in two halves of d, the chain's partial sum waits in the score row
  constexpr int HDH = HD / NH;
  constexpr int NP = HDH / 32;            // a query vector's half = NP scalar loads of 16 dwords
  constexpr int QW = NQ / 4;              // query vectors per wave in the softmax phase
  static_assert(NQ % GQ == 0 && NQ % 8 == 0 && TQ <= 32 && (NQH * HD) % 256 == 0 && PPB % IXT_SB == 0, "tile geometry");
  extern __shared__ __attribute__((aligned(16))) float sm[];
  __shared__ int s_pos[TQ], s_slot[TQ];
  const AttnGeom& g = a.g;
  const int kvh = blockIdx.x, r0 = blockIdx.y * TQ;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n_out = g.nq * HD;
  const int tp = a.lmax, mpgr = (g.mpg + 3) & ~3;
  float* ebuf = sm;                                  // [NQ][tp] scores -> exps -> P
  float* psum = ebuf + (size_t)NQ * tp;              // [NQ][mpgr]... one row per query vector of the wave in flight: [4 waves][QW][mpgr]
  float* pgs = psum + (size_t)4 * QW * mpgr;         // [PPB][NQH][HD] page chains of one pass and half
  if (tid < TQ) {
    const int r = r0 + tid;
    const bool ok = r < a.R && a.row_active[r] != 0;
    s_pos[tid] = ok ? a.row_pos[r] : -1;
    s_slot[tid] = ok ? a.row_slot[r] : -1;
  }
  __syncthreads();
  unsigned pending = 0;
#pragma unroll
  for (int i = 0; i < TQ; ++i) {
    const bool act = __builtin_amdgcn_readfirstlane(s_pos[i]) >= 0;
    if (act) pending |= 1u << i;
    else if (r0 + i < a.R) {                         // idle row: zeros, as the per-row kernel writes
      for (int idx = tid; idx < GQ * HD; idx += 256) a.out[xidx(r0 + i, kvh * GQ * HD + idx, n_out)] = 0;
    }
  }
  const int tq = lane / LPG, dq = lane % LPG;
  while (pending != 0) {
    // the rows of the leader's page group
    const int lead = __builtin_ctz(pending);
    const int slot = __builtin_amdgcn_readfirstlane(s_slot[lead]);
    unsigned grp = 0;
    int posmax = -1;
#pragma unroll
    for (int i = 0; i < TQ; ++i) {
      const int ps_ = __builtin_amdgcn_readfirstlane(s_pos[i]), sl_ = __builtin_amdgcn_readfirstlane(s_slot[i]);
      const bool in = ((pending >> i) & 1u) && sl_ == slot && ps_ < tp;
      if (in) { grp |= 1u << i; posmax = max(posmax, ps_); }
    }
    if (((grp >> lead) & 1u) == 0) {                 // the leader reaches beyond the score rows (the caller's max_len guard prevents it): zeros
      for (int idx = tid; idx < GQ * HD; idx += 256) a.out[xidx(r0 + lead, kvh * GQ * HD + idx, n_out)] = 0;
      pending &= ~(1u << lead);
      continue;
    }
    pending &= ~grp;
    const int kv_all = posmax + 1, kv4 = min((kv_all + 3) & ~3, tp);
    const int pages_all = (kv_all + g.P - 1) / g.P;
    const int32_t* prow = a.page_map + (size_t)slot * g.mpg;

    // ---- scores: lane = key, the key's HD values in registers; two query vectors per pass over them, one fma chain over d each
    const int ntile = (kv_all + 63) >> 6;
    for (int t = wave; t < ntile; t += 4) {
      const int j = t * 64 + lane;
#define IXT_QPTR(QV_) ((ixt_cq16p)(a.qbuf + ((size_t)min(r0 + (QV_) / GQ, a.R - 1) * g.nq + kvh * GQ + (QV_) % GQ) * HD))
#pragma unroll
      for (int hh = 0; hh < NH; ++hh) {
        ixt_f32x2 kf[HDH / 2];                           // (k[2c], k[2c+1]): one packed fma takes either half for both of its lanes (op_sel)
        if (j < kv_all) {
          const bf16_t* kp = a.kpages + kpage_off(g, prow[j / g.P], kvh, hh * (HDH / 8), j % g.P);
#pragma unroll
          for (int c = 0; c < HDH / 8; ++c) {
            const u32x4 kr = *reinterpret_cast<const u32x4*>(kp + (size_t)c * g.P * 8);
            kf[c * 4 + 0] = (ixt_f32x2){bflo(kr.x), bfhi(kr.x)}; kf[c * 4 + 1] = (ixt_f32x2){bflo(kr.y), bfhi(kr.y)};
            kf[c * 4 + 2] = (ixt_f32x2){bflo(kr.z), bfhi(kr.z)}; kf[c * 4 + 3] = (ixt_f32x2){bflo(kr.w), bfhi(kr.w)};
          }
        } else {
#pragma unroll
          for (int c = 0; c < HDH / 2; ++c) kf[c] = (ixt_f32x2){0.0f, 0.0f};
        }
        // query vectors (qv, qv + 1); rows past the end of the call read the last row (their scores are never stored)
        ixt_u32x16 ca = IXT_QPTR(0)[hh * NP], cb = IXT_QPTR(1)[hh * NP];
        for (int qv = 0; qv < NQ; qv += 2) {
          ixt_f32x2 acc = {0.0f, 0.0f};
          if (hh > 0 && j < kv_all) acc = (ixt_f32x2){ebuf[(size_t)qv * tp + j], ebuf[(size_t)(qv + 1) * tp + j]};   // the chain goes on where the first half of d left it
#pragma unroll
          for (int part = 0; part < NP; ++part) {
            ixt_u32x16 na, nb;                                   // the next 32 d of both vectors (the next pair's first 32 of this half after the last part)
            if (part + 1 < NP) { na = IXT_QPTR(qv)[hh * NP + part + 1]; nb = IXT_QPTR(qv + 1)[hh * NP + part + 1]; }
            else { const int nq_ = qv + 2 < NQ ? qv + 2 : 0; na = IXT_QPTR(nq_)[hh * NP]; nb = IXT_QPTR(nq_ + 1)[hh * NP]; }
            __builtin_amdgcn_sched_barrier(0);                   // the loads are issued here, ahead of the 32 fmas that hide them
#pragma unroll
            for (int k = 0; k < 16; ++k) {
              // acc.x += qa[d] k[d], acc.y += qb[d] k[d] for d = 2k (first) and 2k + 1 (second): the key operand is one half of the
              // register pair for both lanes of the packed fma -- the compiler would copy it into a second register instead
              const ixt_f32x2 qlo = {bflo(ca[k]), bflo(cb[k])}, qhi = {bfhi(ca[k]), bfhi(cb[k])};
              asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "s"(qlo), "v"(kf[part * 16 + k]));
              asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(qhi), "v"(kf[part * 16 + k]));
            }
            ca = na; cb = nb;
          }
          if (hh + 1 < NH) {
            if (j < kv_all) { ebuf[(size_t)qv * tp + j] = acc.x; ebuf[(size_t)(qv + 1) * tp + j] = acc.y; }
          } else {
            if (a.cap > 0.0f) { acc.x = ixt_tanh(acc.x / a.cap) * a.cap; acc.y = ixt_tanh(acc.y / a.cap) * a.cap; }
            const int i0 = qv / GQ, i1 = (qv + 1) / GQ;
            const int p0 = ((grp >> i0) & 1u) ? __builtin_amdgcn_readfirstlane(s_pos[i0]) : -1;
            const int p1 = ((grp >> i1) & 1u) ? __builtin_amdgcn_readfirstlane(s_pos[i1]) : -1;
            if (j <= p0) ebuf[(size_t)qv * tp + j] = acc.x;
            if (j <= p1) ebuf[(size_t)(qv + 1) * tp + j] = acc.y;
          }
        }
      }
#undef IXT_QPTR
    }
    __syncthreads();

    // ---- softmax statistics: wave w owns query vectors w QW .. w QW + QW - 1 and walks them through the phases side by side
    {
      int kvl[QW], jv[QW];                                // keys the vector sees: [jv, kvl)
      float m[QW], l[QW];
#pragma unroll
      for (int u = 0; u < QW; ++u) {
        const int i = (wave * QW + u) / GQ;
        kvl[u] = ((grp >> i) & 1u) ? __builtin_amdgcn_readfirstlane(s_pos[i]) + 1 : 0;
        jv[u] = (a.window > 0 && kvl[u] > a.window) ? kvl[u] - a.window : 0;
        m[u] = -INFINITY;
      }
#pragma unroll
      for (int u = 0; u < QW; ++u) {
        const float* e = ebuf + (size_t)(wave * QW + u) * tp;
        for (int j = lane; j < kvl[u]; j += 64) { if (j >= jv[u]) m[u] = fmaxf(m[u], e[j]); }
      }
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
#pragma unroll
        for (int u = 0; u < QW; ++u) m[u] = fmaxf(m[u], __shfl_xor(m[u], off, 64));
      }
#pragma unroll
      for (int u = 0; u < QW; ++u) {
        float* e = ebuf + (size_t)(wave * QW + u) * tp;
        for (int j = lane; j < kvl[u]; j += 64) e[j] = j >= jv[u] ? ixt_exp(e[j] - m[u]) : 0.0f;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      // page sums in key order: one lane per (query vector, page), the wave's QW vectors side by side
      {
        int maxpages = 0;
#pragma unroll
        for (int u = 0; u < QW; ++u) maxpages = max(maxpages, (kvl[u] + g.P - 1) / g.P);
        for (int it = lane; it < QW * maxpages; it += 64) {
          const int u = it % QW, p = it / QW;
          int kvl_u = kvl[0];
#pragma unroll
          for (int k = 1; k < QW; ++k) kvl_u = u == k ? kvl[k] : kvl_u;
          const int n = min(g.P, kvl_u - p * g.P);
          if (n > 0) {
            const float* ep = ebuf + (size_t)(wave * QW + u) * tp + (size_t)p * g.P;
            float sacc = 0.0f;
            int t0 = 0;
            for (; t0 + 8 <= n; t0 += 8) {
              const float4 x0 = *reinterpret_cast<const float4*>(ep + t0), x1 = *reinterpret_cast<const float4*>(ep + t0 + 4);
              sacc = sacc + x0.x; sacc = sacc + x0.y; sacc = sacc + x0.z; sacc = sacc + x0.w;
              sacc = sacc + x1.x; sacc = sacc + x1.y; sacc = sacc + x1.z; sacc = sacc + x1.w;
            }
            for (; t0 < n; ++t0) sacc = sacc + ep[t0];
            psum[(size_t)(wave * QW + u) * mpgr + p] = sacc;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      // l in page order: lane u adds up vector u's page sums, every lane takes its copy
      {
        float lv = 0.0f;
        if (lane < QW) {
          int kvl_u = kvl[0];
#pragma unroll
          for (int k = 1; k < QW; ++k) kvl_u = lane == k ? kvl[k] : kvl_u;
          const float* ps = psum + (size_t)(wave * QW + lane) * mpgr;
          const int pages_q = (kvl_u + g.P - 1) / g.P;
          int p = 0;
          for (; p + 4 <= pages_q; p += 4) {
            const float4 x0 = *reinterpret_cast<const float4*>(ps + p);
            lv = lv + x0.x; lv = lv + x0.y; lv = lv + x0.z; lv = lv + x0.w;
          }
          for (; p < pages_q; ++p) lv = lv + ps[p];
        }
#pragma unroll
        for (int u = 0; u < QW; ++u) l[u] = __shfl(lv, u, 64);
      }
#pragma unroll
      for (int u = 0; u < QW; ++u) {
        float* e = ebuf + (size_t)(wave * QW + u) * tp;
        for (int j = lane; j < kvl[u]; j += 64) e[j] = rbf(e[j] / l[u]);
        for (int j = kvl[u] + lane; j < kv4; j += 64) e[j] = 0.0f;   // the tile's keys this row does not see: weight 0 (see PV)
      }
    }
    __syncthreads();

    // ---- PV: lane = (page of the wave's group, 4 d); a V batch of 4 tokens feeds all NQ accumulator sets.  No masks: a key a row does not
    //      see (a later row's key of the same tile) has P = +0, and fma(+0, v, acc) leaves acc as it is for every finite v -- a chain
    //      that sees nothing stays +0, as the page the per-row kernel never enters.  (With an infinite or NaN V entry such a key would
    //      turn up to 15 earlier rows into NaN; the model is lost at that point anyway.)
    float tot[2 * NT];
#pragma unroll
    for (int k = 0; k < 2 * NT; ++k) tot[k] = 0.0f;
    const int npass = (pages_all + PPB - 1) / PPB;
    for (int pass = 0; pass < npass; ++pass) {
      const int pl = wave * PPW + tq, p = pass * PPB + pl;
      float acc[NQ][4];
#pragma unroll
      for (int qv = 0; qv < NQ; ++qv)
#pragma unroll
        for (int x = 0; x < 4; ++x) acc[qv][x] = 0.0f;
      if (p < pages_all) {
        const int nt = min(g.P, kv_all - p * g.P);
        const bf16_t* vbase = a.vpages + vpage_off(g, prow[p], kvh, 0) + dq * 4;
        uint2 raw[4], nxt[4];                          // the next batch's loads fly while this one is multiplied
#pragma unroll
        for (int u = 0; u < 4; ++u) raw[u] = *reinterpret_cast<const uint2*>(vbase + (size_t)min(u, nt - 1) * HD);
        for (int t0 = 0; t0 < nt; t0 += 4) {
          // unconditional (a token past the page's end re-reads its last one and is never used): a fixed number of loads in flight lets the
          // wait for the current batch leave the next one outstanding
#pragma unroll
          for (int u = 0; u < 4; ++u) nxt[u] = *reinterpret_cast<const uint2*>(vbase + (size_t)min(t0 + 4 + u, nt - 1) * HD);
          float vf[4][4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            vf[u][0] = bflo(raw[u].x); vf[u][1] = bfhi(raw[u].x); vf[u][2] = bflo(raw[u].y); vf[u][3] = bfhi(raw[u].y);
          }
          const int j0 = p * g.P + t0;
#pragma unroll
          for (int q8 = 0; q8 < NQ; q8 += 8) {           // eight P reads in flight, then their 64 packed fmas
            float4 p4[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) p4[k] = *reinterpret_cast<const float4*>(ebuf + (size_t)(q8 + k) * tp + j0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const int qv = q8 + k;
#pragma unroll
              for (int x = 0; x < 4; ++x) acc[qv][x] = __builtin_fmaf(p4[k].x, vf[0][x], acc[qv][x]);
#pragma unroll
              for (int x = 0; x < 4; ++x) acc[qv][x] = __builtin_fmaf(p4[k].y, vf[1][x], acc[qv][x]);
#pragma unroll
              for (int x = 0; x < 4; ++x) acc[qv][x] = __builtin_fmaf(p4[k].z, vf[2][x], acc[qv][x]);
#pragma unroll
              for (int x = 0; x < 4; ++x) acc[qv][x] = __builtin_fmaf(p4[k].w, vf[3][x], acc[qv][x]);
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) raw[u] = nxt[u];   // only now: the copy waits for the loads
        }
      }
      // the page chains meet in LDS, half of the query vectors at a time; the owner of (vector, d) adds four pages in order into the
      // super-block sum and the super-block sums, in order, onto its total (a page past the row's range holds zeros, as in k_attn_decode)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int k = 0; k < NQH; ++k)
          *reinterpret_cast<float4*>(pgs + ((size_t)pl * NQH + k) * HD + dq * 4) = make_float4(acc[h * NQH + k][0], acc[h * NQH + k][1], acc[h * NQH + k][2], acc[h * NQH + k][3]);
        __syncthreads();
#pragma unroll
        for (int k4 = 0; k4 < NT; ++k4) {
          const int idx = tid + 256 * k4, k = idx / HD, d = idx % HD, qv = h * NQH + k;
          const int pq = s_pos[qv / GQ];
          const int nsb_q = ((grp >> (qv / GQ)) & 1u) ? ((pq + g.P) / g.P + IXT_SB - 1) / IXT_SB : 0;
#pragma unroll
          for (int s = 0; s < SBB; ++s) {
            if (pass * SBB + s < nsb_q) {
              float ss = 0.0f;
#pragma unroll
              for (int i = 0; i < IXT_SB; ++i) ss = ss + pgs[((size_t)(s * IXT_SB + i) * NQH + k) * HD + d];
              tot[h * NT + k4] = tot[h * NT + k4] + ss;
            }
          }
        }
        __syncthreads();
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int k4 = 0; k4 < NT; ++k4) {
        const int idx = tid + 256 * k4, qv = h * NQH + idx / HD, d = idx % HD;
        if ((grp >> (qv / GQ)) & 1u) a.out[xidx(r0 + qv / GQ, (kvh * GQ + qv % GQ) * HD + d, n_out)] = f2bf(tot[h * NT + k4]);
      }
    __syncthreads();                                 // ebuf / psum / pgs are reused by the next group
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
// inference_utils.log_prob_of_chosen_token (inference_utils.py:56-63) on rows of f32 logits: (x[chosen] - max) - log(sum exp(x - max)); the sum
// in blocks of 1024 consecutive columns (one thread each, left to right), block sums added left to right.  grid = rows, block 256.
// =====================================================================================================
__global__ __launch_bounds__(256) void k_logprob(const float* __restrict__ logits, int vocab, const int32_t* __restrict__ chosen, float* __restrict__ out) {
  extern __shared__ float lp_sm[];              // [ceil(vocab / 1024)] block sums
  __shared__ float wred[4];
  const int r = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float* row = logits + (size_t)r * vocab;
  float m = -INFINITY;
  for (int n = tid; n < vocab; n += 256) m = fmaxf(m, row[n]);
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if (lane == 0) wred[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]));
  const int nblk = (vocab + 1023) / 1024;
  for (int b = tid; b < nblk; b += 256) {
    float s = 0.0f;
    const int n1 = min(vocab, (b + 1) * 1024);
    for (int n = b * 1024; n < n1; ++n) s = s + ixt_exp(row[n] - m);
    lp_sm[b] = s;
  }
  __syncthreads();
  if (tid == 0) {
    float total = 0.0f;
    for (int b = 0; b < nblk; ++b) total = total + lp_sm[b];
    out[r] = (row[chosen[r]] - m) - ixt_log(total);
  }
}

// =====================================================================================================
// RoPE on dense rows (op-level).  grid = (R, nheads), block = hd/2.
// =====================================================================================================
// rope32 (rope_type "llama3.1"): the pair of table index i is the NEIGHBOURING features (2 i, 2 i + 1) of the caller's row (embeddings.py:446-458).
__global__ void k_rope(bf16_t* x, const int32_t* pos, const bf16_t* rope_sin, const bf16_t* rope_cos, int nheads, int hd, int rope32) {
  const int r = blockIdx.x, h = blockIdx.y, i = threadIdx.x, half = hd / 2;
  if (i >= half) return;
  bf16_t* p = x + ((size_t)r * nheads + h) * hd;
  const int i1 = rope32 ? 2 * i : i, i2 = rope32 ? 2 * i + 1 : half + i;
  float o1, o2;
  ixt_rope_pair(rope32, rope_sin, rope_cos, (size_t)pos[r] * half + i, bf2f(p[i1]), bf2f(p[i2]), o1, o2);
  p[i1] = f2bf(o1);
  p[i2] = f2bf(o2);
}

// =====================================================================================================
// Row descriptors and state update.
// =====================================================================================================
// A row is computed (K/V written, keys attended) only where the page manager has granted the page of its position: a reservation that
// found no room leaves the group without pages (page_manager.py:305-311), and such rows must not write through stale page_map entries.
__global__ void k_rows_prefill(int32_t* row_slot, int32_t* row_pos, int32_t* row_active, int slot, int start, int n,
                               const int32_t* num_pages_used, int P) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { row_slot[i] = slot; row_pos[i] = start + i; row_active[i] = (start + i) / P < num_pages_used[slot] ? 1 : 0; }
}
// caller-described rows (packed prefill): slot and position per row, same page guard; out-of-range descriptors make the row inactive
// max_len: the caller's bound on position + 1 (it sizes the tiled attention's score rows); a row at or beyond it is idle and counted in
// status[3] instead of being computed against a bound the caller did not mean.
__global__ void k_rows_packed(int32_t* row_slot, int32_t* row_pos, int32_t* row_active, const int32_t* slot_in, const int32_t* pos_in, int n,
                              const int32_t* num_pages_used, int P, int groups, int Tmax, int max_len, int32_t* status) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const int g = slot_in[i], pos = pos_in[i];
    const bool ok = g >= 0 && g < groups && pos >= 0 && pos < Tmax && pos < max_len;
    if (g >= 0 && g < groups && pos >= max_len && pos < Tmax) atomicAdd(&status[3], 1);
    row_slot[i] = ok ? g : 0; row_pos[i] = ok ? pos : 0; row_active[i] = (ok && pos / P < num_pages_used[g]) ? 1 : 0;
  }
}
// Decode rows: slot i at position pos[i] (the caller's decode_state["next_pos"]).  Active only when the page manager agrees: the slot
// has an active page, the position is inside max_target_length and inside the pages granted so far (a page the pool could not grant
// deactivates the row instead of letting it write through a stale page_map entry), and -- check_len, ixt_generate -- the position is
// the one the page manager has just advanced to (sequence_lengths - 1): a slot that was prefilled but not inserted yet carries the
// previous occupant's next_pos and stays idle until MaxEngine.insert.
__global__ void k_rows_decode(int32_t* row_slot, int32_t* row_pos, int32_t* row_active, int32_t* row_tok,
                              const int32_t* tokens, const int32_t* pos, const int32_t* has_active, const int32_t* num_pages_used,
                              const int32_t* sequence_lengths, int check_len, int P, int Tmax, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) {
    const int p = pos[i];
    bool act = has_active[i] != 0 && p >= 0 && p < Tmax && p / P < num_pages_used[i];
    if (check_len) act = act && p == sequence_lengths[i] - 1;
    row_slot[i] = i; row_pos[i] = act ? p : 0; row_active[i] = act ? 1 : 0; row_tok[i] = tokens[i];
  }
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
  int32_t* status;               // [4] build extension: {failed reservations, last group that failed, decode rows left without a page, packed rows beyond max_len}
  int32_t* meta;                 // [2] bookkeeping of page_status, kept by every kernel that changes it: {free pages, an index below which no page is free}
  int num_pages, groups, mpg, P; // groups = decode slots + staging groups of packed prefill (build extension)
  int decode_groups;             // the reference's page groups (slots); only these advance in update_decode_pages
};

// Every access is an agent-scope (L1-bypassing) access followed by a wave-level drain, because the
// same wave re-reads words that its lane 0 has just written.
__device__ __forceinline__ void wave_drain() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

__device__ void wave_release_group(PageStateDev s, int g, int lane) {
  const int nvalid = ldg_i32(s.num_pages_used + g);
  int nrel = 0, lowest = 0x7fffffff;
  for (int i = lane; i < s.mpg; i += 64) {
    const int idx = ldg_i32(s.page_map + (size_t)g * s.mpg + i);
    if (i < nvalid && idx > 0) {
      if (ldg_i32(s.page_status + idx) != 0) ++nrel;     // (a page listed twice is counted once)
      stg_i32(s.page_status + idx, 0);
      lowest = min(lowest, idx);
    }
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { nrel += __shfl_xor(nrel, off, 64); lowest = min(lowest, __shfl_xor(lowest, off, 64)); }
  if (lane == 0) {
    stg_i32(s.meta + 0, ldg_i32(s.meta + 0) + nrel);
    if (lowest < ldg_i32(s.meta + 1)) stg_i32(s.meta + 1, lowest);
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
  if (lane < 4) s.status[lane] = 0;
  if (lane == 0) { s.meta[0] = s.num_pages - 1; s.meta[1] = 1; }
}
// page_status written from outside (tests): count the free pages and find the lowest one again
__global__ __launch_bounds__(64) void k_pages_recount(PageStateDev s) {
  const int lane = threadIdx.x;
  int c = 0, lowest = s.num_pages;
  for (int i = 1 + lane; i < s.num_pages; i += 64)
    if (s.page_status[i] == 0) { ++c; lowest = min(lowest, i); }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { c += __shfl_xor(c, off, 64); lowest = min(lowest, __shfl_xor(lowest, off, 64)); }
  if (lane == 0) { s.meta[0] = c; s.meta[1] = lowest; }
}
__global__ __launch_bounds__(64) void k_pages_release(PageStateDev s, int g) { wave_release_group(s, g, threadIdx.x); }

// The m lowest free pages (index >= 1) in ascending order -> out[0..m) (LDS), each marked used; returns how many exist (<= m).
// Sequential first-fit (page_manager.py:268-283, :373-395: "next free page" for one request after the other, nothing freed in
// between) hands out exactly these pages in exactly this order, so one scan serves a whole batch of requests.
// Plain loads: within one launch every status word is read before this wave may write it.
// The scan starts at meta[1] (no page below it is free: releases lower it, and the m lowest free pages having been taken, everything up to
// the last of them is in use) and meta[0] counts the free pages -- the pool of a 1024-slot engine has 41 000 status words, and a scan from
// index 0 per reservation cost 117 us.
__device__ int wave_take_free(PageStateDev s, int lane, int m, int32_t* out) {
  int32_t* status = s.page_status;
  const int num_pages = s.num_pages;
  int cnt = 0, last = -1;
  if (m <= 0) return 0;
  const int hint = ldg_i32(s.meta + 1);
  for (int base = (hint / 512) * 512; base < num_pages && cnt < m; base += 512) {
    int v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * 64 + lane;
      v[u] = (i >= 1 && i < num_pages) ? status[i] : 1;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * 64 + lane;
      const bool free_ = v[u] == 0;
      const unsigned long long mask = __ballot(free_);
      if (free_) {
        const int rank = cnt + __popcll(mask & ((1ull << lane) - 1ull));
        if (rank < m) { out[rank] = i; status[i] = 1; last = max(last, i); }
      }
      cnt += __popcll(mask);
    }
  }
  const int found = cnt < m ? cnt : m;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) last = max(last, __shfl_xor(last, off, 64));
  if (lane == 0) {
    stg_i32(s.meta + 0, ldg_i32(s.meta + 0) - found);
    stg_i32(s.meta + 1, found < m ? num_pages : last + 1);      // fewer than asked for: the pool is exhausted from here on
  }
  return found;
}
#define IXT_MAX_GROUPS 4096

// release + reserve (page_manager.py:208-328); extend != 0: keep pages, grow to `length` tokens (build extension)
__global__ __launch_bounds__(64) void k_pages_reserve(PageStateDev s, int g, int length, int extend) {
  __shared__ int32_t got[IXT_MAX_GROUPS];
  const int lane = threadIdx.x;
  if (!extend) {
    wave_release_group(s, g, lane);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");   // the freed status words (L1-bypassing stores) before the plain scans below
  }
  __syncthreads();
  const int need = (length + s.P - 1) / s.P;
  const int have = extend ? ldg_i32(s.num_pages_used + g) : 0;
  const int extra = need > have ? need - have : 0;
  const int nfree = ldg_i32(s.meta + 0);
  if (!(nfree >= extra && need <= s.mpg)) {   // the reference returns the released state silently (page_manager.py:305-311); recorded here
    if (lane == 0) { s.status[0] = s.status[0] + 1; s.status[1] = g; }
    return;
  }
  const int found = wave_take_free(s, lane, extra, got);
  __syncthreads();
  for (int k = lane; k < found; k += 64) s.page_map[(size_t)g * s.mpg + have + k] = got[k];
  __syncthreads();
  if (lane == 0) {
    s.num_pages_used[g] = have + found;
    s.sequence_lengths[g] = length;
    s.active_page[g] = (need - 1 < have) ? s.page_map[(size_t)g * s.mpg + need - 1] : got[need - 1 - have];
    s.has_active_page[g] = 1;
    s.active_page_position[g] = ((length - 1) % s.P + 1) % s.P;
  }
}

// page_manager.py:331-412: every active group advances one token; groups that cross a page boundary take the next free page,
// in group order
__global__ __launch_bounds__(64) void k_pages_advance(PageStateDev s) {
  __shared__ int32_t need[IXT_MAX_GROUPS];
  __shared__ int32_t got[IXT_MAX_GROUPS];
  const int lane = threadIdx.x;
  int m = 0;
  for (int base = 0; base < s.decode_groups; base += 64) {
    const int g = base + lane;
    bool needs = false;
    if (g < s.decode_groups) {
      const int act = s.has_active_page[g];
      const int nl = s.sequence_lengths[g] + (act ? 1 : 0);
      s.sequence_lengths[g] = nl;
      if (act) s.active_page_position[g] = (nl - 1) % s.P;
      const int required = (nl + s.P - 1) / s.P;
      needs = act && required > s.num_pages_used[g] && required <= s.mpg;
    }
    const unsigned long long mask = __ballot(needs);
    if (needs) need[m + __popcll(mask & ((1ull << lane) - 1ull))] = g;
    m += __popcll(mask);
  }
  if (m == 0) return;
  __syncthreads();
  const int found = wave_take_free(s, lane, m, got);
  __syncthreads();
  if (lane == 0 && found < m) s.status[2] = s.status[2] + (m - found);   // these rows stay idle (k_rows_decode) until a page is free
  for (int k = lane; k < found; k += 64) {
    const int gg = need[k], nf = got[k];
    const int used = s.num_pages_used[gg];
    s.page_map[(size_t)gg * s.mpg + used] = nf;
    s.num_pages_used[gg] = used + 1;
    s.active_page[gg] = nf;
  }
}
__global__ void k_pages_set_active(PageStateDev s, int g, int v) { s.has_active_page[g] = v; }

// ---- build extensions around MaxEngine.insert (no reference counterpart) ----------------------------------
// insert of a prefix of `length` tokens into the slot that already holds its pages: has_active_page = 1 (maxengine.py:1189-1191) and the
// length fields back to their post-prefill values -- a generate() between prefill(slot) and insert(slot) has advanced them.
__global__ void k_pages_resync(PageStateDev s, int g, int length) {
  const int need = (length + s.P - 1) / s.P;
  if (need > s.num_pages_used[g] || need < 1) return;           // the reservation failed: leave the (empty) group alone
  s.sequence_lengths[g] = length;
  s.active_page[g] = s.page_map[(size_t)g * s.mpg + need - 1];
  s.active_page_position[g] = ((length - 1) % s.P + 1) % s.P;
  s.has_active_page[g] = 1;
}
// hand the pages of group `src` (a staging group filled by a packed prefill) to slot `dst`: dst's old pages are released, the page list
// moves (no K/V byte is copied), src ends empty.  One wave.
__global__ __launch_bounds__(64) void k_pages_move(PageStateDev s, int src, int dst) {
  const int lane = threadIdx.x;
  wave_release_group(s, dst, lane);
  __syncthreads();
  const int used = s.num_pages_used[src];
  for (int i = lane; i < s.mpg; i += 64) s.page_map[(size_t)dst * s.mpg + i] = i < used ? s.page_map[(size_t)src * s.mpg + i] : 0;
  __syncthreads();
  if (lane == 0) {
    s.num_pages_used[dst] = used; s.sequence_lengths[dst] = s.sequence_lengths[src]; s.active_page[dst] = s.active_page[src];
    s.has_active_page[dst] = used > 0 ? 1 : 0; s.active_page_position[dst] = s.active_page_position[src];
    s.num_pages_used[src] = 0; s.sequence_lengths[src] = 0; s.active_page[src] = 0; s.has_active_page[src] = 0; s.active_page_position[src] = 0;
  }
}
// copy the K and V pages of group `src` (first `npages` logical pages) into the pages of group `dst`, all layers (the reference's page
// copy loop, maxengine.py:1106-1125).  grid = (npages, layers), block 256; kp / vp: per-layer pool pointers.
__global__ __launch_bounds__(256) void k_pages_copy(PageStateDev s, int src, int dst, bf16_t* const* kp, bf16_t* const* vp, size_t page_elems) {
  const int i = blockIdx.x, l = blockIdx.y;
  if (i >= s.num_pages_used[dst] || i >= s.num_pages_used[src]) return;
  const size_t so = (size_t)s.page_map[(size_t)src * s.mpg + i] * page_elems, d_o = (size_t)s.page_map[(size_t)dst * s.mpg + i] * page_elems;
  const u32x4* ks = reinterpret_cast<const u32x4*>(kp[l] + so);
  const u32x4* vs = reinterpret_cast<const u32x4*>(vp[l] + so);
  u32x4* kd = reinterpret_cast<u32x4*>(kp[l] + d_o);
  u32x4* vd = reinterpret_cast<u32x4*>(vp[l] + d_o);
  for (size_t e = threadIdx.x; e < page_elems / 8; e += 256) { kd[e] = ks[e]; vd[e] = vs[e]; }
}

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

#include "ixt_gemm.h"
