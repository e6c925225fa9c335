// ixt_small.h -- the decode step at FEW rows (one row block: R <= 16), five launches per layer (included by ixt_api.hip after ixt_gemm.h):
//
//   k_sdot<QKV, norm>     RMSNorm of the hidden rows (recomputed per block) -> QKV projection -> RoPE -> q buffer, K / V pages
//   k_attn_small<NW>      one block of NW (8) waves per (row, kv head), page list in LDS, the whole K and V streams one round of loads each
//   k_sdot<RESID>         output projection + residual
//   k_sdot<SWIGLU, norm>  RMSNorm -> (wi_0, wi_1) projection -> SwiGLU
//   k_sdot<RESID>         down projection + residual
//
// Why: below ~64 rows every kernel of the layer is a weight stream whose time is one memory latency plus bytes / bandwidth, and the
// split-K form of the larger batches (k_dotb: one f32 slice per 512-k chain, then slice-sum / residual / norm and SwiGLU kernels) pays
// 8 launches per layer for 68 us of kernels against 28 us of HBM time (profiles/r02_b8_by_grid.txt).  Here a block owns ALL chains of
// its column tiles (one wave per chain: every weight byte of the launch is requested in the first microsecond), the chain totals meet in
// LDS and are added in chain order by one wave, and the epilogue of ixt_gemm.h's ring GEMMs runs on chip: no slice leaves the chip.
// The arithmetic is the canonical dot's (chain c: 16 v_mfma_f32_16x16x32_bf16 over k in [512c, 512c+512) into one accumulator;
// t = 0; t += chain(c), c ascending), so the results equal the other forms' bit for bit.
//
// Activations in LDS, row-compact: piece (kc, r) = the 8 bf16 of row r at k = 8 kc .. 8 kc + 7 lives at sx[(kc << rpl) + r], Rp = 1 << rpl
// >= R rows per k-chunk (HBM holds 16-row fragments whatever R is; only the R real rows are fetched).  The MFMA's B operand of lane
// (r_in, kq) at step s is piece (4 (16 c + s) + kq, r_in mod Rp): rows r_in >= R compute garbage that is never stored (an MFMA output
// row depends on its own B column only).  ds_read_b128 of that pattern is conflict-free for every Rp (MI355X_MICROARCH.md, LDS).
#pragma once

// Weight prefetch (performance only, results untouched): extra blocks behind a launch's own grid stream the NEXT launch's kernel through the
// memory hierarchy and throw the values away.  At few rows every launch of the layer is one burst of weight loads between a start-up and a
// tail during which HBM idles (5 launches x ~5 us per layer against ~22 us of streaming); the prefetch blocks fill those gaps, and the next
// launch finds its 8 - 60 MB in the 256 MB Infinity Cache (hit latency 227 ns against 375+ ns, 1.4 x the streaming rate: MI355X_MICROARCH.md).
// Block b of nb takes the b-th slice; 8 loads of 16 bytes in flight per thread.
__device__ __forceinline__ void ixt_prefetch(const u32x4* __restrict__ p, size_t n, int b, int nb, int tid, int nthr) {
  const size_t lo = n * (size_t)b / (size_t)nb, hi = n * (size_t)(b + 1) / (size_t)nb;
  for (size_t i = lo + tid; i < hi; i += (size_t)8 * nthr) {
    u32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[min(i + (size_t)u * nthr, hi - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u) asm volatile("" ::"v"(v[u].x), "v"(v[u].y), "v"(v[u].z), "v"(v[u].w));   // the loads must complete; nothing is kept
  }
}

struct SdotArgs {
  const bf16_t* x;       // NORM = false: [K/8][16][8] (xidx layout, row block 0); NORM = true: hidden rows [R][K] row-major
  const bf16_t* scale;   // NORM: RMSNorm scale [K]
  float eps;
  const u32x4* w;        // widx layout
  int K, N, R, rpl, tpb, chains;
  GemmEpi e;
  const u32x4* pf; size_t pf_n; int main_blocks;   // prefetch: blocks main_blocks .. gridDim.x - 1 stream pf[0 .. pf_n) (16-byte units)
};

template <int EPI, bool NORM>
__global__ __launch_bounds__(1024) void k_sdot(SdotArgs a) {
  extern __shared__ __attribute__((aligned(16))) u32x4 sx[];
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63;
  if ((int)blockIdx.x >= a.main_blocks) {        // a prefetch block (whole block: no barrier is skipped)
    ixt_prefetch(a.pf, a.pf_n, (int)blockIdx.x - a.main_blocks, (int)gridDim.x - a.main_blocks, tid, nthr);
    return;
  }
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction: keeps (tile, chain) and every branch on them scalar
  const int K = a.K, N = a.N, R = a.R, tpb = a.tpb, chains = a.chains;
  const int nsteps = K >> 5, ntile = N >> 4, nch = K >> 3;
  const int rpl = a.rpl, Rp = 1 << rpl;
  const int t = wave % tpb, c = wave / tpb;
  const int tile = blockIdx.x * tpb + t;
  const bool tile_ok = tile < ntile;
  const int r_in = lane & 15, cq = lane >> 4;
  f32x4* red = reinterpret_cast<f32x4*>(sx + ((size_t)nch << rpl));     // [chains][tpb][64]
  float* blk = reinterpret_cast<float*>(red + (size_t)chains * tpb * 64);   // NORM: [R][nblk] block sums of squares (R nblk <= 240), [240 + row] = 1 / sqrt(mean + eps)
  u32x4* ssc = reinterpret_cast<u32x4*>(blk + 256);                         // NORM: the norm's scale, [K/8] pieces

  // Round 0 of the activation loads goes out BEFORE the weight fragments: loads return in order, so the (L2-resident) activations can be
  // waited for with the sixteen weight loads still in flight, and the whole prologue below runs under the weight stream.
  // Piece p = (k-chunk p >> rpl, row p & (Rp - 1)); unconditional loads from clamped addresses (rows R .. Rp - 1 repeat row R - 1 / hold
  // whatever the producer left in the fragment: never used).
  const int np = nch << rpl;
  auto xsrc = [&](int p) -> const u32x4* {
    p = min(p, np - 1);
    if constexpr (NORM) return reinterpret_cast<const u32x4*>(a.x + (size_t)min(p & (Rp - 1), R - 1) * K + (size_t)(p >> rpl) * 8);
    else return reinterpret_cast<const u32x4*>(a.x) + (size_t)(p >> rpl) * 16 + (p & (Rp - 1));
  };
  u32x4 x0[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) x0[u] = *xsrc(tid + u * nthr);

  // this wave's chain of weight fragments: all sixteen in flight
  const int s0 = c * 16, cs = min(16, nsteps - s0);
  const u32x4* wbase = a.w + ((size_t)(tile_ok ? tile : 0) * nsteps + s0) * 64 + lane;
  const bool full = tile_ok && cs == 16;      // the usual case: a whole chain, no per-step conditions anywhere
  u32x4 wv[16];
  // (a short last chain re-reads its last fragment for the steps it does not have, a block past the last tile reads tile 0: valid
  // addresses, values never used -- one unconditional load sequence for every wave)
#pragma unroll
  for (int s = 0; s < 16; ++s) wv[s] = __builtin_nontemporal_load(wbase + (size_t)min(s, cs - 1) * 64);

#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int p = tid + u * nthr;
    if (p < np) sx[p] = x0[u];
  }
#pragma unroll 1
  for (int p0 = tid + 8 * nthr; p0 < np; p0 += 4 * nthr) {   // more than 8 pieces per thread (many rows): further rounds of 4
    u32x4 tmp[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) tmp[u] = *xsrc(p0 + u * nthr);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = p0 + u * nthr;
      if (p < np) sx[p] = tmp[u];
    }
  }
  if constexpr (NORM) {
    // RMSNorm (normalizations.py:57-70) in the canonical order of k_rownorm: chunk of 8 = one fma chain, block of 64 chunks by
    // xor-butterfly, blocks added in order; y = bf16(x r), out = bf16(y scale).  Every block norms the rows for itself.
    for (int i = tid; i < nch; i += nthr) ssc[i] = *reinterpret_cast<const u32x4*>(a.scale + i * 8);
    __syncthreads();
    // sums of squares from LDS: a wave takes (row, block) pairs, lane = chunk
    const int nblk = (nch + 63) >> 6, nw = nthr >> 6;
    for (int p = wave; p < R * nblk; p += nw) {
      const int row = p / nblk, b = p - row * nblk, ch = b * 64 + lane;
      float v = 0.0f;
      if (ch < nch) {
        const u32x4 hv = sx[((size_t)ch << rpl) + row];
        const float h[8] = {bflo(hv.x), bfhi(hv.x), bflo(hv.y), bfhi(hv.y), bflo(hv.z), bfhi(hv.z), bflo(hv.w), bfhi(hv.w)};
#pragma unroll
        for (int j = 0; j < 8; ++j) v = __builtin_fmaf(h[j], h[j], v);
      }
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) v = v + __shfl_xor(v, off, 64);
      if (lane == 0) blk[row * nblk + b] = v;
    }
    __syncthreads();
    if (tid < R) {                                   // 1 / sqrt(mean + eps) once per row
      float total = 0.0f;
      for (int b = 0; b < nblk; ++b) total = total + blk[tid * nblk + b];
      blk[240 + tid] = 1.0f / sqrtf(total / (float)K + a.eps);
    }
    __syncthreads();
    for (int p = tid; p < np; p += nthr) {           // normalise in place
      const int ch = p >> rpl, row = p & (Rp - 1);
      if (row < R) {
        const float rs = blk[240 + row];
        const u32x4 hv = sx[p];
        const u32x4 sv = ssc[ch];
        const float h[8] = {bflo(hv.x), bfhi(hv.x), bflo(hv.y), bfhi(hv.y), bflo(hv.z), bfhi(hv.z), bflo(hv.w), bfhi(hv.w)};
        const float sc[8] = {bflo(sv.x), bfhi(sv.x), bflo(sv.y), bfhi(sv.y), bflo(sv.z), bfhi(sv.z), bflo(sv.w), bfhi(sv.w)};
        bf16_t o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = f2bf(rbf(h[j] * rs) * sc[j]);
        u32x4 ov;
        ov.x = pack2(o[0], o[1]); ov.y = pack2(o[2], o[3]); ov.z = pack2(o[4], o[5]); ov.w = pack2(o[6], o[7]);
        sx[p] = ov;
      }
    }
  }
  __syncthreads();

  f32x4 acc = f32x4(0.0f);
  {
    const u32x4* xb = sx + (((size_t)s0 * 4 + cq) << rpl) + (r_in & (Rp - 1));
    if (full) {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wv[s]), __builtin_bit_cast(bf16x8, xb[(size_t)(s * 4) << rpl]), acc, 0, 0, 0);
        if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // at most four operand reads ahead of their MFMAs: 64 registers hold the weight fragments
      }
    } else if (tile_ok) {
#pragma unroll
      for (int s = 0; s < 16; ++s)
        if (s < cs) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wv[s]), __builtin_bit_cast(bf16x8, xb[(size_t)(s * 4) << rpl]), acc, 0, 0, 0);
    }
  }
  red[((size_t)c * tpb + t) * 64 + lane] = acc;
  __syncthreads();
  if (c != 0 || !tile_ok) return;
  if (EPI == GEPI_SWIGLU && (t & 1)) return;        // the wi_1 tile's totals are taken by the wi_0 tile's wave
  f32x4 tot = f32x4(0.0f), tot2 = f32x4(0.0f);
  for (int cc = 0; cc < chains; ++cc) {
    tot = tot + red[((size_t)cc * tpb + t) * 64 + lane];
    if (EPI == GEPI_SWIGLU) tot2 = tot2 + red[((size_t)cc * tpb + t + 1) * 64 + lane];
  }
  const int row = r_in;
  const bool row_ok = row < R;

  if constexpr (EPI == GEPI_F32) {
    if (row_ok) *reinterpret_cast<float4*>(a.e.out + (size_t)row * N + tile * 16 + cq * 4) = make_float4(tot[0], tot[1], tot[2], tot[3]);
  } else if constexpr (EPI == GEPI_RESID) {
    if (row_ok) {
      const size_t o = (size_t)row * N + tile * 16 + cq * 4;
      const uint2 hv = *reinterpret_cast<const uint2*>(a.e.h_in + o);
      uint2 ov;
      ov.x = pack2(f2bf(bflo(hv.x) + rbf(tot[0])), f2bf(bfhi(hv.x) + rbf(tot[1])));
      ov.y = pack2(f2bf(bflo(hv.y) + rbf(tot[2])), f2bf(bfhi(hv.y) + rbf(tot[3])));
      *reinterpret_cast<uint2*>(a.e.h_out + o) = ov;
    }
  } else if constexpr (EPI == GEPI_SWIGLU) {
    if (row_ok && tile + 1 < ntile) {
      const int F = N >> 1, f0 = (tile >> 1) * 16 + cq * 4;
      bf16_t o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = ixt_swiglu(rbf(tot[j]), rbf(tot2[j]));
      uint2 ov;
      ov.x = pack2(o[0], o[1]); ov.y = pack2(o[2], o[3]);
      *reinterpret_cast<uint2*>(a.e.act + xidx(row, f0, F)) = ov;
    }
  } else if constexpr (EPI == GEPI_QKV) {
    // as GEPI_QKV of k_gemm_ring: bf16(t), RoPE by partner exchange across lanes l and l ^ 32, q -> qbuf, K / V -> the row's page
    const AttnGeom& g = a.e.g;
    const int half = g.hd >> 1, tph = g.hd >> 4;
    const int QN = g.nq * g.hd, KN = g.nkv * g.hd;
    const int pos = row_ok ? a.e.row_pos[row] : 0;
    const bool wr_kv = row_ok && a.e.row_active[row] != 0;
    int page = 0;
    if (wr_kv) page = a.e.page_map[(size_t)a.e.row_slot[row] * g.mpg + pos / g.P];
    const int tok = pos % g.P;
    const int col0 = tile * 16;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = rbf(tot[j]);
    const bool is_v = col0 >= QN + KN;
    const int hh = col0 / g.hd, th = tile % tph;
    if (!is_v) {
      const int fi = 8 * th + 4 * (cq & 1);
      if (g.rope32) {   // rope_type "llama3.1": f32 tables, one rounding (ixt_rope_pair)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float other = __shfl_xor(v[j], 32, 64);
          float o1, o2;
          ixt_rope_pair(1, a.e.rope_sin, a.e.rope_cos, (size_t)pos * half + fi + j, cq < 2 ? v[j] : other, cq < 2 ? other : v[j], o1, o2);
          v[j] = cq < 2 ? o1 : o2;
        }
      } else {
        float sn[4], cs_[4];
        {
          const uint2 sv = *reinterpret_cast<const uint2*>(a.e.rope_sin + (size_t)pos * half + fi);
          const uint2 cv = *reinterpret_cast<const uint2*>(a.e.rope_cos + (size_t)pos * half + fi);
          sn[0] = bflo(sv.x); sn[1] = bfhi(sv.x); sn[2] = bflo(sv.y); sn[3] = bfhi(sv.y);
          cs_[0] = bflo(cv.x); cs_[1] = bfhi(cv.x); cs_[2] = bflo(cv.y); cs_[3] = bfhi(cv.y);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float other = __shfl_xor(v[j], 32, 64);
          v[j] = cq < 2 ? rbf(rbf(v[j] * cs_[j]) - rbf(other * sn[j])) : rbf(rbf(v[j] * cs_[j]) + rbf(other * sn[j]));
        }
      }
    }
    if (!row_ok) return;
    uint2 ov;
    ov.x = pack2(f2bf(v[0]), f2bf(v[1])); ov.y = pack2(f2bf(v[2]), f2bf(v[3]));
    if (g.rope32 && !is_v) {   // llama3.1: the lane's four de-interleaved features are every second feature of the model's order (ixt_qk_feat)
      const int feat = (cq >= 2 ? half : 0) + 8 * th + 4 * (cq & 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int fn_ = ixt_qk_feat(1, feat + j, half);
        if (hh < g.nq) a.e.qbuf[((size_t)row * g.nq + hh) * g.hd + fn_] = f2bf(v[j]);
        else if (wr_kv) a.e.kpages[kpage_off(g, page, hh - g.nq, fn_ >> 3, tok) + (fn_ & 7)] = f2bf(v[j]);
      }
    } else if (hh < g.nq) {
      const int feat = (cq >= 2 ? half : 0) + 8 * th + 4 * (cq & 1);
      *reinterpret_cast<uint2*>(a.e.qbuf + ((size_t)row * g.nq + hh) * g.hd + feat) = ov;
    } else if (!is_v) {
      if (wr_kv) {
        const int feat = (cq >= 2 ? half : 0) + 8 * th + 4 * (cq & 1);
        *reinterpret_cast<uint2*>(a.e.kpages + kpage_off(g, page, hh - g.nq, feat >> 3, tok) + (feat & 7)) = ov;
      }
    } else if (wr_kv) {
      const int feat = th * 16 + cq * 4;
      *reinterpret_cast<uint2*>(a.e.vpages + vpage_off(g, page, hh - g.nq - g.nkv, tok) + feat) = ov;
    }
  }
}

// =====================================================================================================
// Decode attention for few rows: the one-launch form (k_attn_decode<HD, GQ, false>) for blocks that are ALONE on their CU.  With R x nkv
// blocks below the CU count a block's time is the length of ITS serial chain of memory round trips, so here
//   * the row's page list sits in LDS before the first K load (k_attn_decode looks every page up in HBM in front of its loads: one more
//     dependent round trip per tile and per V batch);
//   * NW = 8 waves (not 4) share the key range, and every wave keeps up to TWO key tiles (4 half-tile buffers) and FOUR value batches
//     in flight: at the benchmark's contexts (<= 40 pages of 32 keys) the whole K stream is one round of loads and the whole V stream
//     another, issued before the softmax between them.
// Same operation order (scores: one fma chain over d; l: page sums in key order, pages in order; PV: page chains, 4 pages per super-block
// in order, super-blocks in order).  q from qbuf, every key / value from the pages.  grid = (nkv, R), block 64 NW,
// LDS = GQ HD + GQ Tmax' + 2 mpg' + 16 + nsb_max HD floats.
// =====================================================================================================
template <int HD, int GQ, int NW>
__global__ __launch_bounds__(NW * 64) void k_attn_small(DecAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int NT = NW * 64;
  const AttnGeom& g = a.g;
  const int kvh = blockIdx.x, r = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  if (r >= a.rows) {                               // prefetch blocks behind the rows' blocks (ixt_prefetch)
    ixt_prefetch(a.pf, a.pf_n, (r - a.rows) * (int)gridDim.x + kvh, ((int)gridDim.y - a.rows) * (int)gridDim.x, tid, NT);
    return;
  }
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_out = g.nq * HD;
  if (!a.row_active[r]) {
    for (int i = tid; i < GQ * HD; i += NT) a.out[xidx(r, kvh * GQ * HD + i, n_out)] = 0;
    return;
  }
  float* qs = sm;                               // [GQ][HD]
  float* ebuf = qs + GQ * HD;                   // [GQ][Tmax'] scores -> exps -> P
  const int tp = (g.Tmax + 3) & ~3;
  const int mpgp = (g.mpg + 3) & ~3;
  float* psum = ebuf + (size_t)GQ * tp;         // [mpg']
  int32_t* pm = reinterpret_cast<int32_t*>(psum + mpgp);   // [mpg'] the row's page list
  float* wred = reinterpret_cast<float*>(pm + mpgp);       // [16]
  float* sbs = wred + 16;                       // [nsb_max][HD]
  const int pos = a.row_pos[r];
  const int kv_len = pos + 1;
  const int pages = (kv_len + g.P - 1) / g.P;
  const int nsb = (pages + IXT_SB - 1) / IXT_SB;
  const int ppw = 64 / g.P;
  {
    const int32_t* prow = a.page_map + (size_t)a.row_slot[r] * g.mpg;
    for (int i = tid; i < pages; i += NT) pm[i] = prow[i];
    const bf16_t* qsrc = a.qbuf + ((size_t)r * g.nq + kvh * GQ) * HD;
    for (int i = tid; i < GQ * HD; i += NT) qs[i] = bf2f(qsrc[i]);
  }
  __syncthreads();

  constexpr int KH = HD / 16;             // d-chunks per half tile
  u32x4 k0[KH], k1[KH], k2[KH], k3[KH];   // half tiles f, f + 1 (one key tile) and f + 2, f + 3 (the next)
#define IXT_KLOAD(DST_, F_)                                                                                       \
  do {                                                                                                            \
    const int ps_ = wave + NW * ((F_) >> 1), h_ = (F_) & 1;                                                       \
    const int pidx_ = ps_ * ppw + lane / g.P, j_ = pidx_ * g.P + lane % g.P;                                      \
    if (ps_ * ppw < pages && j_ < kv_len) {                                                                       \
      const bf16_t* kp_ = a.kpages + kpage_off(g, pm[pidx_], kvh, h_ * KH, lane % g.P);                           \
      _Pragma("unroll") for (int dc = 0; dc < KH; ++dc) DST_[dc] = *reinterpret_cast<const u32x4*>(kp_ + (size_t)dc * g.P * 8); \
    }                                                                                                             \
  } while (0)
#define IXT_KHALF(BUF_, F_)                                                                                       \
  do {                                                                                                            \
    const int ps_ = wave + NW * ((F_) >> 1), h_ = (F_) & 1;                                                       \
    const int j_ = (ps_ * ppw + lane / g.P) * g.P + lane % g.P;                                                   \
    if (h_ == 0) { _Pragma("unroll") for (int q = 0; q < GQ; ++q) acc[q] = 0.0f; }                                \
    if (j_ < kv_len) {                                                                                            \
      _Pragma("unroll") for (int dc = 0; dc < KH; ++dc) {                                                         \
        const float kf[8] = {bflo(BUF_[dc].x), bfhi(BUF_[dc].x), bflo(BUF_[dc].y), bfhi(BUF_[dc].y),              \
                             bflo(BUF_[dc].z), bfhi(BUF_[dc].z), bflo(BUF_[dc].w), bfhi(BUF_[dc].w)};             \
        _Pragma("unroll") for (int q = 0; q < GQ; ++q) {                                                          \
          const float4 qa = *reinterpret_cast<const float4*>(qs + q * HD + (h_ * KH + dc) * 8);                   \
          const float4 qb = *reinterpret_cast<const float4*>(qs + q * HD + (h_ * KH + dc) * 8 + 4);               \
          float sacc = acc[q];                                                                                    \
          sacc = __builtin_fmaf(qa.x, kf[0], sacc); sacc = __builtin_fmaf(qa.y, kf[1], sacc);                     \
          sacc = __builtin_fmaf(qa.z, kf[2], sacc); sacc = __builtin_fmaf(qa.w, kf[3], sacc);                     \
          sacc = __builtin_fmaf(qb.x, kf[4], sacc); sacc = __builtin_fmaf(qb.y, kf[5], sacc);                     \
          sacc = __builtin_fmaf(qb.z, kf[6], sacc); sacc = __builtin_fmaf(qb.w, kf[7], sacc);                     \
          acc[q] = sacc;                                                                                          \
        }                                                                                                         \
      }                                                                                                           \
      if (h_ == 1) { _Pragma("unroll") for (int q = 0; q < GQ; ++q) ebuf[(size_t)q * tp + j_] = acc[q]; }         \
    }                                                                                                             \
  } while (0)
  {
    const int ntile = (pages > wave * ppw) ? (pages - wave * ppw + NW * ppw - 1) / (NW * ppw) : 0;
    float acc[GQ];
    IXT_KLOAD(k0, 0); IXT_KLOAD(k1, 1); IXT_KLOAD(k2, 2); IXT_KLOAD(k3, 3);
    for (int f = 0; f < 2 * ntile; f += 4) {
      IXT_KHALF(k0, f);
      IXT_KHALF(k1, f + 1);
      IXT_KLOAD(k0, f + 4); IXT_KLOAD(k1, f + 5);
      if (f + 2 < 2 * ntile) {
        IXT_KHALF(k2, f + 2);
        IXT_KHALF(k3, f + 3);
        IXT_KLOAD(k2, f + 6); IXT_KLOAD(k3, f + 7);
      }
    }
  }
#undef IXT_KHALF
#undef IXT_KLOAD
  __syncthreads();

  constexpr int LPP = HD / 8;             // lanes per page (d-chunks)
  constexpr int PPW = 64 / LPP;           // pages a wave covers at once
  const int tq = lane / LPP, dc = lane % LPP;
  const int nb = g.P / 8;                 // batches of 8 tokens per page
  const int ngrp = (pages > wave * PPW) ? (pages - wave * PPW + NW * PPW - 1) / (NW * PPW) : 0;
  const int nf = ngrp * nb;
#define IXT_VLOAD(BUF_, F_)                                                                                       \
  do {                                                                                                            \
    const int f_ = (F_);                                                                                          \
    if (f_ < nf) {                                                                                                \
      const int p_ = (wave + NW * (f_ / nb)) * PPW + tq, t0_ = (f_ % nb) * 8;                                     \
      if (p_ < pages) {                                                                                           \
        const bf16_t* vp_ = a.vpages + vpage_off(g, pm[p_], kvh, t0_) + dc * 8;                                   \
        const int nt_ = min(g.P, kv_len - p_ * g.P) - t0_;                                                        \
        _Pragma("unroll") for (int u = 0; u < 8; ++u)                                                             \
          if (u < nt_) BUF_[u] = *reinterpret_cast<const u32x4*>(vp_ + (size_t)u * HD);                           \
      }                                                                                                           \
    }                                                                                                             \
  } while (0)
#define IXT_VBATCH(BUF_, F_)                                                                                      \
  do {                                                                                                            \
    const int f_ = (F_);                                                                                          \
    const int pg0_ = (wave + NW * (f_ / nb)) * PPW, p_ = pg0_ + tq, bt_ = f_ % nb, t0_ = bt_ * 8;                 \
    if (bt_ == 0) { _Pragma("unroll") for (int j = 0; j < 8; ++j) acc[j] = 0.0f; }                                \
    if (p_ < pages) {                                                                                             \
      const int nt_ = min(g.P, kv_len - p_ * g.P) - t0_;                                                          \
      const float* pj_ = e + (size_t)p_ * g.P + t0_;                                                              \
      _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                             \
        if (u < nt_) {                                                                                            \
          const float vf[8] = {bflo(BUF_[u].x), bfhi(BUF_[u].x), bflo(BUF_[u].y), bfhi(BUF_[u].y),                \
                               bflo(BUF_[u].z), bfhi(BUF_[u].z), bflo(BUF_[u].w), bfhi(BUF_[u].w)};               \
          const float pv = pj_[u];                                                                                \
          _Pragma("unroll") for (int j = 0; j < 8; ++j) acc[j] = __builtin_fmaf(pv, vf[j], acc[j]);               \
        }                                                                                                         \
      }                                                                                                           \
    }                                                                                                             \
    if (bt_ == nb - 1) {                                                                                          \
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
    u32x4 v0[8], v1[8], v2[8], v3[8];
    IXT_VLOAD(v0, 0); IXT_VLOAD(v1, 1); IXT_VLOAD(v2, 2); IXT_VLOAD(v3, 3);
    float m = -INFINITY;
    for (int j = tid; j < kv_len; j += NT) m = fmaxf(m, e[j]);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if (lane == 0) wred[wave] = m;
    __syncthreads();
    m = wred[0];
#pragma unroll
    for (int w2 = 1; w2 < NW; ++w2) m = fmaxf(m, wred[w2]);
    for (int j = tid; j < kv_len; j += NT) e[j] = ixt_exp(e[j] - m);
    __syncthreads();
    for (int p = tid; p < pages; p += NT) {
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
    for (int j = tid; j < kv_len; j += NT) e[j] = rbf(e[j] / l);
    __syncthreads();
    float acc[8];
    for (int f = 0; f < nf; f += 4) {
      IXT_VBATCH(v0, f);
      IXT_VLOAD(v0, f + 4);
      if (f + 1 < nf) { IXT_VBATCH(v1, f + 1); IXT_VLOAD(v1, f + 5); }
      if (f + 2 < nf) { IXT_VBATCH(v2, f + 2); IXT_VLOAD(v2, f + 6); }
      if (f + 3 < nf) { IXT_VBATCH(v3, f + 3); IXT_VLOAD(v3, f + 7); }
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

// =====================================================================================================
// Decode attention for VERY few rows: the keys of one (row, kv head) split over S blocks.  With R x nkv blocks far below the CU count a block of
// k_attn_small reads its head's whole K and V through ONE CU (394 KB at 770 keys: 8-9 us at the 30-47 GB/s a CU draws from HBM); here S
// blocks -- on S CUs of the same XCD: the grid is (nkv, R, S) and nkv is a multiple of 8, so the S blocks of a (row, kv head) differ by a
// multiple of 8 in dispatch order -- take a contiguous run of super-blocks each.  The operation order is the contract's, unchanged:
//   scores: one fma chain over d                                  (local)
//   m = max over ALL keys                                         -> exchange 1: the blocks' maxima
//   e = exp*(s - m); page sums in key order                       (local)
//   l = page sums added in page order                             -> exchange 2: every block's page sums (each block adds ALL of them in order)
//   p = bf16(e / l); PV: one fma chain over a page's tokens, the 4 pages of a super-block added in order   (local)
//   out = super-block sums added in order                         -> exchange 3: the LAST block to arrive adds them (and re-arms the counters)
// Exchanges go through a small global workspace per (row, kv head): agent-scope stores, a counter, agent-scope loads; the waits are bounded
// (IXT_SPIN_LIMIT polls, then the error word is set and the block goes on -- wrong numbers, which the caller reports, but never a hang).  All
// S blocks of a launch are resident at once (the host launches no more blocks than the occupancy query says fit the chip), so nobody waits for a
// block that cannot start.
// Wave w of a block owns page 4 sb + w of each of its super-blocks in both phases: scores with lane = key (P <= 64 lanes busy), PV with
// lane = 2 or 4 features (HD / FPL <= 64 lanes busy), all of a page's loads in flight at once.  Soft cap and sliding window as in k_attn_pv.
// grid = (nkv, R, S), block 256.  LDS = GQ HD + GQ Tmax' + mpg' + 32 + GQ 4 HD + GQ mpg' floats.
// =====================================================================================================
#define IXT_SPIN_LIMIT (1 << 18)
struct SplitWs {
  int32_t* cnt;     // [R][nkv][4] arrival counters (zero between launches)
  float* ws;        // [R][nkv][stride]: maxima [GQ][16], page sums [GQ][mpgp], super-block sums [GQ][nsb_max][HD]
  size_t stride;
  int32_t* err;     // set to 1 when a wait gave up
  int mpgp;
  float cap;        // > 0: score = ixt_tanh(score / cap) * cap before the max (attn_logits_soft_cap, attentions.py:1257-1259), as k_attn_pv
  int window;       // > 0: keys j <= pos - window are invisible (LOCAL_SLIDING, attentions.py:625-631): e_j = p_j = +0, every sum unchanged
};

// The exchanged words are written and read with agent-scope accesses (sc1: they bypass the XCD-private L2 state, so no cache write-back or
// invalidate is needed -- measured: with __threadfence() / acquire fences around the counter a one-utterance step took 1.64 instead of 1.05 ms).
__device__ __forceinline__ void ixt_xst(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ixt_xld(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Eight agent-scope loads IN FLIGHT together (v[i] = p[i * stride], i < n <= 8; the rest 0).  The compiler waits for every relaxed atomic
// load before it issues the next one (a 33-term sum took 33 memory round trips: Gemma-2 decode 4.5 -> 5.1 ms per step), so these are written
// out: the loads, then ONE wait that owns all eight registers -- no use of a value can be scheduled in front of it.  n is wave-uniform.
__device__ __forceinline__ void ixt_xld8(const float* p, size_t stride, int n, float (&v)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    v[i] = 0.0f;
    if (i < n) asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v[i]) : "v"(p + (size_t)i * stride) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : : "memory");
}
__device__ __forceinline__ void ixt_xld16(const float* p, size_t stride, int n, float (&v)[16]) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    v[i] = 0.0f;
    if (i < n) asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v[i]) : "v"(p + (size_t)i * stride) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
               "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : : "memory");
}
// two such batches (two output elements of a thread) behind ONE wait
__device__ __forceinline__ void ixt_xld16x2(const float* p0, const float* p1, size_t stride, int n, float (&v)[16], float (&u)[16]) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    v[i] = 0.0f; u[i] = 0.0f;
    if (i < n) {
      asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v[i]) : "v"(p0 + (size_t)i * stride) : "memory");
      asm volatile("global_load_dword %0, %1, off sc1" : "=v"(u[i]) : "v"(p1 + (size_t)i * stride) : "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
               "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : : "memory");
  asm volatile("" : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]),
               "+v"(u[8]), "+v"(u[9]), "+v"(u[10]), "+v"(u[11]), "+v"(u[12]), "+v"(u[13]), "+v"(u[14]), "+v"(u[15]) : : "memory");   // (after the wait: volatile statements keep their order)
}
__device__ __forceinline__ void ixt_arrive_and_wait(int32_t* cnt, int target, int32_t* err, int tid) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's exchange stores have completed (at agent scope: they are sc1 stores)
  __syncthreads();
  if (tid == 0) {
    __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int it = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++it > IXT_SPIN_LIMIT) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
  }
  __syncthreads();
}

template <int HD, int GQ>
__global__ __launch_bounds__(256) void k_attn_split(DecAttnArgs a, SplitWs w) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  static_assert(HD <= 256 && HD % 16 == 0, "one wave covers a page's features, FPL per lane");
  constexpr int FPL = HD >= 128 ? HD / 64 : 2;      // PV: features per lane (2: one dword per token, 4: two)
  constexpr int VW = FPL / 2;
  const AttnGeom& g = a.g;
  const int kvh = blockIdx.x, r = blockIdx.y, s = blockIdx.z, S = gridDim.z;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_out = g.nq * HD;
  if (!a.row_active[r]) {
    if (s == 0) for (int i = tid; i < GQ * HD; i += 256) a.out[xidx(r, kvh * GQ * HD + i, n_out)] = 0;
    return;
  }
  const int tp = (g.Tmax + 3) & ~3;
  float* qs = sm;                                   // [GQ][HD]
  float* ebuf = qs + GQ * HD;                       // [GQ][Tmax'] scores -> exps -> P of the LOCAL keys (index: local page * P + token)
  int32_t* pm = reinterpret_cast<int32_t*>(ebuf + (size_t)GQ * tp);   // [mpg'] local page list
  float* wred = reinterpret_cast<float*>(pm + w.mpgp);                // [16] wave maxima, [16..16+GQ) m, [24..24+GQ) l
  float* pacc = wred + 32;                          // [GQ][4][HD] page chains of the current super-block
  float* lps = pacc + (size_t)GQ * 4 * HD;          // [GQ][mpg'] every block's page sums
  const int pos = a.row_pos[r];
  const int kv_len = pos + 1;
  const int pages = (kv_len + g.P - 1) / g.P;
  const int nsb = (pages + IXT_SB - 1) / IXT_SB;
  const int sb0 = (int)((long)nsb * s / S), sb1 = (int)((long)nsb * (s + 1) / S);
  const int p0 = sb0 * IXT_SB, p1 = min(pages, sb1 * IXT_SB);
  const int npl = max(p1 - p0, 0);                  // local pages
  const int nkl = npl > 0 ? min(kv_len, p1 * g.P) - p0 * g.P : 0;   // local keys
  const int jvis = ((w.window > 0 && kv_len > w.window) ? kv_len - w.window : 0) - p0 * g.P;   // first visible key, as a local index
  const size_t pair = (size_t)r * g.nkv + kvh;
  int32_t* cnt = w.cnt + pair * 4;
  float* gmx = w.ws + pair * w.stride;              // [GQ][16]
  float* gps = gmx + GQ * 16;                        // [GQ][mpgp]
  float* gsb = gps + (size_t)GQ * w.mpgp;           // [GQ][nsb_max][HD]
  {
    const int32_t* prow = a.page_map + (size_t)a.row_slot[r] * g.mpg + p0;
    for (int i = tid; i < npl; i += 256) pm[i] = prow[i];
    const bf16_t* qsrc = a.qbuf + ((size_t)r * g.nq + kvh * GQ) * HD;
    for (int i = tid; i < GQ * HD; i += 256) qs[i] = bf2f(qsrc[i]);
  }
  __syncthreads();

  // ---- scores of the local keys: wave = page of the super-block, lane = key.  A page's key goes through the registers in two halves of d
  //      (half f = 2 i + h of the wave's i-th page); four half buffers in flight up to head_dim 128, two at 256 (with four, the 256 registers
  //      they take leave nothing else in the file: the loads were waited for one by one, 87 us per launch at 4100 keys).
  constexpr int KH = HD / 16;             // 16-byte d-chunks per half
  {
    const int nloc = sb1 - sb0;           // the wave's pages: one per local super-block
    float acc[GQ];
#define IXT_SKLOAD(DST_, F_)                                                                                      \
    do {                                                                                                          \
      const int i_ = (F_) >> 1, h_ = (F_) & 1, p_ = (sb0 + i_) * IXT_SB + wave;                                   \
      if (i_ < nloc && p_ < pages && lane < g.P && p_ * g.P + lane < kv_len) {                                    \
        const bf16_t* kp_ = a.kpages + kpage_off(g, pm[p_ - p0], kvh, h_ * KH, lane);                             \
        _Pragma("unroll") for (int dc = 0; dc < KH; ++dc) DST_[dc] = *reinterpret_cast<const u32x4*>(kp_ + (size_t)dc * g.P * 8); \
      }                                                                                                           \
    } while (0)
#define IXT_SKHALF(BUF_, F_)                                                                                      \
    do {                                                                                                          \
      const int i_ = (F_) >> 1, h_ = (F_) & 1, p_ = (sb0 + i_) * IXT_SB + wave;                                   \
      if (h_ == 0) { _Pragma("unroll") for (int q = 0; q < GQ; ++q) acc[q] = 0.0f; }                              \
      if (i_ < nloc && p_ < pages && lane < g.P && p_ * g.P + lane < kv_len) {                                    \
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
        if (h_ == 1) {                                                                                            \
          if (w.cap > 0.0f) { _Pragma("unroll") for (int q = 0; q < GQ; ++q) acc[q] = ixt_tanh(acc[q] / w.cap) * w.cap; } \
          _Pragma("unroll") for (int q = 0; q < GQ; ++q) ebuf[(size_t)q * tp + (p_ - p0) * g.P + lane] = acc[q];  \
        }                                                                                                         \
      }                                                                                                           \
    } while (0)
    if constexpr (HD <= 128) {
      u32x4 k0[KH], k1[KH], k2[KH], k3[KH];
      IXT_SKLOAD(k0, 0); IXT_SKLOAD(k1, 1); IXT_SKLOAD(k2, 2); IXT_SKLOAD(k3, 3);
      for (int f = 0; f < 2 * nloc; f += 4) {
        IXT_SKHALF(k0, f);
        IXT_SKHALF(k1, f + 1);
        IXT_SKLOAD(k0, f + 4); IXT_SKLOAD(k1, f + 5);
        if (f + 2 < 2 * nloc) {
          IXT_SKHALF(k2, f + 2);
          IXT_SKHALF(k3, f + 3);
          IXT_SKLOAD(k2, f + 6); IXT_SKLOAD(k3, f + 7);
        }
      }
    } else {
      u32x4 k0[KH], k1[KH];
      IXT_SKLOAD(k0, 0); IXT_SKLOAD(k1, 1);
      for (int f = 0; f < 2 * nloc; f += 2) {
        IXT_SKHALF(k0, f);
        IXT_SKLOAD(k0, f + 2);
        IXT_SKHALF(k1, f + 1);
        IXT_SKLOAD(k1, f + 3);
      }
    }
#undef IXT_SKLOAD
#undef IXT_SKHALF
  }
  // the first super-block's values: in flight across both exchanges.  lane = feature pair, one 4-byte load per token
  constexpr int VT = 32;                             // tokens of a page in flight at once
  const bool dl = lane < HD / FPL;
  uint32_t va[VT][VW], vb[VT][VW];
#define IXT_SVLOAD(DST_, SB_, T0_)                                                                                \
  do {                                                                                                            \
    const int p_ = (SB_) * IXT_SB + wave;                                                                         \
    if ((SB_) < sb1 && p_ < pages && dl) {                                                                        \
      const int nt_ = min(g.P, kv_len - p_ * g.P) - (T0_);                                                        \
      const bf16_t* vp_ = a.vpages + vpage_off(g, pm[p_ - p0], kvh, (T0_)) + FPL * lane;                          \
      _Pragma("unroll") for (int u = 0; u < VT; ++u) {                                                            \
        if (u < nt_) {                                                                                            \
          if constexpr (VW == 1) DST_[u][0] = *reinterpret_cast<const uint32_t*>(vp_ + (size_t)u * HD);           \
          else { const uint2 t2_ = *reinterpret_cast<const uint2*>(vp_ + (size_t)u * HD); DST_[u][0] = t2_.x; DST_[u][1] = t2_.y; } \
        }                                                                                                         \
      }                                                                                                           \
    }                                                                                                             \
  } while (0)
  IXT_SVLOAD(va, sb0, 0);
  IXT_SVLOAD(vb, sb0 + 1, 0);
  __syncthreads();
  // ---- exchange 1: maxima
  for (int q = 0; q < GQ; ++q) {
    float m = -INFINITY;
    const float* e = ebuf + (size_t)q * tp;
    for (int j = tid; j < nkl; j += 256) { if (j >= jvis) m = fmaxf(m, e[j]); }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if (lane == 0) wred[q * 4 + wave] = m;
  }
  __syncthreads();
  if (tid < GQ) ixt_xst(gmx + tid * 16 + s, fmaxf(fmaxf(wred[tid * 4], wred[tid * 4 + 1]), fmaxf(wred[tid * 4 + 2], wred[tid * 4 + 3])));
  ixt_arrive_and_wait(cnt, S, w.err, tid);
  if (tid < GQ) {
    float mv[16];
    ixt_xld16(gmx + tid * 16, 1, S, mv);
    float m = mv[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) if (i < S) m = fmaxf(m, mv[i]);
    wred[16 + tid] = m;
  }
  __syncthreads();
  // ---- exps, page sums; exchange 2
  for (int q = 0; q < GQ; ++q) {
    float* e = ebuf + (size_t)q * tp;
    const float m = wred[16 + q];
    for (int j = tid; j < nkl; j += 256) e[j] = j >= jvis ? ixt_exp(e[j] - m) : 0.0f;
  }
  __syncthreads();
  for (int i = tid; i < GQ * npl; i += 256) {
    const int q = i / npl, pl = i % npl;
    const float* ep = ebuf + (size_t)q * tp + (size_t)pl * g.P;
    const int n = min(g.P, kv_len - (p0 + pl) * g.P);
    float sacc = 0.0f;
    for (int t = 0; t < n; ++t) sacc = sacc + ep[t];
    ixt_xst(gps + (size_t)q * w.mpgp + p0 + pl, sacc);
  }
  ixt_arrive_and_wait(cnt + 1, S, w.err, tid);
  for (int i = tid; i < GQ * pages; i += 256) lps[(i / pages) * w.mpgp + i % pages] = ixt_xld(gps + (size_t)(i / pages) * w.mpgp + i % pages);   // one load per thread, all in flight
  __syncthreads();
  if (tid < GQ) {
    float l = 0.0f;
    for (int p = 0; p < pages; ++p) l = l + lps[tid * w.mpgp + p];
    wred[24 + tid] = l;
  }
  __syncthreads();
  for (int q = 0; q < GQ; ++q) {
    float* e = ebuf + (size_t)q * tp;
    const float l = wred[24 + q];
    for (int j = tid; j < nkl; j += 256) e[j] = rbf(e[j] / l);
  }
  __syncthreads();
  // ---- PV: wave = page, lane = feature pair; the super-block's four page chains added in page order
#define IXT_SVDOT(BUF_, SB_, T0_)                                                                                 \
  do {                                                                                                            \
    const int p_ = (SB_) * IXT_SB + wave;                                                                         \
    if (p_ < pages && dl) {                                                                                       \
      const int nt_ = min(g.P, kv_len - p_ * g.P) - (T0_);                                                        \
      _Pragma("unroll") for (int u = 0; u < VT; ++u) {                                                            \
        if (u < nt_) {                                                                                            \
          _Pragma("unroll") for (int q = 0; q < GQ; ++q) {                                                        \
            const float pv_ = ebuf[(size_t)q * tp + (size_t)(p_ - p0) * g.P + (T0_) + u];                         \
            _Pragma("unroll") for (int x = 0; x < VW; ++x) {                                                      \
              acc[q][2 * x] = __builtin_fmaf(pv_, bflo(BUF_[u][x]), acc[q][2 * x]);                               \
              acc[q][2 * x + 1] = __builtin_fmaf(pv_, bfhi(BUF_[u][x]), acc[q][2 * x + 1]);                       \
            }                                                                                                     \
          }                                                                                                       \
        }                                                                                                         \
      }                                                                                                           \
    }                                                                                                             \
  } while (0)
  for (int sb = sb0; sb < sb1; ++sb) {       // pages of at most VT tokens (the host's condition); va / vb alternate, two super-blocks ahead
    float acc[GQ][FPL];
#pragma unroll
    for (int q = 0; q < GQ; ++q)
#pragma unroll
      for (int x = 0; x < FPL; ++x) acc[q][x] = 0.0f;
    if (((sb - sb0) & 1) == 0) {
      IXT_SVDOT(va, sb, 0);
      IXT_SVLOAD(va, sb + 2, 0);
    } else {
      IXT_SVDOT(vb, sb, 0);
      IXT_SVLOAD(vb, sb + 2, 0);
    }
    if (dl) {
#pragma unroll
      for (int q = 0; q < GQ; ++q)
#pragma unroll
        for (int x = 0; x < FPL; x += 2) *reinterpret_cast<float2*>(pacc + ((size_t)q * 4 + wave) * HD + FPL * lane + x) = make_float2(acc[q][x], acc[q][x + 1]);
    }
    __syncthreads();
    for (int i = tid; i < GQ * HD; i += 256) {
      const int q = i / HD, d = i % HD;
      float ss = 0.0f;
#pragma unroll
      for (int pg = 0; pg < IXT_SB; ++pg) ss = ss + pacc[((size_t)q * 4 + pg) * HD + d];
      ixt_xst(gsb + ((size_t)q * g.nsb_max + sb) * HD + d, ss);
    }
    __syncthreads();
  }
#undef IXT_SVLOAD
#undef IXT_SVDOT
  // ---- exchange 3: the last block to arrive adds the super-block sums in order
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) wred[0] = (float)__hip_atomic_fetch_add(cnt + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if ((int)wred[0] != S - 1) return;
  if constexpr (GQ * HD % 512 == 0) {       // two output elements per thread and round: their loads fly together
    for (int i = tid; i < GQ * HD; i += 512) {
      const int q0 = i / HD, d0 = i % HD, q1 = (i + 256) / HD, d1 = (i + 256) % HD;
      float t0 = 0.0f, t1 = 0.0f;
      for (int sb = 0; sb < nsb; sb += 16) {
        float sv[16], su[16];
        ixt_xld16x2(gsb + ((size_t)q0 * g.nsb_max + sb) * HD + d0, gsb + ((size_t)q1 * g.nsb_max + sb) * HD + d1, HD, min(16, nsb - sb), sv, su);
#pragma unroll
        for (int k = 0; k < 16; ++k) if (sb + k < nsb) { t0 = t0 + sv[k]; t1 = t1 + su[k]; }
      }
      a.out[xidx(r, (kvh * GQ + q0) * HD + d0, n_out)] = f2bf(t0);
      a.out[xidx(r, (kvh * GQ + q1) * HD + d1, n_out)] = f2bf(t1);
    }
  } else {
    for (int i = tid; i < GQ * HD; i += 256) {
      const int q = i / HD, d = i % HD;
      float t = 0.0f;
      for (int sb = 0; sb < nsb; sb += 16) {
        float sv[16];
        ixt_xld16(gsb + ((size_t)q * g.nsb_max + sb) * HD + d, HD, min(16, nsb - sb), sv);
#pragma unroll
        for (int k = 0; k < 16; ++k) if (sb + k < nsb) t = t + sv[k];
      }
      a.out[xidx(r, (kvh * GQ + q) * HD + d, n_out)] = f2bf(t);
    }
  }
  if (tid < 3) __hip_atomic_store(cnt + tid, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
