// ixt_gemm.h -- the canonical DOT for many rows (>= 128) as an LDS-ring GEMM with fused epilogues (included by ixt_kernels.h).
//
// Same arithmetic as k_dotb / k_gemm (chain c: 16 v_mfma_f32_16x16x32_bf16 over k in [512c, 512c+512) into one accumulator;
// total t = t + chain(c), c ascending), different data movement:
//   * both operands are stored in HBM in fragment order (1 KiB = one wave-instruction), so a K-chunk of a block tile is a list of
//     fragments that LDS-DMA (`global_load_lds_dwordx4`: 64 lanes x 16 B, no VGPRs) drops into an LDS ring slot exactly as the MFMAs
//     read them (lane-linear, conflict-free ds_read_b128);
//   * NS ring slots, NS - 1 chunks in flight per block behind a counted `s_waitcnt vmcnt(N)` and ONE raw s_barrier per chunk (the
//     memory latency of a chunk is hidden behind the two or more chunks ahead of it; k_gemm had one chunk in flight and waited for
//     every one of them);
//   * a block owns whole (tile, all chains) work items: the chain totals stay in registers and the epilogue consumes them on chip --
//     no f32 slice leaves the chip except in the op-level / split forms.
// Block = WR x WC waves; wave tile = RB 16-row blocks x CT 16-column tiles; block tile = 16 WR RB rows x 16 WC CT columns.
// Measured with in-kernel stamps (tools/ring_probe.hip): an LDS-DMA issue holds its wave for about 75 cycles and MFMA issue is in order,
// so one wave per SIMD runs load issue and MFMAs back to back (128 x 128 x 64-k chunk: 1 550 cycles); with TWO waves per SIMD (8-wave
// blocks) one wave's MFMAs run under the other's load issue.
//
// Epilogues (EPI):
//   GEPI_F32      f32 [slice][R][N] (op-level entry point, split-K form)
//   GEPI_UNEMBED  greedy sampling (decoders.py:566-587, inference_utils.py:74-75): bf16-rounded logits on request, one
//                 (max, first index) pair per row and wave column for k_argmax_final
//   GEPI_RESID    h_out = bf16(h_in + bf16(t))                                   (llama2.py:127, :157)
//   GEPI_SWIGLU   act = bf16(bf16(silu(bf16(t_0))) * bf16(t_1))                  (linears.py:449-464); the fused wi kernel stores
//                 wi_0 tile j and wi_1 tile j as column tiles 2j, 2j+1, so both gates of a feature are in one wave
//   GEPI_QKV      bf16(t), RoPE (embeddings.py:277-315), q -> qbuf, k / v -> the row's page (paged_attention.py:446-471).  Q and K
//                 head columns are stored pair-interleaved (qk_col): a 16-column tile holds features 8j..8j+7 and half+8j..half+8j+7,
//                 so the two RoPE partners of a feature sit in lanes l and l ^ 32 of one accumulator register.
#pragma once

enum { GEPI_F32 = 0, GEPI_UNEMBED = 1, GEPI_RESID = 2, GEPI_SWIGLU = 3, GEPI_QKV = 4 };

struct GemmEpi {
  float* out;                                           // F32: [slices][R][N]; UNEMBED: rounded logits [R][N] or nullptr
  float* amax_val; int32_t* amax_idx; ColRanges cr;     // UNEMBED: [R][WC ncolg]
  const bf16_t* h_in; bf16_t* h_out;                    // RESID: [R][N] row-major
  bf16_t* act;                                          // SWIGLU: [R][N/2] in xidx layout
  const int32_t* row_slot; const int32_t* row_pos; const int32_t* row_active; const int32_t* page_map;   // QKV
  const bf16_t* rope_sin; const bf16_t* rope_cos;
  bf16_t* qbuf; bf16_t* kpages; bf16_t* vpages;
  AttnGeom g;
  const float* xscale; const float* wscale;             // F8: activation scales [rows][chains], column scales [N] (ixt_fp8.h)
  size_t sstride;                                       // F8 + routed experts: floats between two experts' column scales
  const int32_t* rb_exp; size_t wstride;                // routed experts: expert of every 16-row block (-1: none), 16-byte units between experts' kernels
};

template <int N_>
__device__ __forceinline__ void ring_wait() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N_) : "memory");
}

//   grid = (8 ceil(ncolg / 8) zb, SPLIT ? chains : 1), block 64 WR WC, dynamic LDS = NS (WC CT + WR RB) KS KiB.
//   Row groups of one column group are block ids b, b + 8, ..: same XCD, the weights meet in its L2.
//   F8 (ixt_fp8.h): both operands are e4m3 fragments of 64 k (still 1 KiB: the ring moves them exactly as bf16 fragments of 32 k); a fragment
//   feeds two v_mfma_f32_16x16x32_fp8_fp8; at the end of every 512-k chain the accumulator is scaled by (a_rc s_n) before it joins the total.
template <int WR, int WC, int RB, int CT, int KS, int NS, int EPI, int SPLIT, bool F8 = false>
__global__ __launch_bounds__(WR * WC * 64, 2) void k_gemm_ring(const u32x4* __restrict__ xt, const u32x4* __restrict__ wt, int K, int N, int R,
                                                               int ncolg, int zb, int wring, GemmEpi e) {
  constexpr int NW = WR * WC;          // waves
  constexpr int TC = WC * CT;          // column tiles of a block
  constexpr int TR = WR * RB;          // row blocks of a block
  constexpr int NWF = TC * KS;         // weight fragments of a chunk
  constexpr int NXF = TR * KS;         // activation fragments
  constexpr int NF = NWF + NXF;
  constexpr int NJ = NF / NW;          // fragments a wave fetches per chunk
  constexpr int D = NS - 1;            // chunks in flight
  static_assert(NF % NW == 0 && 16 % KS == 0 && (D - 1) * NJ < 64, "ring geometry");
  static_assert(EPI != GEPI_SWIGLU || CT % 2 == 0, "gate pairs need two tiles per wave");
  extern __shared__ __attribute__((aligned(16))) u32x4 st[];   // [NS][NF fragments][64 lanes]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int bx = blockIdx.x;
  const int colg = (bx / (8 * zb)) * 8 + (bx & 7), rowg = (bx >> 3) % zb;
  if (colg >= ncolg) return;                                    // whole block: no barrier is skipped
  if (e.rb_exp != nullptr) {                                    // routed experts (k_moe_plan, align 128): this row group's expert, or nothing to do
    const int ex = (rowg * (WR * RB) < ((R + 15) >> 4)) ? e.rb_exp[rowg * (WR * RB)] : -1;
    if (ex < 0) return;
    wt += (size_t)ex * e.wstride;
    if constexpr (F8) e.wscale += (size_t)ex * e.sstride;       // the expert's column scales
  }
  const int nsteps = F8 ? K >> 6 : K >> 5, ntile = N >> 4, nrbk = (R + 15) >> 4;   // fragments along k
  const int nchunks = nsteps / KS;
  const int wr = wave / WC, wc = wave % WC;
  const int r_in = lane & 15, cq = lane >> 4;
  const u32x4* src[NJ];
  // weights: fragment order [tile][k-step] (wring == 0: the layout of the small-batch kernels) or the ring layout (r8frag)
  const size_t wadv = wring ? (size_t)KS * 8 * 64 : (size_t)KS * 64;   // fragments (x 64 lanes) between two chunks of one tile
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int f = wave + NW * j;
    if (f < NWF) {
      const int tile = wring ? colg * TC + f / KS : min(colg * TC + f / KS, ntile - 1);    // past the end: a valid (wring: padding) tile, results dropped
      src[j] = wt + (wring ? r8frag(tile, f % KS, nsteps) : (size_t)tile * nsteps + (f % KS)) * 64 + lane;
    } else {
      const int rb = min(rowg * TR + (f - NWF) / KS, nrbk - 1);
      src[j] = xt + ((size_t)rb * nsteps + ((f - NWF) % KS)) * 64 + lane;
    }
  }
  constexpr int CPC = (F8 ? 8 : 16) / KS;                       // chunks per 512-k chain
  const int chb = SPLIT ? (int)blockIdx.y * CPC : 0;
  const int che = SPLIT ? min(nchunks, chb + CPC) : nchunks;
#define IXT_RISSUE(CH_, SLOT_)                                                                                         \
  do {                                                                                                                 \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                                                     \
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + (size_t)(CH_) * (wave + NW * j < NWF ? wadv : (size_t)KS * 64)), \
                                       (__attribute__((address_space(3))) void*)(st + ((SLOT_) * NF + wave + NW * j) * 64), 16, 0, 0); \
  } while (0)
#pragma unroll
  for (int c = 0; c < D; ++c)
    if (chb + c < che) IXT_RISSUE(chb + c, c);
  f32x4 acc[CT][RB], tot[CT][RB];
#pragma unroll
  for (int i = 0; i < CT; ++i)
#pragma unroll
    for (int b = 0; b < RB; ++b) { acc[i][b] = f32x4(0.0f); tot[i][b] = f32x4(0.0f); }
  int slot = 0;                                                 // ring slot of chunk ch
#ifdef IXT_RING_STAMPS   // diagnostic build (tools/ring_probe.hip): where an iteration spends its cycles; never defined in the product
  unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0, st4 = 0, sum_wait = 0, sum_bar = 0, sum_issue = 0, sum_comp = 0;
#define IXT_STAMP(T_) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(T_) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define IXT_STAMP(T_) do { } while (0)
#endif
  for (int ch = chb; ch < che; ++ch) {
    IXT_STAMP(st0);
    // chunk ch has landed once at most min(D - 1, chunks after it) x NJ of this wave's younger loads are outstanding; the LDS reads
    // of the previous chunk are retired (lgkmcnt(0)) before the barrier lets any wave refill their slot
    const int rem = che - 1 - ch;
    if (rem >= D - 1) ring_wait<(D - 1) * NJ>();
    else if (D > 2 && rem == 2) ring_wait<(D > 2 ? 2 : 0) * NJ>();
    else if (D > 1 && rem == 1) ring_wait<(D > 1 ? 1 : 0) * NJ>();
    else if (D > 3 && rem == 3) ring_wait<(D > 3 ? 3 : 0) * NJ>();
    else ring_wait<0>();
    IXT_STAMP(st1);
    __builtin_amdgcn_s_barrier();
    IXT_STAMP(st2);
    // All LDS reads of the chunk first (their latency then runs under the issue of the refill loads), then the MFMAs in groups of RB
    // with the NJ refill loads of chunk ch + D dealt between the groups: an LDS-DMA issue costs the wave about 50 cycles, a group of
    // 4 MFMAs keeps the matrix pipe busy for 64, so the refill rides under the MFMAs instead of in front of them.
    const u32x4* sb = st + (size_t)slot * NF * 64;
    bf16x8 wf[KS][CT], xf[KS][RB];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int i = 0; i < CT; ++i) wf[ks][i] = __builtin_bit_cast(bf16x8, sb[((wc * CT + i) * KS + ks) * 64 + lane]);
#pragma unroll
      for (int b = 0; b < RB; ++b) xf[ks][b] = __builtin_bit_cast(bf16x8, sb[(NWF + (wr * RB + b) * KS + ks) * 64 + lane]);
    }
    IXT_STAMP(st3);
    const bool refill = ch + D < che;
    const int fill = slot == 0 ? NS - 1 : slot - 1;             // the slot of chunk ch - 1 = that of chunk ch + D
    constexpr int NG = KS * CT;                                 // MFMA groups of an iteration
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int i = 0; i < CT; ++i) {
#pragma unroll
        for (int b = 0; b < RB; ++b) {
          if constexpr (F8) {
            typedef long l2 __attribute__((ext_vector_type(2)));
            const l2 wl = __builtin_bit_cast(l2, wf[ks][i]), xl = __builtin_bit_cast(l2, xf[ks][b]);
            acc[i][b] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wl[0], xl[0], acc[i][b], 0, 0, 0);
            acc[i][b] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wl[1], xl[1], acc[i][b], 0, 0, 0);
          } else {
            acc[i][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][i], xf[ks][b], acc[i][b], 0, 0, 0);
          }
        }
        const int gi = ks * CT + i;
        if (refill) {
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            if (j * NG / NJ == gi)                              // load j rides behind group j NG / NJ
              __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + (size_t)(ch + D) * (wave + NW * j < NWF ? wadv : (size_t)KS * 64)),
                                               (__attribute__((address_space(3))) void*)(st + (fill * NF + wave + NW * j) * 64), 16, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (((ch + 1) % CPC) == 0 || ch + 1 == che) {               // a 512-k chain is complete
      if constexpr (F8) {                                       // scale the chain by (activation block scale) x (column scale)
        const int chain = ch / CPC, nchains = (nchunks + CPC - 1) / CPC;
#pragma unroll
        for (int b = 0; b < RB; ++b) {
          const int row = min((rowg * TR + wr * RB + b) * 16 + r_in, R - 1);
          const float a = e.xscale[(size_t)row * nchains + chain];
#pragma unroll
          for (int i = 0; i < CT; ++i) {
            const int tile = min(colg * TC + wc * CT + i, ntile - 1);
            const float4 sn = *reinterpret_cast<const float4*>(e.wscale + tile * 16 + cq * 4);
            acc[i][b][0] *= a * sn.x; acc[i][b][1] *= a * sn.y; acc[i][b][2] *= a * sn.z; acc[i][b][3] *= a * sn.w;
          }
        }
      }
#pragma unroll
      for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int b = 0; b < RB; ++b) { tot[i][b] = tot[i][b] + acc[i][b]; acc[i][b] = f32x4(0.0f); }
    }
    slot = slot + 1 == NS ? 0 : slot + 1;
#ifdef IXT_RING_STAMPS
    IXT_STAMP(st4);
    sum_wait += st1 - st0; sum_bar += st2 - st1; sum_issue += st3 - st2; sum_comp += st4 - st3;
#endif
  }
#ifdef IXT_RING_STAMPS
  if (lane == 0 && e.amax_val != nullptr) {   // the probe passes a buffer of its own here: [block][wave][4] cycle sums
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(e.amax_val) + ((size_t)bx * NW + wave) * 4;
    dbg[0] = sum_wait; dbg[1] = sum_bar; dbg[2] = sum_issue; dbg[3] = sum_comp;
  }
#endif
#undef IXT_STAMP
#undef IXT_RISSUE

  // ---------------------------------------------------------------- epilogues
  if constexpr (EPI == GEPI_F32) {
    float* out = e.out + (SPLIT ? (size_t)blockIdx.y * R * N : 0);
#pragma unroll
    for (int i = 0; i < CT; ++i) {
      const int tile = colg * TC + wc * CT + i;
#pragma unroll
      for (int b = 0; b < RB; ++b) {
        const int row = (rowg * TR + wr * RB + b) * 16 + r_in;
        if (tile < ntile && row < R)
          *reinterpret_cast<float4*>(out + (size_t)row * N + tile * 16 + cq * 4) = make_float4(tot[i][b][0], tot[i][b][1], tot[i][b][2], tot[i][b][3]);
      }
    }
  } else if constexpr (EPI == GEPI_UNEMBED) {
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      const int row = (rowg * TR + wr * RB + b) * 16 + r_in;
      float bv = -INFINITY;
      int bi = 0x7fffffff;
#pragma unroll
      for (int i = 0; i < CT; ++i) {
        const int tile = colg * TC + wc * CT + i;
        if (tile < ntile) {
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = rbf(tot[i][b][j]);
          if (e.out != nullptr && row < R) *reinterpret_cast<float4*>(e.out + (size_t)row * N + tile * 16 + cq * 4) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (v[j] > bv && col_allowed(e.cr, tile * 16 + cq * 4 + j)) { bv = v[j]; bi = tile * 16 + cq * 4 + j; }   // ascending columns
        }
      }
#pragma unroll
      for (int off = 16; off < 64; off <<= 1) {
        const float ov = __shfl_xor(bv, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
      }
      if (cq == 0 && row < R) {
        e.amax_val[(size_t)row * (WC * ncolg) + colg * WC + wc] = bv;
        e.amax_idx[(size_t)row * (WC * ncolg) + colg * WC + wc] = bi;
      }
    }
  } else if constexpr (EPI == GEPI_RESID) {
#pragma unroll
    for (int i = 0; i < CT; ++i) {
      const int tile = colg * TC + wc * CT + i;
#pragma unroll
      for (int b = 0; b < RB; ++b) {
        const int row = (rowg * TR + wr * RB + b) * 16 + r_in;
        if (tile < ntile && row < R) {
          const size_t o = (size_t)row * N + tile * 16 + cq * 4;
          const uint2 hv = *reinterpret_cast<const uint2*>(e.h_in + o);
          uint2 ov;
          ov.x = pack2(f2bf(bflo(hv.x) + rbf(tot[i][b][0])), f2bf(bfhi(hv.x) + rbf(tot[i][b][1])));
          ov.y = pack2(f2bf(bflo(hv.y) + rbf(tot[i][b][2])), f2bf(bfhi(hv.y) + rbf(tot[i][b][3])));
          *reinterpret_cast<uint2*>(e.h_out + o) = ov;
        }
      }
    }
  } else if constexpr (EPI == GEPI_SWIGLU) {
    const int F = N >> 1;
#pragma unroll
    for (int i = 0; i < CT; i += 2) {
      const int tile = colg * TC + wc * CT + i;             // even: wi_0 tile tile/2; tile + 1: wi_1 tile tile/2
#pragma unroll
      for (int b = 0; b < RB; ++b) {
        const int row = (rowg * TR + wr * RB + b) * 16 + r_in;
        if (tile + 1 < ntile && row < R) {
          const int f0 = (tile >> 1) * 16 + cq * 4;
          bf16_t o[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = ixt_swiglu(rbf(tot[i][b][j]), rbf(tot[i + 1][b][j]));
          uint2 ov;
          ov.x = pack2(o[0], o[1]); ov.y = pack2(o[2], o[3]);
          *reinterpret_cast<uint2*>(e.act + xidx(row, f0, F)) = ov;
        }
      }
    }
  } else if constexpr (EPI == GEPI_QKV) {
    const AttnGeom& g = e.g;
    const int half = g.hd >> 1, tph = g.hd >> 4;                // tiles per head
    const int QN = g.nq * g.hd, KN = g.nkv * g.hd;
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      const int row = (rowg * TR + wr * RB + b) * 16 + r_in;
      const bool row_ok = row < R;
      const int pos = row_ok ? e.row_pos[row] : 0;
      const bool wr_kv = row_ok && e.row_active[row] != 0;
      int page = 0;
      if (wr_kv) page = e.page_map[(size_t)e.row_slot[row] * g.mpg + pos / g.P];
      const int tok = pos % g.P;
#pragma unroll
      for (int i = 0; i < CT; ++i) {
        const int tile = colg * TC + wc * CT + i;
        const int col0 = tile * 16;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = rbf(tot[i][b][j]);
        const bool is_v = col0 >= QN + KN;
        const int hh = col0 / g.hd, th = tile % tph;            // head (q heads, then k heads, then v heads), tile inside the head
        if (!is_v) {
          // pair-interleaved tile: lanes cq 0,1 hold x1 = features 8 th + 4 (cq & 1) + j, lanes cq 2,3 the partners x2 = half + the same
          const int fi = 8 * th + 4 * (cq & 1);
          if (g.rope32) {   // rope_type "llama3.1": f32 tables, one rounding (ixt_rope_pair)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float other = __shfl_xor(v[j], 32, 64);
              float o1, o2;
              ixt_rope_pair(1, e.rope_sin, e.rope_cos, (size_t)pos * half + fi + j, cq < 2 ? v[j] : other, cq < 2 ? other : v[j], o1, o2);
              v[j] = cq < 2 ? o1 : o2;
            }
          } else {
          float sn[4], cs[4];
          {
            const uint2 sv = *reinterpret_cast<const uint2*>(e.rope_sin + (size_t)pos * half + fi);
            const uint2 cv = *reinterpret_cast<const uint2*>(e.rope_cos + (size_t)pos * half + fi);
            sn[0] = bflo(sv.x); sn[1] = bfhi(sv.x); sn[2] = bflo(sv.y); sn[3] = bfhi(sv.y);
            cs[0] = bflo(cv.x); cs[1] = bfhi(cv.x); cs[2] = bflo(cv.y); cs[3] = bfhi(cv.y);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float other = __shfl_xor(v[j], 32, 64);
            // first half: bf16(bf16(x1 c) - bf16(x2 s)); second half: bf16(bf16(x2 c) + bf16(x1 s))
            v[j] = cq < 2 ? rbf(rbf(v[j] * cs[j]) - rbf(other * sn[j])) : rbf(rbf(v[j] * cs[j]) + rbf(other * sn[j]));
          }
          }
        }
        if (tile >= ntile || !row_ok) continue;
        uint2 ov;
        ov.x = pack2(f2bf(v[0]), f2bf(v[1])); ov.y = pack2(f2bf(v[2]), f2bf(v[3]));
        if (g.rope32 && !is_v) {   // llama3.1: the lane's four de-interleaved features are every second feature of the model's order (ixt_qk_feat)
          const int feat = (cq >= 2 ? half : 0) + 8 * th + 4 * (cq & 1);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int fn_ = ixt_qk_feat(1, feat + j, half);
            if (hh < g.nq) e.qbuf[((size_t)row * g.nq + hh) * g.hd + fn_] = f2bf(v[j]);
            else if (wr_kv) e.kpages[kpage_off(g, page, hh - g.nq, fn_ >> 3, tok) + (fn_ & 7)] = f2bf(v[j]);
          }
        } else if (hh < g.nq) {
          const int feat = (cq >= 2 ? half : 0) + 8 * th + 4 * (cq & 1);
          *reinterpret_cast<uint2*>(e.qbuf + ((size_t)row * g.nq + hh) * g.hd + feat) = ov;
        } else if (!is_v) {
          if (wr_kv) {
            const int feat = (cq >= 2 ? half : 0) + 8 * th + 4 * (cq & 1);
            *reinterpret_cast<uint2*>(e.kpages + kpage_off(g, page, hh - g.nq, feat >> 3, tok) + (feat & 7)) = ov;
          }
        } else if (wr_kv) {
          const int feat = th * 16 + cq * 4;
          *reinterpret_cast<uint2*>(e.vpages + vpage_off(g, page, hh - g.nq - g.nkv, tok) + feat) = ov;
        }
      }
    }
  }
}
