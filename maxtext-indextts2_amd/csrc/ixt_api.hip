// ixt_api.hip -- C ABI (include/ixt.h) over the gfx950 kernels in ixt_kernels.h.
// Host code here is plumbing only: argument checks, HBM ownership, launch geometry, hipGraph capture.
#include "../../include/ixt.h"
#include "ixt_kernels.h"
#include "ixt_gemma_kernels.h"
#include "ixt_moe_kernels.h"
#include "ixt_small.h"
#include "ixt_fp8.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

static thread_local char g_err[512] = "";
static int set_err(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
#define HIPCHK(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess) return set_err(IXT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)
#define LAUNCHCHK() HIPCHK(hipGetLastError())

extern "C" const char* ixt_last_error(void) { return g_err; }

struct LayerW {
  bf16_t *ln1, *wqkv, *wo, *ln2, *wi, *wout;
  bf16_t *pan = nullptr, *pfn = nullptr;   // Gemma-2 / 3: post-attention / post-MLP norm scales
  bf16_t *qn = nullptr, *kn = nullptr;     // Gemma-3: query / key head norm scales [head_dim]
  bf16_t* gate = nullptr;                  // Mixtral: MoeBlock_0/gate/kernel, row-major [emb][experts]
  std::vector<bf16_t*> ewi, ewout;         // Mixtral: per expert the (wi_0, wi_1) pair and wo in the dot kernels' layout (wi / wout unused)
};

enum { W_LN1 = 0, W_Q, W_K, W_V, W_O, W_LN2, W_WI0, W_WI1, W_WOUT, W_PER_LAYER, W_PAN = W_PER_LAYER, W_PFN, W_QN, W_KN, W_GATE };

struct TimingPair { hipEvent_t a, b; double bytes; int family; };   // family: 0 canonical dot (incl. unembedding), 1 one-launch decode attention

struct GraphEntry {
  hipGraphExec_t exec;
};

struct ixt_ctx {
  ixt_config cfg;
  int E, Q, KV, D, F, V, L, P, B, Rmax, NQKV, QN, KN, Tmax, mpg, nsb_max, GQ;
  int G = 0, S = 0;            // page groups = B decode slots + S staging groups of packed prefill
  int NLmax = 1;               // rows that can be unembedded in one call (decode slots / prompts of a packed prefill)
  // weights
  bf16_t* embedding = nullptr;
  bf16_t* final_norm = nullptr;
  bf16_t* logits_w = nullptr;
  std::vector<LayerW> layers;
  std::vector<uint32_t> loaded;  // per layer bitmask; index L = globals (bit0 emb, bit1 norm, bit2 logits)
  // tables / pools / state
  bf16_t *rope_sin = nullptr, *rope_cos = nullptr;
  bf16_t *rope_sin_local = nullptr, *rope_cos_local = nullptr;   // Gemma-3 local layers (local_rope_max_timescale); nullptr: the tables above
  std::vector<uint16_t> rope_sin_h, rope_cos_h;
  std::vector<float> rope_sin32_h, rope_cos32_h;   // rope_type "llama3.1": f32 tables; rope_sin / rope_cos then point at their device copies
  bool rope32 = false;
  std::vector<bf16_t*> kpages_l, vpages_l;   // one allocation per layer: [page][kv_head][...]
  bf16_t **kptrs_d = nullptr, **vptrs_d = nullptr;   // the same pointers in HBM (k_pages_copy)
  size_t layer_page_elems = 0;
  int32_t* lrows = nullptr;                  // rows to unembed (ixt_prefill_rows)
  PageStateDev ps{};
  // workspaces
  bf16_t *h_a = nullptr, *h_b = nullptr, *xn = nullptr, *qbuf = nullptr, *ao = nullptr, *hact = nullptr;
  float *part = nullptr, *scores = nullptr, *pvpart = nullptr, *amax_val = nullptr;
  int32_t* amax_idx = nullptr;
  size_t amax_cap = 0;
  bf16_t* xt_op = nullptr;  // op-level entry points: the caller's row-major x in the dot kernel's x layout
  size_t xt_cap = 0;
  bf16_t* wt_op = nullptr;  // ixt_op_dot: the caller's row-major kernel in the tiled weight layout (grown on demand)
  size_t wt_cap = 0;
  size_t part_floats = 0;
  int32_t *row_slot = nullptr, *row_pos = nullptr, *row_active = nullptr, *row_tok = nullptr, *new_tok = nullptr;
  AttnGeom geom{};
  // timing
  bool timing = false;
  int attn_fused_min_blocks = 64;    // (Mixtral-8x7B, 8 slots x 8 kv heads: one launch 19.5 ms per step against 20.5 for the split form)
  // tiled GEMM form of the dot: from 128 rows and 150 blocks of 128 x 128 (at C2 / 256 rows: the 2048 x 14336 MLP kernel only; the
  // 2048 x 6144 QKV kernel has 96 such blocks and measured slower than split-K, as did 64-column blocks)
  int gemm_min_rows = 128, gemm_min_blocks = 150;
  int gemm_prefix_halves = 0;   // measured at 256 rows on the MLP up-projection: decode step 5.39 vs 4.84 ms: off
  int gemm_split_min_blocks = 300;   // 128 x 128 x one-chain blocks (at C2 / 256 rows: the QKV and MLP-down kernels; +2.4 % end to end)
  int unembed_gemm = 1;
  // LDS-ring GEMM with fused epilogues (ixt_gemm.h) from this many rows; ring_cfg forces a tile (1: 128 x 128, 2: 128 x 64, 3: 64 x 32)
  int ring_min_rows = 128, ring_cfg = 0, ring_split = 0, ring_wlayout = 0;
  // MLP down-projection on the ring path: 1 = ring GEMM with the residual epilogue; 0 = split-K tiles (one f32 slice per 512-k chain, two blocks
  // per CU) + slice-sum / residual / norm kernel; -1 (default) = by shape: the ring form needs at least 192 whole-K blocks of 128 x 128 or 128 x 64
  int ring_down = -1;
  int unembed_tall = 1;      // 256 x 128 tiles for the unembedding at 129..256 rows (0: 128 x 128)
  int ring_two_per_cu = 1;   // ring GEMMs of >= 1024 rows: 32-k chunks in 4 slots (64 KiB of LDS), two blocks per CU
  int attn_tile = 1;         // prefill rows: tiled attention (k_attn_tile) from attn_tile_min_rows rows on
  int attn_tile_min_rows = 64;
  int attn_wave_pairs = 0;   // measured at 256 rows, 765 keys: 339 us per launch (4.7 TB/s) against 305 us (5.3 TB/s) for the block form
  // few rows (one row block): five launches per layer (ixt_small.h).  small_rows: most rows of that path (0: off); small_tpb: column tiles per
  // block of k_sdot (0: by shape); small_attn_waves: waves per (row, kv head) block of k_attn_small (4: k_attn_decode instead)
  int small_rows = 16, small_tpb = 0, small_attn_waves = 16, small_attn_max_blocks = 512;
  int dot_small = 0;         // ixt_op_dot through k_sdot where it fits (tests)
  // few rows: the keys of a (row, kv head) split over `S` blocks of k_attn_split (S = min(small_attn_split, 256 / (rows * kv heads)); < 2: k_attn_small)
  int small_attn_split = 16;
  int split_capacity = -1;   // blocks of k_attn_split the chip holds at once (occupancy query, once)
  int moe_mid_pairs = 0;     // routed experts: up to this many (row, expert) copies per launch take 64-row whole-K dots instead of the ring GEMM.  Measured SLOWER at Mixtral-8x7B's 256-token prefill (512 copies): 74.7 against 30.2 ms -- 4 waves per block cannot hide both operands' latency at 64 rows: off
  int moe_fused_gate = 1;    // routed experts at few copies: wi as whole-K blocks with the SwiGLU epilogue (0: split-K slices + k_swiglu_part)
  int gemma_tile = 1;        // Gemma blocks: tiled prefill attention (0: the per-row scores / PV / combine kernels)
  int moe_route_blocks = 1;  // router: one wave per 512-k block, all experts at once (0: one wave per expert)
  int32_t* split_cnt = nullptr;   // [16][KV][4] arrival counters (zero between launches) + the error word behind them
  float* split_ws = nullptr;      // [16][KV][split_stride]
  size_t split_stride = 0;
  int small_prefetch = 0;    // few-rows path: prefetch blocks stream the next launch's kernel into the Infinity Cache (ixt_prefetch).  Measured SLOWER (one utterance 1.19 against 1.03 ms per step, eight 1.33 against 1.23: a launch cannot end before its prefetch has, and an Infinity-Cache hit is only 1.4 x an HBM stream): off
  int small_prefetch_blocks = 64;
  int small_norm = 1;        // 1: the RMSNorm in front of the QKV / wi projections runs inside k_sdot; 0: k_rownorm launches (experiments)
  int dot_stream_min_blocks = 1 << 30;   // off: measured slower than split-K at 256 rows (decode 7.84 vs 7.68 ms/step, prefill 3.53 vs 3.31 ms)
  ColRanges sample_cols{};   // greedy sampling restricted to these columns (n == 0: all)
  // Gemma-2 block (ixt_config.decoder_block == 1): L counts (attention, MLP) sub-layers, two per reference layer (even: local sliding)
  bool gemma = false;
  bool moe = false;          // Mixtral (decoder_block == 3): the llama2 wiring with a routed mixture of NE expert MLPs, TK per row
  int NE = 1, TK = 1;
  int NEL = 1, E0 = 0;       // expert parallelism: this context's experts are E0 .. E0 + NEL - 1 (ixt_config.ep_rank / ep_size)
  int32_t* moe_idx = nullptr;   // [Rmax][TK]
  float* moe_w = nullptr;       // [Rmax][TK]
  bf16_t* moe_y = nullptr;      // [Rmax][TK][E]
  // routed evaluation (k_moe_plan / k_moe_gather, grouped k_dotb / k_gemm_ring): copies of a row sit in the rows of their expert
  int moe_grouped = 1;          // 0: every expert over all rows (round 2's form; experiments)
  int moe_rows_cap = 0;         // copy rows incl. alignment padding the workspaces hold
  int32_t* moe_pos = nullptr;   // [Rmax TK] destination row of every copy
  int32_t* moe_rbexp = nullptr; // [moe_rows_cap / 16] expert of every 16-row block, -1: none
  int32_t* moe_cnt = nullptr;   // [2 NE] counts, offsets
  bf16_t* moe_xg = nullptr;     // [moe_rows_cap][E] gathered inputs (xidx layout)
  bf16_t* moe_hact = nullptr;   // [moe_rows_cap][F] SwiGLU outputs (xidx layout)
  float* moe_comb = nullptr;    // [Rmax][E] the mixture as one f32 slice
  size_t moe_wi_stride = 0, moe_wo_stride = 0;   // elements between two experts' kernels
  bool gemma3 = false;       // Gemma-3 (decoder_block == 2): gemma = true as well; one sub-layer per layer, layer l local unless l % 6 == 5
  float pv_cap = 0.0f;       // soft cap / window of the attention launch being issued (k_attn_pv)
  int pv_window = 0;
  float* glogits = nullptr;  // [NLmax][V] logits rows when the caller asks for none (the cap and the sampling need them)
  // return_prompt_logp (ixt_prefill_prompt_logp): the hidden rows the last forward pass left behind, and a logits workspace made on first use
  const bf16_t* h_final = nullptr;
  int h_final_rows = 0;
  float* plogits = nullptr;     // [NLmax][V]
  int32_t* pargmax = nullptr;   // [NLmax]
  // IXT-FP8-v1 (ixt_fp8.h): e4m3 twin + column scales of every projection kernel (made from the bf16 copy on first use), activation workspace
  struct F8W { uint8_t* w8; float* s; int K, N; };
  bool fp8 = false, fp8_ready = false;
  std::map<const bf16_t*, F8W> f8w;
  uint8_t* x8 = nullptr;        // [rows16][Kmax] e4m3 fragments
  float* xs8 = nullptr;         // [rows16][chains]
  std::vector<TimingPair> tpairs;
  size_t tused = 0;
  // graphs keyed by the pointer tuple of ixt_generate
  std::map<std::string, GraphEntry> graphs;
  bool capturing = false;
  std::vector<void*> allocs;
};

template <typename T>
static int dmalloc(ixt_ctx* c, T** p, size_t n) {
  void* q = nullptr;
  HIPCHK(hipMalloc(&q, n * sizeof(T) + 256));
  c->allocs.push_back(q);
  *p = (T*)q;
  return IXT_OK;
}
#define TRY(x) do { int rc_ = (x); if (rc_ != IXT_OK) return rc_; } while (0)

static bool is_pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }

static float host_rbf(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  u &= 0xffff0000u;
  memcpy(&f, &u, 4);
  return f;
}

static void host_rope_tables(const ixt_config& cfg, float max_timescale, std::vector<uint16_t>& sn, std::vector<uint16_t>& cs) {
  // embeddings.py:270-307; every step's rounding is listed in DESIGN.md section 3.
  const int half = cfg.head_dim / 2, T = cfg.max_target_length;
  sn.resize((size_t)T * half);
  cs.resize((size_t)T * half);
  auto f2bf_h = [](float f) {
    uint32_t u; memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
  };
  std::vector<float> ts(half);
  for (int i = 0; i < half; ++i) {
    const float frac = (float)(2 * i) / (float)cfg.head_dim;
    ts[i] = (float)((double)cfg.rope_min_timescale * pow((double)max_timescale / (double)cfg.rope_min_timescale, (double)frac));
  }
  for (int p = 0; p < T; ++p)
    for (int i = 0; i < half; ++i) {
      const float ang = (float)p / ts[i];
      sn[(size_t)p * half + i] = f2bf_h((float)sin((double)ang));
      cs[(size_t)p * half + i] = f2bf_h((float)cos((double)ang));
    }
}

// LLaMARotaryEmbedding (embeddings.py:396-443): timescale_i = min (max / min)^(2 i / D) for the pair (2 i, 2 i + 1); with use_scale the
// frequency 1 / timescale goes through the llama-3.1 rule: wavelengths below 8192 / 4 keep it, above 8192 divide it by 8, between the two the
// smooth blend.  angle = position / timescale in f32; sin / cos evaluated in double and rounded to f32.
static void host_rope_tables_llama31(const ixt_config& cfg, std::vector<float>& sn, std::vector<float>& cs) {
  const int half = cfg.head_dim / 2, T = cfg.max_target_length;
  sn.resize((size_t)T * half);
  cs.resize((size_t)T * half);
  std::vector<float> ts(half);
  for (int i = 0; i < half; ++i) {
    const float frac = (float)(2 * i) / (float)cfg.head_dim;
    float t = (float)((double)cfg.rope_min_timescale * pow((double)cfg.rope_max_timescale / (double)cfg.rope_min_timescale, (double)frac));
    if (cfg.rope_use_scale) {
      const float scale_factor = 8.0f, low = 1.0f, high = 4.0f, old_len = 8192.0f;
      float freq = 1.0f / t;
      const float wavelen = 2.0f * 3.14159265358979323846f / freq;
      if (wavelen < old_len / high) {
      } else if (wavelen > old_len / low) {
        freq = freq / scale_factor;
      } else {
        const float smooth = (old_len / wavelen - low) / (high - low);
        freq = (1.0f - smooth) * freq / scale_factor + smooth * freq;
      }
      t = 1.0f / freq;
    }
    ts[i] = t;
  }
  for (int p = 0; p < T; ++p)
    for (int i = 0; i < half; ++i) {
      const float ang = (float)p / ts[i];
      sn[(size_t)p * half + i] = (float)sin((double)ang);
      cs[(size_t)p * half + i] = (float)cos((double)ang);
    }
}

static int sdot_set_attributes();

extern "C" int ixt_create(const ixt_config* cfg, ixt_ctx** out) {
  if (!cfg || !out) return set_err(IXT_ERR_INVALID, "null argument");
  const ixt_config& c = *cfg;
  const bool gemma3 = c.decoder_block == IXT_BLOCK_GEMMA3;
  const bool gemma = c.decoder_block == IXT_BLOCK_GEMMA2 || gemma3;
  if (c.head_dim != 64 && c.head_dim != 128 && !(c.head_dim == 256 && gemma))
    return set_err(IXT_ERR_UNSUPPORTED, "head_dim %d (64 and 128 built; 256 for decoder_block gemma2 / gemma3)", c.head_dim);
  const bool moe = c.decoder_block == IXT_BLOCK_MIXTRAL;
  if (c.decoder_block != IXT_BLOCK_LLAMA2 && !gemma && !moe) return set_err(IXT_ERR_UNSUPPORTED, "decoder_block %d (llama2, gemma2, gemma3 and mixtral built)", c.decoder_block);
  if (moe && (c.num_experts < 1 || c.num_experts > IXT_MOE_MAX_EXPERTS || c.num_experts_per_tok < 1 || c.num_experts_per_tok > IXT_MOE_MAX_TOPK ||
              c.num_experts_per_tok > c.num_experts))
    return set_err(IXT_ERR_UNSUPPORTED, "mixtral: num_experts in [1, %d], num_experts_per_tok in [1, min(%d, num_experts)]", IXT_MOE_MAX_EXPERTS, IXT_MOE_MAX_TOPK);
  if (!moe && (c.num_experts > 1 || c.num_experts_per_tok > 1)) return set_err(IXT_ERR_UNSUPPORTED, "num_experts > 1 is built for decoder_block mixtral only");
  if (c.ep_size > 1 && (!moe || c.ep_rank < 0 || c.ep_rank >= c.ep_size || c.num_experts % c.ep_size))
    return set_err(IXT_ERR_INVALID, "expert parallelism: decoder_block mixtral, 0 <= ep_rank < ep_size, num_experts %% ep_size == 0");
  if (!gemma3 && (c.local_rope_max_timescale != 0.0f || c.query_pre_attn_scalar != 0.0f))
    return set_err(IXT_ERR_UNSUPPORTED, "local_rope_max_timescale and query_pre_attn_scalar are read for decoder_block gemma3 only");
  if (c.query_pre_attn_scalar < 0.0f) return set_err(IXT_ERR_INVALID, "negative query_pre_attn_scalar");
  if (!gemma && (c.attn_logits_soft_cap != 0.0f || c.final_logits_soft_cap != 0.0f || c.sliding_window_size != 0 || c.use_post_attn_norm || c.use_post_ffw_norm ||
                 c.logits_via_embedding))
    return set_err(IXT_ERR_UNSUPPORTED, "soft caps, sliding window, post norms and the tied head are built for decoder_block gemma2 / gemma3 only");
  if (c.attn_logits_soft_cap < 0.0f || c.final_logits_soft_cap < 0.0f || c.sliding_window_size < 0) return set_err(IXT_ERR_INVALID, "negative soft cap or window");
  if (c.emb_dim % IXT_KALIGN || c.mlp_dim % IXT_KALIGN || (c.num_query_heads * c.head_dim) % IXT_KALIGN)
    return set_err(IXT_ERR_UNSUPPORTED, "emb_dim, mlp_dim and num_query_heads*head_dim must be multiples of %d", IXT_KALIGN);
  if (c.num_kv_heads <= 0 || c.num_query_heads % c.num_kv_heads) return set_err(IXT_ERR_INVALID, "num_query_heads %% num_kv_heads != 0");
  const int gq = c.num_query_heads / c.num_kv_heads;
  if (gq != 1 && gq != 2 && gq != 4) return set_err(IXT_ERR_UNSUPPORTED, "query heads per kv head = %d (1, 2, 4 built)", gq);
  if (!is_pow2(c.pagedattn_tokens_per_page) || c.pagedattn_tokens_per_page < 8 || c.pagedattn_tokens_per_page > 64)
    return set_err(IXT_ERR_UNSUPPORTED, "pagedattn_tokens_per_page must be a power of two in [8, 64]");
  if (c.vocab_size % 16) return set_err(IXT_ERR_UNSUPPORTED, "vocab_size must be a multiple of 16");
  if (c.staging_groups < 0 || c.max_packed_rows < 0) return set_err(IXT_ERR_INVALID, "staging_groups / max_packed_rows must be >= 0");
  if (c.max_slots + c.staging_groups > IXT_MAX_GROUPS || c.pagedattn_max_pages_per_group > IXT_MAX_GROUPS)
    return set_err(IXT_ERR_UNSUPPORTED, "more than %d page groups or pages per group", IXT_MAX_GROUPS);
  // page_manager.py:450-473
  if (c.pagedattn_max_pages_per_group <= 0) return set_err(IXT_ERR_INVALID, "`pagedattn_max_pages_per_group` must be positive.");
  if (c.pagedattn_max_pages_per_group < (c.max_target_length + c.pagedattn_tokens_per_page - 1) / c.pagedattn_tokens_per_page)
    return set_err(IXT_ERR_INVALID, "`pagedattn_max_pages_per_group` is insufficient for `max_target_length`.");
  if (c.pagedattn_num_pages <= 1) return set_err(IXT_ERR_INVALID, "`pagedattn_num_pages` must be greater than 1.");
  if (c.max_slots <= 0) return set_err(IXT_ERR_INVALID, "`pagedattn_max_page_groups` must be positive.");
  if (c.max_prefill_predict_length <= 0 || c.max_target_length <= 0) return set_err(IXT_ERR_INVALID, "lengths must be positive");

  ixt_ctx* x = new ixt_ctx();
  x->cfg = c;
  x->E = c.emb_dim; x->Q = c.num_query_heads; x->KV = c.num_kv_heads; x->D = c.head_dim; x->F = c.mlp_dim;
  x->V = c.vocab_size; x->L = c.num_decoder_layers * ((gemma && !gemma3) ? 2 : 1); x->gemma = gemma; x->gemma3 = gemma3;
  x->NEL = (moe && c.ep_size > 1) ? c.num_experts / c.ep_size : (moe ? c.num_experts : 1);
  x->E0 = (moe && c.ep_size > 1) ? c.ep_rank * x->NEL : 0;
  x->moe = moe; x->NE = moe ? c.num_experts : 1; x->TK = moe ? c.num_experts_per_tok : 1; x->P = c.pagedattn_tokens_per_page; x->B = c.max_slots;
  x->QN = x->Q * x->D; x->KN = x->KV * x->D; x->NQKV = x->QN + 2 * x->KN; x->Tmax = c.max_target_length;
  x->mpg = c.pagedattn_max_pages_per_group; x->GQ = gq;
  x->nsb_max = (x->mpg + IXT_SB - 1) / IXT_SB;
  x->Rmax = c.max_prefill_predict_length > c.max_slots ? c.max_prefill_predict_length : c.max_slots;
  if (c.max_packed_rows > x->Rmax) x->Rmax = c.max_packed_rows;
  x->S = c.staging_groups; x->G = x->B + x->S;
  x->NLmax = x->B > x->S ? x->B : x->S;
  if (x->NLmax < 1) x->NLmax = 1;
  const int E = x->E, F = x->F, V = x->V, L = x->L, R = x->Rmax;
  *out = nullptr;
  int rc = IXT_OK;
  auto fail = [&](int code) { ixt_destroy(x); return code; };
#define A(expr) do { rc = (expr); if (rc != IXT_OK) return fail(rc); } while (0)
  A(dmalloc(x, &x->embedding, (size_t)V * E));
  A(dmalloc(x, &x->final_norm, (size_t)E));
  A(dmalloc(x, &x->logits_w, (size_t)E * V));
  x->layers.resize(L);
  x->loaded.assign(L + 1, 0);
  for (int i = 0; i < L; ++i) {
    LayerW& w = x->layers[i];
    A(dmalloc(x, &w.ln1, (size_t)E));
    A(dmalloc(x, &w.wqkv, (size_t)E * x->NQKV));
    A(dmalloc(x, &w.wo, (size_t)x->QN * E));
    A(dmalloc(x, &w.ln2, (size_t)E));
    if (moe) {
      A(dmalloc(x, &w.gate, (size_t)E * x->NE));
      // all experts' kernels of a layer in ONE allocation each, a fixed stride apart: the routed launches index them by expert on the device
      w.ewi.assign(x->NEL, nullptr); w.ewout.assign(x->NEL, nullptr);
      x->moe_wi_stride = (size_t)E * 2 * F; x->moe_wo_stride = (size_t)F * E;
      bf16_t *wi_all = nullptr, *wo_all = nullptr;
      A(dmalloc(x, &wi_all, x->moe_wi_stride * x->NEL));
      A(dmalloc(x, &wo_all, x->moe_wo_stride * x->NEL));
      for (int e = 0; e < x->NEL; ++e) { w.ewi[e] = wi_all + (size_t)e * x->moe_wi_stride; w.ewout[e] = wo_all + (size_t)e * x->moe_wo_stride; }
      w.wi = w.ewi[0]; w.wout = w.ewout[0];
    } else {
      A(dmalloc(x, &w.wi, (size_t)E * 2 * F));
      A(dmalloc(x, &w.wout, (size_t)F * E));
    }
    if (gemma && c.use_post_attn_norm) A(dmalloc(x, &w.pan, (size_t)E));
    if (gemma && c.use_post_ffw_norm) A(dmalloc(x, &w.pfn, (size_t)E));
    if (gemma3) { A(dmalloc(x, &w.qn, (size_t)x->D)); A(dmalloc(x, &w.kn, (size_t)x->D)); }
  }
  if (gemma) A(dmalloc(x, &x->glogits, (size_t)x->NLmax * V));
  if (moe) {
    A(dmalloc(x, &x->moe_idx, (size_t)R * x->TK));
    A(dmalloc(x, &x->moe_w, (size_t)R * x->TK));
    A(dmalloc(x, &x->moe_y, (size_t)R * x->TK * E));
    // copy rows: R TK copies, every expert's range padded to whole row groups of 128 rows (16 when there are at most 64 copies)
    const int cap128 = (R * x->TK + 127) / 128 * 128 + 128 * x->NEL, cap16 = 64 + 16 * x->NEL;
    x->moe_rows_cap = cap128 > cap16 ? cap128 : cap16;
    A(dmalloc(x, &x->moe_pos, (size_t)R * x->TK));
    A(dmalloc(x, &x->moe_rbexp, (size_t)x->moe_rows_cap / 16));
    A(dmalloc(x, &x->moe_cnt, (size_t)2 * x->NEL));
    A(dmalloc(x, &x->moe_xg, (size_t)x->moe_rows_cap * E));
    A(dmalloc(x, &x->moe_hact, (size_t)x->moe_rows_cap * F));
    A(dmalloc(x, &x->moe_comb, (size_t)R * E));
  }
  // RoPE tables
  host_rope_tables(c, c.rope_max_timescale, x->rope_sin_h, x->rope_cos_h);
  x->rope32 = c.rope_llama31 != 0;
  if (x->rope32) {
    if (gemma || moe) return fail(set_err(IXT_ERR_UNSUPPORTED, "rope_type llama3.1 is built for the llama2 block"));
    host_rope_tables_llama31(c, x->rope_sin32_h, x->rope_cos32_h);
    float *s32 = nullptr, *c32 = nullptr;
    A(dmalloc(x, &s32, x->rope_sin32_h.size()));
    A(dmalloc(x, &c32, x->rope_cos32_h.size()));
    if (hipMemcpy(s32, x->rope_sin32_h.data(), x->rope_sin32_h.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(c32, x->rope_cos32_h.data(), x->rope_cos32_h.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
      return fail(set_err(IXT_ERR_HIP, "rope table upload failed"));
    x->rope_sin = reinterpret_cast<bf16_t*>(s32);     // the kernels read them as f32 when AttnGeom.rope32 is set (ixt_rope_pair)
    x->rope_cos = reinterpret_cast<bf16_t*>(c32);
  } else {
  A(dmalloc(x, &x->rope_sin, x->rope_sin_h.size()));
  A(dmalloc(x, &x->rope_cos, x->rope_cos_h.size()));
  if (hipMemcpy(x->rope_sin, x->rope_sin_h.data(), x->rope_sin_h.size() * 2, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(x->rope_cos, x->rope_cos_h.data(), x->rope_cos_h.size() * 2, hipMemcpyHostToDevice) != hipSuccess)
    return fail(set_err(IXT_ERR_HIP, "rope table upload failed"));
  }
  if (gemma3 && c.local_rope_max_timescale > 0.0f) {   // attentions.py:2085-2088
    std::vector<uint16_t> sl, cl;
    host_rope_tables(c, c.local_rope_max_timescale, sl, cl);
    A(dmalloc(x, &x->rope_sin_local, sl.size()));
    A(dmalloc(x, &x->rope_cos_local, cl.size()));
    if (hipMemcpy(x->rope_sin_local, sl.data(), sl.size() * 2, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(x->rope_cos_local, cl.data(), cl.size() * 2, hipMemcpyHostToDevice) != hipSuccess)
      return fail(set_err(IXT_ERR_HIP, "rope table upload failed"));
  }
  // page pools
  x->layer_page_elems = (size_t)c.pagedattn_num_pages * x->KV * x->P * x->D;
  x->kpages_l.assign(L, nullptr);
  x->vpages_l.assign(L, nullptr);
  bool zero_ok = true;
  for (int l = 0; l < L; ++l) {
    A(dmalloc(x, &x->kpages_l[l], x->layer_page_elems));
    A(dmalloc(x, &x->vpages_l[l], x->layer_page_elems));
    zero_ok = zero_ok && hipMemset(x->kpages_l[l], 0, x->layer_page_elems * 2) == hipSuccess && hipMemset(x->vpages_l[l], 0, x->layer_page_elems * 2) == hipSuccess;
  }
  if (!zero_ok)
    return fail(set_err(IXT_ERR_HIP, "page pool memset failed"));
  // page state
  PageStateDev& s = x->ps;
  s.num_pages = c.pagedattn_num_pages; s.groups = x->G; s.decode_groups = x->B; s.mpg = x->mpg; s.P = x->P;
  A(dmalloc(x, &s.status, (size_t)4));
  A(dmalloc(x, &s.meta, (size_t)2));
  A(dmalloc(x, &s.page_status, (size_t)s.num_pages));
  A(dmalloc(x, &s.page_map, (size_t)s.groups * s.mpg));
  A(dmalloc(x, &s.num_pages_used, (size_t)s.groups));
  A(dmalloc(x, &s.sequence_lengths, (size_t)s.groups));
  A(dmalloc(x, &s.active_page, (size_t)s.groups));
  A(dmalloc(x, &s.has_active_page, (size_t)s.groups));
  A(dmalloc(x, &s.active_page_position, (size_t)s.groups));
  // workspaces
  A(dmalloc(x, &x->h_a, (size_t)R * E));
  A(dmalloc(x, &x->h_b, (size_t)R * E));
  const size_t Rp = (size_t)((R + 15) / 16) * 16;  // k-major activations are padded to whole 16-row blocks
  A(dmalloc(x, &x->xn, Rp * E));
  A(dmalloc(x, &x->qbuf, (size_t)R * x->QN));
  {
    const int mpgp = (x->mpg + 3) & ~3;
    x->split_stride = (size_t)x->GQ * 16 + (size_t)x->GQ * mpgp + (size_t)x->GQ * x->nsb_max * x->D;
    A(dmalloc(x, &x->split_cnt, (size_t)16 * x->KV * 4 + 64));
    A(dmalloc(x, &x->split_ws, (size_t)16 * x->KV * x->split_stride));
    HIPCHK(hipMemset(x->split_cnt, 0, sizeof(int32_t) * ((size_t)16 * x->KV * 4 + 64)));
  }
  A(dmalloc(x, &x->ao, Rp * x->QN));
  A(dmalloc(x, &x->hact, Rp * F));
  {
    // partial slices: SPLIT launches write ceil(K/512) slices of [R][N]; FULL launches one slice of [R][N]
    size_t nmax = (size_t)x->NQKV > (size_t)2 * F ? x->NQKV : (size_t)2 * F;
    if ((size_t)E > nmax) nmax = E;
    size_t full = (size_t)R * nmax;
    size_t gmaxn = 0;
    auto upd = [&](int K, int N) { size_t v = (size_t)((K + IXT_CHAIN - 1) / IXT_CHAIN) * N; if (v > gmaxn) gmaxn = v; };
    upd(E, x->NQKV); upd(x->QN, E); upd(E, 2 * F); upd(F, E);
    size_t split = gmaxn * (size_t)R;  // generous: any R may use SPLIT
    if (moe) {   // routed experts: up to 64 + 16 NE copy rows on the split-K form, moe_rows_cap rows of one slice on the ring form
      const size_t s16 = gmaxn * (size_t)(64 + 16 * x->NEL), s128 = (size_t)x->moe_rows_cap * E;
      if (s16 > split) split = s16;
      if (s128 > full) full = s128;
    }
    x->part_floats = full > split ? full : split;
    A(dmalloc(x, &x->part, x->part_floats));
  }
  A(dmalloc(x, &x->scores, (size_t)R * x->Q * x->Tmax));
  A(dmalloc(x, &x->pvpart, (size_t)R * x->Q * x->nsb_max * x->D));
  x->amax_cap = (size_t)x->NLmax * ((V + 15) / 16);
  A(dmalloc(x, &x->amax_val, x->amax_cap));
  A(dmalloc(x, &x->amax_idx, x->amax_cap));
  {
    const int kmax = F > x->QN ? (F > E ? F : E) : (x->QN > E ? x->QN : E);
    x->xt_cap = (size_t)((R + 15) / 16) * 16 * (kmax > 8192 ? kmax : 8192);
    A(dmalloc(x, &x->xt_op, x->xt_cap));
  }
  x->fp8 = c.fp8_weights != 0;
  if (x->fp8) {
    if (moe && c.ep_size > 1) return fail(set_err(IXT_ERR_UNSUPPORTED, "fp8 weights with expert-sharded contexts are not built"));
    const int kmax8 = F > x->QN ? (F > E ? F : E) : (x->QN > E ? x->QN : E);
    const size_t rows8 = std::max(Rp, (size_t)x->moe_rows_cap);     // the routed experts quantise their copy rows (incl. alignment padding)
    A(dmalloc(x, &x->x8, rows8 * (size_t)kmax8));
    A(dmalloc(x, &x->xs8, rows8 * (size_t)((kmax8 + IXT_CHAIN - 1) / IXT_CHAIN)));
    auto twin = [&](const bf16_t* w, int K_, int N_) -> int {
      ixt_ctx::F8W t{nullptr, nullptr, K_, N_};
      int r = dmalloc(x, &t.w8, (size_t)K_ * N_);
      if (r == IXT_OK) r = dmalloc(x, &t.s, (size_t)N_);
      x->f8w[w] = t;
      return r;
    };
    for (int i = 0; i < L; ++i) {
      LayerW& w = x->layers[i];
      A(twin(w.wqkv, E, x->NQKV));
      A(twin(w.wo, x->QN, E));
      if (moe) {   // all experts' twins of a layer in one allocation each, fixed strides apart (the routed launches index them by expert on the device)
        uint8_t *wi8 = nullptr, *wo8 = nullptr;
        float *si = nullptr, *so = nullptr;
        A(dmalloc(x, &wi8, (size_t)x->NEL * E * 2 * F)); A(dmalloc(x, &wo8, (size_t)x->NEL * F * E));
        A(dmalloc(x, &si, (size_t)x->NEL * 2 * F)); A(dmalloc(x, &so, (size_t)x->NEL * E));
        for (int e = 0; e < x->NEL; ++e) {
          x->f8w[w.ewi[e]] = ixt_ctx::F8W{wi8 + (size_t)e * E * 2 * F, si + (size_t)e * 2 * F, E, 2 * F};
          x->f8w[w.ewout[e]] = ixt_ctx::F8W{wo8 + (size_t)e * F * E, so + (size_t)e * E, F, E};
        }
      } else {
        A(twin(w.wi, E, 2 * F));
        A(twin(w.wout, F, E));
      }
    }
  }
  A(dmalloc(x, &x->row_slot, (size_t)R));
  A(dmalloc(x, &x->row_pos, (size_t)R));
  A(dmalloc(x, &x->row_active, (size_t)R));
  A(dmalloc(x, &x->row_tok, (size_t)R));
  A(dmalloc(x, &x->new_tok, (size_t)R));
  A(dmalloc(x, &x->lrows, (size_t)x->NLmax));
  A(dmalloc(x, &x->kptrs_d, (size_t)L));
  A(dmalloc(x, &x->vptrs_d, (size_t)L));
  if (hipMemcpy(x->kptrs_d, x->kpages_l.data(), sizeof(bf16_t*) * L, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(x->vptrs_d, x->vpages_l.data(), sizeof(bf16_t*) * L, hipMemcpyHostToDevice) != hipSuccess)
    return fail(set_err(IXT_ERR_HIP, "page pointer upload failed"));
#undef A
  AttnGeom& g = x->geom;
  g.nq = x->Q; g.nkv = x->KV; g.hd = x->D; g.P = x->P; g.num_pages = c.pagedattn_num_pages; g.mpg = x->mpg;
  g.Tmax = x->Tmax; g.nsb_max = x->nsb_max; g.rope32 = x->rope32 ? 1 : 0;
  hipLaunchKernelGGL(k_pages_reset, dim3(1), dim3(64), 0, 0, x->ps);
  if (hipDeviceSynchronize() != hipSuccess) return fail(set_err(IXT_ERR_HIP, "page state init failed"));
  if (sdot_set_attributes() != IXT_OK) return fail(IXT_ERR_HIP);
  *out = x;
  return IXT_OK;
}

extern "C" int ixt_destroy(ixt_ctx* c) {
  if (!c) return IXT_OK;
  (void)hipDeviceSynchronize();
  for (auto& kv : c->graphs) (void)hipGraphExecDestroy(kv.second.exec);
  for (auto& t : c->tpairs) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
  for (void* p : c->allocs) (void)hipFree(p);
  if (c->wt_op) (void)hipFree(c->wt_op);
  delete c;
  return IXT_OK;
}

// --------------------------------------------------------------------------------------- weights
static int copy_param(const void* src, int dtype, bf16_t* dst, int64_t rows, int64_t cols, int64_t ldd, hipStream_t st) {
  const int64_t n = rows * cols;
  const int blocks = (int)((n + 255) / 256);
  if (dtype == IXT_BF16) hipLaunchKernelGGL((k_copy_convert<bf16_t>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)src, dst, rows, cols, ldd);
  else if (dtype == IXT_F32) hipLaunchKernelGGL((k_copy_convert<float>), dim3(blocks), dim3(256), 0, st, (const float*)src, dst, rows, cols, ldd);
  else return set_err(IXT_ERR_INVALID, "dtype %d", dtype);
  LAUNCHCHK();
  return IXT_OK;
}
// a [K][ncols] kernel of the reference (row-major, linears.py:205-217) -> columns n_off.. of the tiled matrix `dst` (widx)
// mode 0: columns n_off..; mode 1: Q / K heads of hd columns, pair-interleaved; mode 2: MLP gate tiles interleaved (n_off = 0 wi_0, 1 wi_1)
static int tile_param(const void* src, int dtype, bf16_t* dst, int K, int ncols, int n_off, hipStream_t st, int mode = 0, int hd = 0, int ring = 0) {
  if (ncols % 16 || (mode != 2 && n_off % 16) || K % IXT_MK) return set_err(IXT_ERR_UNSUPPORTED, "kernel %d x %d: columns %% 16 or rows %% 32", K, ncols);
  const int64_t n = (int64_t)K * ncols;
  const unsigned blocks = (unsigned)((n + 255) / 256);
  if (dtype == IXT_BF16) hipLaunchKernelGGL((k_tile_w<bf16_t>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)src, dst, K, ncols, n_off, mode, hd, ring);
  else if (dtype == IXT_F32) hipLaunchKernelGGL((k_tile_w<float>), dim3(blocks), dim3(256), 0, st, (const float*)src, dst, K, ncols, n_off, mode, hd, ring);
  else return set_err(IXT_ERR_INVALID, "dtype %d", dtype);
  LAUNCHCHK();
  return IXT_OK;
}

extern "C" int ixt_load_weight(ixt_ctx* c, const char* path, const void* dev_ptr, int dtype, const int64_t* shape, int ndim, void* stream) {
  if (!c || !path || !dev_ptr || !shape) return set_err(IXT_ERR_INVALID, "null argument");
  hipStream_t st = (hipStream_t)stream;
  c->fp8_ready = false;          // the e4m3 twins are made from the bf16 copies on the next forward pass
  std::string p(path);
  if (p.rfind("params/", 0) == 0) p = p.substr(7);
  int64_t numel = 1;
  for (int i = 0; i < ndim; ++i) numel *= shape[i];
  const int E = c->E, F = c->F, V = c->V;
  auto expect = [&](int64_t n) -> int {
    if (numel != n) return set_err(IXT_ERR_INVALID, "%s: %lld elements, expected %lld", path, (long long)numel, (long long)n);
    return IXT_OK;
  };
  if (p == "token_embedder/embedding") {
    TRY(expect((int64_t)V * E)); TRY(copy_param(dev_ptr, dtype, c->embedding, V, E, E, st)); c->loaded[c->L] |= 1;
    if (c->cfg.logits_via_embedding) {   // decoders.py:552-558: the transposed table is the logits kernel
      bf16_t* tmp = nullptr;
      HIPCHK(hipMalloc((void**)&tmp, (size_t)V * E * sizeof(bf16_t)));
      hipLaunchKernelGGL(k_transpose_ve, dim3((V + 31) / 32, (E + 31) / 32), dim3(32, 8), 0, st, c->embedding, tmp, V, E);
      int rc = tile_param(tmp, IXT_BF16, c->logits_w, E, V, 0, st);
      (void)hipStreamSynchronize(st);
      (void)hipFree(tmp);
      TRY(rc);
      c->loaded[c->L] |= 4;
    }
    return IXT_OK;
  }
  if (p == "decoder/decoder_norm/scale") {
    TRY(expect(E)); TRY(copy_param(dev_ptr, dtype, c->final_norm, 1, E, E, st)); c->loaded[c->L] |= 2; return IXT_OK;
  }
  if (p == "decoder/logits_dense/kernel") {
    TRY(expect((int64_t)E * V)); TRY(tile_param(dev_ptr, dtype, c->logits_w, E, V, 0, st)); c->loaded[c->L] |= 4; return IXT_OK;
  }
  int li = -1, consumed = 0;
  if (sscanf(p.c_str(), "decoder/layers_%d/%n", &li, &consumed) < 1 || li < 0 || li >= c->L || consumed == 0)
    return set_err(IXT_ERR_INVALID, "unknown parameter path %s", path);
  std::string rest = p.substr(consumed);
  if (c->gemma3) {
    // layers/gemma3.py:83-200: llama2's spellings except the attention pre-norm; the post norms and the head norms are Gemma's own
    if (rest == "pre_self_attention_norm/scale") rest = "pre_self_attention_layer_norm/scale";
    LayerW& gw = c->layers[li];
    if (rest == "post_self_attention_norm/scale" && gw.pan) { TRY(expect(E)); TRY(copy_param(dev_ptr, dtype, gw.pan, 1, E, E, st)); c->loaded[li] |= 1u << W_PAN; return IXT_OK; }
    if (rest == "post_ffw_norm/scale" && gw.pfn) { TRY(expect(E)); TRY(copy_param(dev_ptr, dtype, gw.pfn, 1, E, E, st)); c->loaded[li] |= 1u << W_PFN; return IXT_OK; }
    if (rest == "self_attention/query_norm/scale") { TRY(expect(c->D)); TRY(copy_param(dev_ptr, dtype, gw.qn, 1, c->D, c->D, st)); c->loaded[li] |= 1u << W_QN; return IXT_OK; }
    if (rest == "self_attention/key_norm/scale") { TRY(expect(c->D)); TRY(copy_param(dev_ptr, dtype, gw.kn, 1, c->D, c->D, st)); c->loaded[li] |= 1u << W_KN; return IXT_OK; }
  } else if (c->gemma) {
    // layers/gemma2.py:69-236: reference layer li holds a "local" and a "global" (attention, MLP) pair = sub-layers 2 li and 2 li + 1.
    // The names are mapped onto the llama2 spellings below; the post norms are Gemma's own.
    if (li >= c->L / 2) return set_err(IXT_ERR_INVALID, "unknown parameter path %s", path);
    int sub = -1;
    const char* tags[2] = {"_local", "_global"};
    for (int t = 0; t < 2 && sub < 0; ++t) {
      const std::string tag(tags[t]);
      const size_t slash = rest.find('/');
      const std::string head = rest.substr(0, slash);
      if (head.size() > tag.size() && head.compare(head.size() - tag.size(), tag.size(), tag) == 0) {
        sub = t;
        rest = head.substr(0, head.size() - tag.size()) + (slash == std::string::npos ? "" : rest.substr(slash));
      }
    }
    if (sub < 0) return set_err(IXT_ERR_INVALID, "unknown parameter path %s", path);
    li = 2 * li + sub;
    if (rest == "pre_self_attention_norm/scale") rest = "pre_self_attention_layer_norm/scale";
    else if (rest == "pre_ffw_norm/scale") rest = "mlp/mlp_layer_norm/scale";
    LayerW& gw = c->layers[li];
    if (rest == "post_self_attention_norm/scale" && gw.pan) { TRY(expect(E)); TRY(copy_param(dev_ptr, dtype, gw.pan, 1, E, E, st)); c->loaded[li] |= 1u << W_PAN; return IXT_OK; }
    if (rest == "post_ffw_norm/scale" && gw.pfn) { TRY(expect(E)); TRY(copy_param(dev_ptr, dtype, gw.pfn, 1, E, E, st)); c->loaded[li] |= 1u << W_PFN; return IXT_OK; }
  }
  LayerW& w = c->layers[li];
  uint32_t& m = c->loaded[li];
  if (c->moe) {   // layers/moe.py:255-312: MoeBlock_0/{gate/kernel [emb, experts], wi_0 / wi_1 [experts, emb, mlp], wo [experts, mlp, emb]}
    const size_t esz = dtype == IXT_F32 ? 4 : 2;
    const char* src = (const char*)dev_ptr;
    if (rest == "MoeBlock_0/gate/kernel") { TRY(expect((int64_t)E * c->NE)); TRY(copy_param(dev_ptr, dtype, w.gate, E, c->NE, c->NE, st)); m |= 1u << W_GATE; return IXT_OK; }
    if (rest == "MoeBlock_0/wi_0" || rest == "MoeBlock_0/wi_1") {
      const int which = rest == "MoeBlock_0/wi_1" ? 1 : 0;
      TRY(expect((int64_t)c->NE * E * F));
      for (int e = 0; e < c->NEL; ++e) TRY(tile_param(src + (size_t)(c->E0 + e) * E * F * esz, dtype, w.ewi[e], E, F, which, st, 2));
      m |= 1u << (which ? W_WI1 : W_WI0);
      return IXT_OK;
    }
    if (rest == "MoeBlock_0/wo") {
      TRY(expect((int64_t)c->NE * F * E));
      for (int e = 0; e < c->NEL; ++e) TRY(tile_param(src + (size_t)(c->E0 + e) * F * E * esz, dtype, w.ewout[e], F, E, 0, st));
      m |= 1u << W_WOUT;
      return IXT_OK;
    }
    if (rest.rfind("mlp/", 0) == 0) return set_err(IXT_ERR_INVALID, "unknown parameter path %s (decoder_block mixtral has MoeBlock_0, no mlp)", path);
  }
  if (rest == "pre_self_attention_layer_norm/scale") { TRY(expect(E)); TRY(copy_param(dev_ptr, dtype, w.ln1, 1, E, E, st)); m |= 1u << W_LN1; }
  else if (rest == "self_attention/query/kernel") { TRY(expect((int64_t)E * c->QN)); TRY(tile_param(dev_ptr, dtype, w.wqkv, E, c->QN, 0, st, c->rope32 ? 3 : 1, c->D)); m |= 1u << W_Q; }
  else if (rest == "self_attention/key/kernel") { TRY(expect((int64_t)E * c->KN)); TRY(tile_param(dev_ptr, dtype, w.wqkv, E, c->KN, c->QN, st, c->rope32 ? 3 : 1, c->D)); m |= 1u << W_K; }
  else if (rest == "self_attention/value/kernel") { TRY(expect((int64_t)E * c->KN)); TRY(tile_param(dev_ptr, dtype, w.wqkv, E, c->KN, c->QN + c->KN, st)); m |= 1u << W_V; }
  else if (rest == "self_attention/out/kernel") { TRY(expect((int64_t)c->QN * E)); TRY(tile_param(dev_ptr, dtype, w.wo, c->QN, E, 0, st)); m |= 1u << W_O; }
  else if (rest == "mlp/mlp_layer_norm/scale" || rest == "post_self_attention_layer_norm/scale") { TRY(expect(E)); TRY(copy_param(dev_ptr, dtype, w.ln2, 1, E, E, st)); m |= 1u << W_LN2; }
  else if (rest == "mlp/wi_0/kernel") { TRY(expect((int64_t)E * F)); TRY(tile_param(dev_ptr, dtype, w.wi, E, F, 0, st, 2)); m |= 1u << W_WI0; }
  else if (rest == "mlp/wi_1/kernel") { TRY(expect((int64_t)E * F)); TRY(tile_param(dev_ptr, dtype, w.wi, E, F, 1, st, 2)); m |= 1u << W_WI1; }
  else if (rest == "mlp/wo/kernel") { TRY(expect((int64_t)F * E)); TRY(tile_param(dev_ptr, dtype, w.wout, F, E, 0, st)); m |= 1u << W_WOUT; }
  else return set_err(IXT_ERR_INVALID, "unknown parameter path %s", path);
  return IXT_OK;
}

extern "C" int ixt_weights_complete(ixt_ctx* c) {
  if (!c) return 0;
  for (int i = 0; i < c->L; ++i) {
    uint32_t want = (1u << W_PER_LAYER) - 1;
    if (c->layers[i].pan) want |= 1u << W_PAN;
    if (c->layers[i].pfn) want |= 1u << W_PFN;
    if (c->layers[i].qn) want |= (1u << W_QN) | (1u << W_KN);
    if (c->layers[i].gate) want |= 1u << W_GATE;
    if (c->loaded[i] != want) return 0;
  }
  return c->loaded[c->L] == 7 ? 1 : 0;
}

// --------------------------------------------------------------------------------------- page state
static int check_slot(ixt_ctx* c, int slot) {
  if (slot < 0 || slot >= c->B) return set_err(IXT_ERR_RANGE, "PageManager: page_group_id (%d) out of range [0, %d)", slot, c->B);
  return IXT_OK;
}
// decode slots and the staging groups behind them (packed prefill)
static int check_group(ixt_ctx* c, int g) {
  if (g < 0 || g >= c->G) return set_err(IXT_ERR_RANGE, "PageManager: page_group_id (%d) out of range [0, %d)", g, c->G);
  return IXT_OK;
}
extern "C" int ixt_pages_reset(ixt_ctx* c, void* stream) {
  if (!c) return set_err(IXT_ERR_INVALID, "null ctx");
  hipLaunchKernelGGL(k_pages_reset, dim3(1), dim3(64), 0, (hipStream_t)stream, c->ps);
  LAUNCHCHK();
  return IXT_OK;
}
extern "C" int ixt_pages_reserve(ixt_ctx* c, int slot, int true_length, void* stream) {
  if (!c) return set_err(IXT_ERR_INVALID, "null ctx");
  TRY(check_group(c, slot));
  if (true_length <= 0 || true_length > c->Tmax) return set_err(IXT_ERR_RANGE, "PageManager: true_length (%d) out of range (0, %d]", true_length, c->Tmax);
  hipLaunchKernelGGL(k_pages_reserve, dim3(1), dim3(64), 0, (hipStream_t)stream, c->ps, slot, true_length, 0);
  LAUNCHCHK();
  return IXT_OK;
}
extern "C" int ixt_pages_extend(ixt_ctx* c, int slot, int total_length, void* stream) {
  if (!c) return set_err(IXT_ERR_INVALID, "null ctx");
  TRY(check_group(c, slot));
  if (total_length <= 0 || total_length > c->Tmax) return set_err(IXT_ERR_RANGE, "PageManager: true_length (%d) out of range (0, %d]", total_length, c->Tmax);
  hipLaunchKernelGGL(k_pages_reserve, dim3(1), dim3(64), 0, (hipStream_t)stream, c->ps, slot, total_length, 1);
  LAUNCHCHK();
  return IXT_OK;
}
extern "C" int ixt_pages_advance(ixt_ctx* c, void* stream) {
  if (!c) return set_err(IXT_ERR_INVALID, "null ctx");
  hipLaunchKernelGGL(k_pages_advance, dim3(1), dim3(64), 0, (hipStream_t)stream, c->ps);
  LAUNCHCHK();
  return IXT_OK;
}
extern "C" int ixt_pages_release(ixt_ctx* c, int slot, void* stream) {
  if (!c) return set_err(IXT_ERR_INVALID, "null ctx");
  TRY(check_group(c, slot));
  hipLaunchKernelGGL(k_pages_release, dim3(1), dim3(64), 0, (hipStream_t)stream, c->ps, slot);
  LAUNCHCHK();
  return IXT_OK;
}
extern "C" int ixt_pages_set_active(ixt_ctx* c, int slot, int active, void* stream) {
  if (!c) return set_err(IXT_ERR_INVALID, "null ctx");
  TRY(check_slot(c, slot));
  hipLaunchKernelGGL(k_pages_set_active, dim3(1), dim3(1), 0, (hipStream_t)stream, c->ps, slot, active ? 1 : 0);
  LAUNCHCHK();
  return IXT_OK;
}
extern "C" int ixt_pages_resync(ixt_ctx* c, int slot, int length, void* stream) {
  if (!c) return set_err(IXT_ERR_INVALID, "null ctx");
  TRY(check_slot(c, slot));
  if (length <= 0 || length > c->Tmax) return set_err(IXT_ERR_RANGE, "PageManager: true_length (%d) out of range (0, %d]", length, c->Tmax);
  hipLaunchKernelGGL(k_pages_resync, dim3(1), dim3(1), 0, (hipStream_t)stream, c->ps, slot, length);
  LAUNCHCHK();
  return IXT_OK;
}
extern "C" int ixt_pages_move(ixt_ctx* c, int src_group, int dst_slot, void* stream) {
  if (!c) return set_err(IXT_ERR_INVALID, "null ctx");
  TRY(check_group(c, src_group));
  TRY(check_slot(c, dst_slot));
  if (src_group == dst_slot) return IXT_OK;
  hipLaunchKernelGGL(k_pages_move, dim3(1), dim3(64), 0, (hipStream_t)stream, c->ps, src_group, dst_slot);
  LAUNCHCHK();
  return IXT_OK;
}
extern "C" int ixt_pages_copy(ixt_ctx* c, int src_group, int dst_slot, int length, void* stream) {
  if (!c) return set_err(IXT_ERR_INVALID, "null ctx");
  TRY(check_group(c, src_group));
  TRY(check_slot(c, dst_slot));
  if (length <= 0 || length > c->Tmax) return set_err(IXT_ERR_RANGE, "PageManager: true_length (%d) out of range (0, %d]", length, c->Tmax);
  if (src_group == dst_slot) return set_err(IXT_ERR_INVALID, "page copy onto itself");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_pages_reserve, dim3(1), dim3(64), 0, st, c->ps, dst_slot, length, 0);
  const int npages = (length + c->P - 1) / c->P;
  hipLaunchKernelGGL(k_pages_copy, dim3(npages, c->L), dim3(256), 0, st, c->ps, src_group, dst_slot, c->kptrs_d, c->vptrs_d,
                     (size_t)c->KV * c->P * c->D);
  LAUNCHCHK();
  return IXT_OK;
}
extern "C" int ixt_pages_status(ixt_ctx* c, int32_t* out4, int reset, void* stream) {
  if (!c || !out4) return set_err(IXT_ERR_INVALID, "null argument");
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(hipMemcpyAsync(out4, c->ps.status, sizeof(int32_t) * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  if (reset) HIPCHK(hipMemsetAsync(c->ps.status, 0, sizeof(int32_t) * 4, st));
  return IXT_OK;
}
extern "C" int ixt_exchange_status(ixt_ctx* c, int32_t* out1, int reset, void* stream) {
  if (!c || !out1) return set_err(IXT_ERR_INVALID, "null argument");
  hipStream_t st = (hipStream_t)stream;
  int32_t* err = c->split_cnt + (size_t)16 * c->KV * 4;
  HIPCHK(hipMemcpyAsync(out1, err, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  if (reset && *out1 != 0) HIPCHK(hipMemsetAsync(c->split_cnt, 0, sizeof(int32_t) * ((size_t)16 * c->KV * 4 + 4), st));   // a wait that gave up leaves its counters armed
  return IXT_OK;
}
extern "C" int ixt_pages_status_async(ixt_ctx* c, int32_t* host_out4, void* stream) {
  if (!c || !host_out4) return set_err(IXT_ERR_INVALID, "null argument");
  HIPCHK(hipMemcpyAsync(host_out4, c->ps.status, sizeof(int32_t) * 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return IXT_OK;
}
#ifndef IXT_SOURCE_HASH
#define IXT_SOURCE_HASH "unhashed"
#endif
extern "C" const char* ixt_build_id(void) { return IXT_SOURCE_HASH; }

extern "C" int ixt_pages_export(ixt_ctx* c, int32_t* page_status, int32_t* page_map, int32_t* num_pages_used, int32_t* sequence_lengths,
                                int32_t* active_page, uint8_t* has_active_page, int32_t* active_page_position, void* stream) {
  if (!c) return set_err(IXT_ERR_INVALID, "null ctx");
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(hipStreamSynchronize(st));
  const PageStateDev& s = c->ps;
  if (page_status) HIPCHK(hipMemcpy(page_status, s.page_status, sizeof(int32_t) * s.num_pages, hipMemcpyDeviceToHost));
  const int ng = s.decode_groups;   // the reference's page groups; staging groups are internal
  if (page_map) HIPCHK(hipMemcpy(page_map, s.page_map, sizeof(int32_t) * ng * s.mpg, hipMemcpyDeviceToHost));
  if (num_pages_used) HIPCHK(hipMemcpy(num_pages_used, s.num_pages_used, sizeof(int32_t) * ng, hipMemcpyDeviceToHost));
  if (sequence_lengths) HIPCHK(hipMemcpy(sequence_lengths, s.sequence_lengths, sizeof(int32_t) * ng, hipMemcpyDeviceToHost));
  if (active_page) HIPCHK(hipMemcpy(active_page, s.active_page, sizeof(int32_t) * ng, hipMemcpyDeviceToHost));
  if (active_page_position) HIPCHK(hipMemcpy(active_page_position, s.active_page_position, sizeof(int32_t) * ng, hipMemcpyDeviceToHost));
  if (has_active_page) {
    std::vector<int32_t> tmp(ng);
    HIPCHK(hipMemcpy(tmp.data(), s.has_active_page, sizeof(int32_t) * ng, hipMemcpyDeviceToHost));
    for (int i = 0; i < ng; ++i) has_active_page[i] = tmp[i] ? 1 : 0;
  }
  return IXT_OK;
}
extern "C" int ixt_op_logprob(ixt_ctx* c, const float* logits_dev, int rows, const int32_t* chosen_dev, float* out_dev, void* stream) {
  if (!c || !logits_dev || !chosen_dev || !out_dev) return set_err(IXT_ERR_INVALID, "null argument");
  if (rows <= 0) return set_err(IXT_ERR_RANGE, "log-prob of %d rows", rows);
  const int nblk = (c->V + 1023) / 1024;
  hipLaunchKernelGGL(k_logprob, dim3(rows), dim3(256), sizeof(float) * nblk, (hipStream_t)stream, logits_dev, c->V, chosen_dev, out_dev);
  LAUNCHCHK();
  return IXT_OK;
}

extern "C" int ixt_pages_import_status(ixt_ctx* c, const int32_t* page_status, void* stream) {
  if (!c || !page_status) return set_err(IXT_ERR_INVALID, "null argument");
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  HIPCHK(hipMemcpy(c->ps.page_status, page_status, sizeof(int32_t) * c->ps.num_pages, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_pages_recount, dim3(1), dim3(64), 0, (hipStream_t)stream, c->ps);
  LAUNCHCHK();
  return IXT_OK;
}

// --------------------------------------------------------------------------------------- launches
static int timing_begin(ixt_ctx* c, hipStream_t st, double bytes, int family = 0) {
  if (!c->timing || c->capturing) return -1;
  if (c->tused == c->tpairs.size()) {
    if (c->tpairs.size() >= 16384) return -1;
    TimingPair t;
    if (hipEventCreate(&t.a) != hipSuccess || hipEventCreate(&t.b) != hipSuccess) return -1;
    c->tpairs.push_back(t);
  }
  TimingPair& t = c->tpairs[c->tused];
  t.bytes = bytes;
  t.family = family;
  (void)hipEventRecord(t.a, st);
  return (int)c->tused++;
}
static void timing_end(ixt_ctx* c, hipStream_t st, int idx) {
  if (idx >= 0) (void)hipEventRecord(c->tpairs[idx].b, st);
}

// --------------------------------------------------------------------------------------- LDS-ring GEMM (ixt_gemm.h)
// Tiles: 1 = 128 x 128 (8 waves as 2 x 4, 4 ring slots of 32 KiB), 2 = 128 rows x 64 columns (8 waves, 5 x 24 KiB), 3 = 64 x 32 (4 waves, 6 x 24 KiB,
// 128-k chunks).
static int ring_pick(const ixt_ctx* c, int epi, int R, int N) {
  if (c->ring_cfg >= 1 && c->ring_cfg <= 3 && !(epi == GEPI_SWIGLU && c->ring_cfg == 3) && !(epi == GEPI_UNEMBED && c->ring_cfg != 1)) return c->ring_cfg;
  if (epi == GEPI_UNEMBED) return 1;
  const long b1 = (long)((N + 127) / 128) * ((R + 127) / 128), b2 = (long)((N + 63) / 64) * ((R + 127) / 128);
  if (b1 >= 192) return 1;                    // the chip has 256 CUs and a block owns one (LDS): fewer blocks leave CUs without work
  if (b2 >= 160 || epi == GEPI_SWIGLU) return 2;
  return 3;
}

template <int WR, int WC, int RB, int CT, int KS, int NS, int EPI, int SPLIT>
static int ring_launch_t(const bf16_t* x, const bf16_t* w, int K, int N, int R, int chains, int wring, const GemmEpi& e, hipStream_t st, int* wave_cols) {
  constexpr int TC = WC * CT, TR = WR * RB;
  const int ncolg = (N / 16 + TC - 1) / TC, zb = (R + 16 * TR - 1) / (16 * TR);
  constexpr size_t lds = (size_t)NS * (TC + TR) * KS * 1024;
  static bool attr_done = false;
  if (!attr_done) {   // more than 64 KiB of dynamic LDS needs the attribute once per kernel
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_ring<WR, WC, RB, CT, KS, NS, EPI, SPLIT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  dim3 grid(((ncolg + 7) / 8) * 8 * zb, SPLIT ? chains : 1, 1);
  hipLaunchKernelGGL((k_gemm_ring<WR, WC, RB, CT, KS, NS, EPI, SPLIT>), grid, dim3(WR * WC * 64), lds, st, reinterpret_cast<const u32x4*>(x),
                     reinterpret_cast<const u32x4*>(w), K, N, R, ncolg, zb, wring, e);
  LAUNCHCHK();
  if (wave_cols) *wave_cols = WC * ncolg;
  return IXT_OK;
}

// x in xidx layout, w in widx layout (wring: ring layout); the epilogue's operands in `e`.  split (GEPI_F32 only): one f32 slice per
// 512-k chain.  wave_cols (GEPI_UNEMBED): entries per row of the (max, index) arrays.
static int launch_ring(ixt_ctx* c, int epi, const bf16_t* x, const bf16_t* w, int K, int N, int R, const GemmEpi& e, bool split, hipStream_t st, int wring = 0,
                       int* wave_cols = nullptr) {
  if (K % IXT_KALIGN || N % 16) return set_err(IXT_ERR_UNSUPPORTED, "ring gemm: K %% 128 or N %% 16");
  const int chains = (K + IXT_CHAIN - 1) / IXT_CHAIN;
  const int cfg = ring_pick(c, epi, R, N);
  double ebytes = 0.0;
  switch (epi) {
    case GEPI_F32: ebytes = (double)(split ? chains : 1) * R * N * 4.0; break;
    case GEPI_UNEMBED: ebytes = e.out ? (double)R * N * 4.0 : 0.0; break;
    case GEPI_RESID: ebytes = (double)R * N * 4.0; break;
    case GEPI_SWIGLU: ebytes = (double)R * N; break;
    default: ebytes = (double)R * N * 2.0; break;
  }
  const int ti = timing_begin(c, st, (double)K * N * 2.0 + (double)R * K * 2.0 + ebytes);
  int rc = IXT_ERR_UNSUPPORTED;
#define RING_A(EPI_, SPLIT_) ring_launch_t<2, 4, 4, 2, 2, 4, EPI_, SPLIT_>(x, w, K, N, R, chains, wring, e, st, wave_cols)
#define RING_B(EPI_) ring_launch_t<2, 4, 4, 1, 2, 5, EPI_, 0>(x, w, K, N, R, chains, wring, e, st, wave_cols)
#define RING_C(EPI_) ring_launch_t<2, 2, 2, 1, 4, 6, EPI_, 0>(x, w, K, N, R, chains, wring, e, st, wave_cols)
#define RING(EPI_) do { if (cfg == 1) rc = RING_A(EPI_, 0); else if (cfg == 2) rc = RING_B(EPI_); else rc = RING_C(EPI_); } while (0)
  // Prefill-sized launches (several rounds of blocks per CU): the 128 x 128 tile with 32-k chunks in 4 slots = 64 KiB of LDS, so TWO blocks
  // share a CU and one's prologue (first chunk's latency) and epilogue run under the other's main loop: 7.6 -> 6.7 ms of GEMMs per 2048-row
  // pass.  At 256 rows a launch has at most one block per CU anyway and the 64-k chunks (half the barriers) are faster.
#define RING_P(EPI_) ring_launch_t<2, 4, 4, 2, 1, 4, EPI_, 0>(x, w, K, N, R, chains, wring, e, st, wave_cols)
  // (launches of up to 256 tiles -- O / MLP down at 2048 rows -- have one block per CU either way, and there the 64-k chunks win: 8.5 against 8.7 ms
  // per pass with 512 tiles of 128 x 64 in 72 KiB; from 384 tiles on -- QKV at 1024 rows -- the two-per-CU form wins)
  // (routed experts: every expert's rows are ONE or two row groups -- the 256-row regime per expert, whatever the launch's total: 64-k chunks.  Mixtral-8x7B
  //  256-token prefill 29.8 -> 26.7 ms)
  const bool two_per_cu = c->ring_two_per_cu && R >= 1024 && cfg == 1 && !split && e.rb_exp == nullptr && (long)((N + 127) / 128) * ((R + 127) / 128) >= 320;
  if (two_per_cu && epi == GEPI_RESID) rc = RING_P(GEPI_RESID);
  else if (two_per_cu && epi == GEPI_QKV) rc = RING_P(GEPI_QKV);
  else if (two_per_cu && epi == GEPI_SWIGLU) rc = RING_P(GEPI_SWIGLU);
  else if (two_per_cu && epi == GEPI_F32) rc = RING_P(GEPI_F32);
  else if (epi == GEPI_F32 && split) rc = RING_A(GEPI_F32, 1);
  else if (epi == GEPI_F32) RING(GEPI_F32);
  else if (epi == GEPI_RESID) RING(GEPI_RESID);
  else if (epi == GEPI_QKV) RING(GEPI_QKV);
  else if (two_per_cu && epi == GEPI_UNEMBED) rc = RING_P(GEPI_UNEMBED);
  else if (epi == GEPI_UNEMBED) {
    // the unembedding at 129..256 rows: ONE block owns all rows of its 128 columns (8 waves of 64 x 64: 256 x 128 tile, 3 slots of 48 KiB), so the
    // 1 GB kernel enters the chip once instead of once per 128-row group
    if (c->unembed_tall && R > 128 && R <= 256) rc = ring_launch_t<4, 2, 4, 4, 2, 3, GEPI_UNEMBED, 0>(x, w, K, N, R, chains, wring, e, st, wave_cols);
    else rc = RING_A(GEPI_UNEMBED, 0);
  }
  else if (epi == GEPI_SWIGLU) {
    if (cfg == 1) rc = RING_A(GEPI_SWIGLU, 0);
    else rc = ring_launch_t<2, 2, 4, 2, 2, 5, GEPI_SWIGLU, 0>(x, w, K, N, R, chains, wring, e, st, wave_cols);   // gate pairs need two tiles per wave: 4 waves
  }
#undef RING
#undef RING_A
#undef RING_B
#undef RING_C
#undef RING_P
  timing_end(c, st, ti);
  if (rc == IXT_ERR_UNSUPPORTED) return set_err(rc, "ring gemm: epilogue %d", epi);
  return rc;
}

struct DotResult { int slices; size_t slice_stride; int ntiles; };

// ---- IXT-FP8-v1 (ixt_fp8.h)
static int fp8_prepare(ixt_ctx* c, hipStream_t st) {
  if (!c->fp8 || c->fp8_ready) return IXT_OK;
  for (auto& kv : c->f8w) {
    hipLaunchKernelGGL(k_f8_quant_w, dim3(kv.second.N), dim3(256), 0, st, kv.first, kv.second.K, kv.second.N, kv.second.w8, kv.second.s);
    LAUNCHCHK();
  }
  c->fp8_ready = true;
  return IXT_OK;
}
template <int WR, int WC, int RB, int CT, int KS, int NS>
static int ring_launch8_t(const uint8_t* x8, const uint8_t* w8, int K, int N, int R, const GemmEpi& e, hipStream_t st) {
  constexpr int TC = WC * CT, TR = WR * RB;
  const int ncolg = (N / 16 + TC - 1) / TC, zb = (R + 16 * TR - 1) / (16 * TR);
  constexpr size_t lds = (size_t)NS * (TC + TR) * KS * 1024;
  static bool attr_done = false;
  if (!attr_done) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_ring<WR, WC, RB, CT, KS, NS, GEPI_F32, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  hipLaunchKernelGGL((k_gemm_ring<WR, WC, RB, CT, KS, NS, GEPI_F32, 0, true>), dim3(((ncolg + 7) / 8) * 8 * zb), dim3(WR * WC * 64), lds, st,
                     reinterpret_cast<const u32x4*>(x8), reinterpret_cast<const u32x4*>(w8), K, N, R, ncolg, zb, 0, e);
  LAUNCHCHK();
  return IXT_OK;
}
// x: bf16 rows in the x layout; t: the kernel's e4m3 twin.  out: f32 [slices][R][N] with the scales applied.
static int launch_dot8(ixt_ctx* c, const bf16_t* x, const ixt_ctx::F8W& t, int K, int N, int R, float* out, size_t out_floats, hipStream_t st, DotResult* res) {
  if (K % 64 || N % 16) return set_err(IXT_ERR_UNSUPPORTED, "fp8 dot: K %% 64 or N %% 16");
  const int chains = (K + IXT_CHAIN - 1) / IXT_CHAIN, rowblocks = (R + 15) / 16, ntile = N / 16;
  hipLaunchKernelGGL(k_f8_quant_x, dim3(chains, rowblocks), dim3(64), 0, st, x, K, R, c->x8, c->xs8);
  LAUNCHCHK();
  if (R >= c->ring_min_rows && K % IXT_KALIGN == 0 && (size_t)R * N <= out_floats) {
    GemmEpi e{};
    e.out = out; e.xscale = c->xs8; e.wscale = t.s;
    const int ti = timing_begin(c, st, (double)K * N + (double)R * K + (double)R * N * 4.0);
    int rc;
    if ((long)((N + 127) / 128) * ((R + 127) / 128) >= 150 || N % 128 == 0) rc = ring_launch8_t<2, 4, 4, 2, 2, 4>(c->x8, t.w8, K, N, R, e, st);   // 128 x 128 tiles
    else rc = ring_launch8_t<2, 4, 4, 1, 2, 5>(c->x8, t.w8, K, N, R, e, st);                                                                  // 128 x 64
    timing_end(c, st, ti);
    res->slices = 1; res->slice_stride = (size_t)R * N; res->ntiles = ntile;
    return rc;
  }
  if ((size_t)chains * R * N > out_floats) return set_err(IXT_ERR_INVALID, "fp8 dot: output workspace too small");
  const int nrb = rowblocks >= 3 ? 4 : rowblocks, zblocks = (rowblocks + nrb - 1) / nrb;
  const int ti = timing_begin(c, st, (double)K * N + (double)R * K + (double)chains * R * N * 4.0);
  dim3 grid((((ntile + 3) / 4 + 7) / 8) * 8 * zblocks, chains, 1);
  const u32x4* xp = reinterpret_cast<const u32x4*>(c->x8);
  const u32x4* wp = reinterpret_cast<const u32x4*>(t.w8);
  if (nrb == 4) hipLaunchKernelGGL((k_dotb8<4>), grid, dim3(256), (size_t)4 * 8192, st, xp, wp, c->xs8, t.s, K, N, R, out, ntile, zblocks);
  else if (nrb == 2) hipLaunchKernelGGL((k_dotb8<2>), grid, dim3(256), (size_t)2 * 8192, st, xp, wp, c->xs8, t.s, K, N, R, out, ntile, zblocks);
  else hipLaunchKernelGGL((k_dotb8<1>), grid, dim3(256), (size_t)8192, st, xp, wp, c->xs8, t.s, K, N, R, out, ntile, zblocks);
  timing_end(c, st, ti);
  LAUNCHCHK();
  res->slices = chains; res->slice_stride = (size_t)R * N; res->ntiles = ntile;
  return IXT_OK;
}

// x: bf16 in the dot x layout (xidx); w: bf16 in the tiled weight layout (widx).  out: f32 [slices][R][N]; force_full: one
// slice (all chains added inside the wave).  Up to 64 rows share each loaded weight fragment (NRB 16-row blocks per block);
// more rows run as partner blocks on the same XCD.
static int launch_dot(ixt_ctx* c, const bf16_t* x, const bf16_t* w, int K, int N, int R, float* out, size_t out_floats,
                      bool force_full, hipStream_t st, DotResult* res, bool no_ring = false) {
  if (K % IXT_MK || N % 16) return set_err(IXT_ERR_UNSUPPORTED, "dot: K %% 32 or N %% 16");
  if (out == nullptr) return set_err(IXT_ERR_INVALID, "dot: null output");
  if (c->fp8) {   // a projection kernel with an e4m3 twin: the fp8 matrix cores (everything else -- op-level kernels, the logits head -- stays bf16)
    auto it = c->f8w.find(w);
    if (it != c->f8w.end()) return launch_dot8(c, x, it->second, K, N, R, out, out_floats, st, res);
  }
  const int chains = (K + IXT_CHAIN - 1) / IXT_CHAIN;
  const int ntile = N / 16;
  const int rowblocks = (R + 15) / 16;
  const int nrb = rowblocks >= 3 ? 4 : rowblocks;
  const int zblocks = (rowblocks + nrb - 1) / nrb;
  bool split = !force_full && chains > 1;
  if (!force_full && !no_ring && R >= c->ring_min_rows && K % IXT_KALIGN == 0) {   // many rows: the LDS-ring GEMM, f32 epilogue
    const bool rsplit = c->ring_split && chains > 1 && (size_t)chains * R * N <= out_floats;
    if ((size_t)R * N <= out_floats) {
      GemmEpi e{};
      e.out = out;
      TRY(launch_ring(c, GEPI_F32, x, w, K, N, R, e, rsplit, st));
      res->slices = rsplit ? chains : 1;
      res->slice_stride = (size_t)R * N;
      res->ntiles = ntile;
      return IXT_OK;
    }
  }
  // Many rows, few chains, many columns: the streaming kernel (the unembedding's structure, block = 128 columns x all chains x 64
  // rows) adds the chain results on chip and writes ONE slice; at 256 rows the `chains` slices of a split-K launch are as large
  // as the weights.  Kept behind an option (default off): with 2 resident blocks per CU and a barrier pair per chain it measured
  // slightly SLOWER than split-K on the 2048 x 14336 and 2048 x 6144 kernels at 256 rows.  (Walking the chains inside k_dotb
  // was worse still: 102 vs 54 us.)
  // Many rows x many columns: the tiled GEMM (128 rows x 128 or 64 columns x all of K per block, one slice) once it has enough blocks
  {
    const int zb = (R + 127) / 128;
    const int ct = 4;
    const int ncolg = (ntile + 2 * ct - 1) / (2 * ct);
    if (!force_full && R >= c->gemm_min_rows && (long)ncolg * zb >= c->gemm_min_blocks && (size_t)R * N <= out_floats) {
      const int ti = timing_begin(c, st, (double)K * N * 2.0 + (double)R * K * 2.0 + (double)R * N * 4.0);
      dim3 grid(((ncolg + 7) / 8) * 8 * zb, 1, 1);
      const u32x4* xp = reinterpret_cast<const u32x4*>(x);
      const u32x4* wp = reinterpret_cast<const u32x4*>(w);
      const size_t lds = (size_t)2 * (4 * ct + 16) * 1024;
#define GEMM(NT_)                                                                                                        \
  do {                                                                                                                   \
    if (ct == 4) hipLaunchKernelGGL((k_gemm<NT_, 4, 0, 0>), grid, dim3(256), lds, st, xp, wp, K, N, R, out, ncolg, zb, (float*)nullptr, (int32_t*)nullptr, ColRanges{});        \
    else hipLaunchKernelGGL((k_gemm<NT_, 2, 0, 0>), grid, dim3(256), lds, st, xp, wp, K, N, R, out, ncolg, zb, (float*)nullptr, (int32_t*)nullptr, ColRanges{});                \
  } while (0)
      int slices_out = 1;
      if (c->gemm_prefix_halves && ct == 4 && chains >= 2 && (size_t)(1 + chains - chains / 2) * R * N <= out_floats) {
        grid.y = 2;   // first half of the chains as a running total, second half chain by chain
        slices_out = 1 + chains - chains / 2;
        if (zb == 1) hipLaunchKernelGGL((k_gemm<true, 4, 0, 2>), grid, dim3(256), lds, st, xp, wp, K, N, R, out, ncolg, zb, (float*)nullptr, (int32_t*)nullptr, ColRanges{});
        else hipLaunchKernelGGL((k_gemm<false, 4, 0, 2>), grid, dim3(256), lds, st, xp, wp, K, N, R, out, ncolg, zb, (float*)nullptr, (int32_t*)nullptr, ColRanges{});
      } else if (zb == 1) GEMM(true); else GEMM(false);
#undef GEMM
      timing_end(c, st, ti);
      LAUNCHCHK();
      res->slices = slices_out;
      res->slice_stride = (size_t)R * N;
      res->ntiles = ntile;
      return IXT_OK;
    }
  }
  // Many rows, too few 128 x 128 blocks for the full-K form: the same tiles split over the chains (one slice per chain)
  {
    const int zb = (R + 127) / 128, ncolg = (ntile + 7) / 8;
    if (!force_full && R >= c->gemm_min_rows && chains > 1 && (long)ncolg * zb * chains >= c->gemm_split_min_blocks &&
        (size_t)chains * R * N <= out_floats) {
      const int ti = timing_begin(c, st, (double)K * N * 2.0 + (double)R * K * 2.0 + (double)chains * R * N * 4.0);
      dim3 grid(((ncolg + 7) / 8) * 8 * zb, chains, 1);
      const u32x4* xp = reinterpret_cast<const u32x4*>(x);
      const u32x4* wp = reinterpret_cast<const u32x4*>(w);
      if (zb == 1) hipLaunchKernelGGL((k_gemm<true, 4, 0, 1>), grid, dim3(256), (size_t)64 * 1024, st, xp, wp, K, N, R, out, ncolg, zb, (float*)nullptr, (int32_t*)nullptr, ColRanges{});
      else hipLaunchKernelGGL((k_gemm<false, 4, 0, 1>), grid, dim3(256), (size_t)64 * 1024, st, xp, wp, K, N, R, out, ncolg, zb, (float*)nullptr, (int32_t*)nullptr, ColRanges{});
      timing_end(c, st, ti);
      LAUNCHCHK();
      res->slices = chains;
      res->slice_stride = (size_t)R * N;
      res->ntiles = ntile;
      return IXT_OK;
    }
  }
  const int stream_min_blocks = c->dot_stream_min_blocks;
  if (!force_full && nrb == 4 && zblocks >= 2 && chains > 1 && chains <= 4 && (long)((ntile + 7) / 8) * zblocks >= stream_min_blocks &&
      (size_t)R * N <= out_floats) {
    const int nw = (ntile + 1) / 2;   // waves: one tile duo each
    const int ti = timing_begin(c, st, (double)K * N * 2.0 + (double)R * K * 2.0 + (double)R * N * 4.0);
    dim3 grid((((nw + 3) / 4 + 7) / 8) * 8 * zblocks, 1, 1);
    hipLaunchKernelGGL((k_unembed<4, false, 2, 0>), grid, dim3(256), (size_t)4 * 16 * 1024, st, reinterpret_cast<const u32x4*>(x),
                       reinterpret_cast<const u32x4*>(w), K, N, R, out, (float*)nullptr, (int32_t*)nullptr, nw, zblocks, ColRanges{});
    timing_end(c, st, ti);
    LAUNCHCHK();
    res->slices = 1;
    res->slice_stride = (size_t)R * N;
    res->ntiles = nw;
    return IXT_OK;
  }
  if (split && (size_t)chains * R * N > out_floats) split = false;
  if (!split && (size_t)R * N > out_floats) return set_err(IXT_ERR_INVALID, "dot: output workspace too small");
  const int nwaves = ntile;
  const bool nt = zblocks == 1;  // weights read once per launch
  const int slices = split ? chains : 1;
  // waves per block: 4 share one LDS copy of the activations; 2 when that would leave CUs without a block
  const int wpb = (split || (long)((nwaves + 3) / 4) * zblocks >= 256) ? 4 : 2;   // (split-K blocks are always 4 waves: measured faster, and the kernel relies on it)
  const double wbytes = (double)K * N * 2.0 + (double)R * K * 2.0;
  const int ti = timing_begin(c, st, wbytes + (double)slices * R * N * 4.0);
  {
    dim3 grid((((nwaves + wpb - 1) / wpb + 7) / 8) * 8 * zblocks, split ? chains : 1, 1);   // tile groups padded to the 8 XCDs
    const dim3 block(64 * wpb);
    const size_t lds = (size_t)nrb * 16 * 1024;
    const u32x4* xp = reinterpret_cast<const u32x4*>(x);
    const u32x4* wp = reinterpret_cast<const u32x4*>(w);
#define DOTB(B_, S_)                                                                                              \
  do {                                                                                                            \
    if (nt) hipLaunchKernelGGL((k_dotb<B_, S_, true>), grid, block, lds, st, xp, wp, K, N, R, out, nwaves, zblocks);   \
    else hipLaunchKernelGGL((k_dotb<B_, S_, false>), grid, block, lds, st, xp, wp, K, N, R, out, nwaves, zblocks);     \
  } while (0)
#define DOTB_B(S_) do { if (nrb == 4) DOTB(4, S_); else if (nrb == 2) DOTB(2, S_); else DOTB(1, S_); } while (0)
    if (split) DOTB_B(true);
    else DOTB_B(false);
#undef DOTB_B
#undef DOTB
  }
  timing_end(c, st, ti);
  LAUNCHCHK();
  res->slices = slices;
  res->slice_stride = (size_t)R * N;
  res->ntiles = nwaves;
  return IXT_OK;
}


// logits (optional store) + greedy argmax for R rows of normed hidden states
static int launch_unembed(ixt_ctx* c, const bf16_t* yT, int R, float* logits_out, int32_t* argmax_out, hipStream_t st) {
  const int K = c->E, N = c->V;
  if (R >= c->ring_min_rows && c->unembed_gemm) {   // many rows: the LDS-ring GEMM with the sampling epilogue
    if ((size_t)R * ((N / 16 + 1) / 2) > c->amax_cap) return set_err(IXT_ERR_INVALID, "argmax workspace too small");   // one pair per row and wave column (>= 2 tiles)
    GemmEpi e{};
    e.out = logits_out; e.amax_val = c->amax_val; e.amax_idx = c->amax_idx; e.cr = c->sample_cols;
    int wave_cols = 0;
    TRY(launch_ring(c, GEPI_UNEMBED, yT, c->logits_w, K, N, R, e, false, st, 0, &wave_cols));
    hipLaunchKernelGGL(k_argmax_final, dim3(R), dim3(256), 0, st, c->amax_val, c->amax_idx, wave_cols, argmax_out);
    LAUNCHCHK();
    return IXT_OK;
  }
  if (R >= c->gemm_min_rows && c->unembed_gemm) {   // many rows: the tiled GEMM with the sampling epilogue (weights pulled through L2 half as often)
    const int ntile_ = N / 16, ncolg = (ntile_ + 7) / 8, zb = (R + 127) / 128;
    if ((size_t)R * 2 * ncolg > c->amax_cap) return set_err(IXT_ERR_INVALID, "argmax workspace too small");
    const int ti = timing_begin(c, st, (double)K * N * 2.0 + (double)R * K * 2.0 + (logits_out ? (double)R * N * 4.0 : 0.0));
    dim3 grid(((ncolg + 7) / 8) * 8 * zb, 1, 1);
    if (zb == 1) hipLaunchKernelGGL((k_gemm<true, 4, 1, 0>), grid, dim3(256), (size_t)64 * 1024, st, reinterpret_cast<const u32x4*>(yT), reinterpret_cast<const u32x4*>(c->logits_w), K, N, R, logits_out, ncolg, zb, c->amax_val, c->amax_idx, c->sample_cols);
    else hipLaunchKernelGGL((k_gemm<false, 4, 1, 0>), grid, dim3(256), (size_t)64 * 1024, st, reinterpret_cast<const u32x4*>(yT), reinterpret_cast<const u32x4*>(c->logits_w), K, N, R, logits_out, ncolg, zb, c->amax_val, c->amax_idx, c->sample_cols);
    timing_end(c, st, ti);
    hipLaunchKernelGGL(k_argmax_final, dim3(R), dim3(256), 0, st, c->amax_val, c->amax_idx, 2 * ncolg, argmax_out);
    LAUNCHCHK();
    return IXT_OK;
  }
  const int ntile = N / 16, nwaves = (ntile + 3) / 4;
  const int rowblocks = (R + 15) / 16;
  const int nrb = rowblocks >= 3 ? 4 : rowblocks;
  const int zblocks = (rowblocks + nrb - 1) / nrb;
  if ((size_t)R * nwaves > c->amax_cap) return set_err(IXT_ERR_INVALID, "argmax workspace too small");
  const int ti = timing_begin(c, st, (double)K * N * 2.0 + (double)R * K * 2.0 + (logits_out ? (double)R * N * 4.0 : 0.0));
  dim3 grid((((nwaves + 3) / 4 + 7) / 8) * 8 * zblocks, 1, 1);   // column groups padded to the 8 XCDs
  const size_t lds = (size_t)nrb * 16 * 1024;
  const u32x4* xp = reinterpret_cast<const u32x4*>(yT);
  const u32x4* wp = reinterpret_cast<const u32x4*>(c->logits_w);
#define UNEMB(B_)                                                                                                                          \
  do {                                                                                                                                     \
    if (zblocks == 1) hipLaunchKernelGGL((k_unembed<B_, true, 4, 1>), grid, dim3(256), lds, st, xp, wp, K, N, R, logits_out, c->amax_val, c->amax_idx, nwaves, zblocks, c->sample_cols); \
    else hipLaunchKernelGGL((k_unembed<B_, false, 4, 1>), grid, dim3(256), lds, st, xp, wp, K, N, R, logits_out, c->amax_val, c->amax_idx, nwaves, zblocks, c->sample_cols);           \
  } while (0)
  if (nrb == 4) UNEMB(4); else if (nrb == 2) UNEMB(2); else UNEMB(1);
#undef UNEMB
  timing_end(c, st, ti);
  hipLaunchKernelGGL(k_argmax_final, dim3(R), dim3(256), 0, st, c->amax_val, c->amax_idx, nwaves, argmax_out);
  LAUNCHCHK();
  return IXT_OK;
}

template <int HD, int GQ>
static void launch_attn(ixt_ctx* c, int layer, int R, int len_hint, const bf16_t* qbuf, bf16_t* out, hipStream_t st) {
  const AttnGeom& g = c->geom;
  const int pages_hint = (len_hint + g.P - 1) / g.P;
  const int ppw = 64 / g.P;
  int nsplit = (pages_hint + 4 * ppw - 1) / (4 * ppw);
  if (nsplit < 1) nsplit = 1;
  ScoreArgs sa;
  sa.qbuf = qbuf; sa.kpages = c->kpages_l[layer];
  sa.row_slot = c->row_slot; sa.row_pos = c->row_pos; sa.row_active = c->row_active; sa.page_map = c->ps.page_map;
  sa.scores = c->scores; sa.g = g;
  hipLaunchKernelGGL((k_attn_scores<HD, GQ>), dim3(R, g.nkv, nsplit), dim3(256), 0, st, sa);
  const int nsb_hint = (pages_hint + IXT_SB - 1) / IXT_SB;
  int nchunk = (nsb_hint + 3) / 4;
  if (nchunk < 1) nchunk = 1;
  PvArgs pa;
  pa.scores = c->scores; pa.vpages = c->vpages_l[layer];
  pa.row_slot = c->row_slot; pa.row_pos = c->row_pos; pa.row_active = c->row_active; pa.page_map = c->ps.page_map;
  pa.pvpart = c->pvpart; pa.g = g; pa.cap = c->pv_cap; pa.window = c->pv_window;
  const size_t lds = sizeof(float) * ((size_t)g.mpg * g.P + g.mpg + 8);
  hipLaunchKernelGGL((k_attn_pv<HD, GQ>), dim3(R, g.nkv, nchunk), dim3(256), lds, st, pa);
  hipLaunchKernelGGL(k_attn_combine, dim3(R), dim3(256), 0, st, c->pvpart, c->row_pos, c->row_active, out, g, 0);
}

static int launch_attention(ixt_ctx* c, int layer, int R, int len_hint, const bf16_t* qbuf, bf16_t* out, hipStream_t st) {
#define ATT(HD_, GQ_) if (c->D == HD_ && c->GQ == GQ_) { launch_attn<HD_, GQ_>(c, layer, R, len_hint, qbuf, out, st); LAUNCHCHK(); return IXT_OK; }
  ATT(128, 1) ATT(128, 2) ATT(128, 4) ATT(64, 1) ATT(64, 2) ATT(64, 4) ATT(256, 1) ATT(256, 2) ATT(256, 4)
#undef ATT
  return set_err(IXT_ERR_UNSUPPORTED, "attention geometry hd=%d gq=%d", c->D, c->GQ);
}

static int launch_rownorm(ixt_ctx* c, int mode, int R, const int32_t* tokens, const DotResult* part, const bf16_t* h_in, bf16_t* h_out,
                          const bf16_t* scale, bf16_t* xn_out, hipStream_t st, int kmajor = 1, const int32_t* row_index = nullptr,
                          const float* part_base = nullptr) {
  RowNormArgs a;
  a.xn_kmajor = kmajor;
  a.row_index = row_index;
  a.tokens = tokens; a.embedding = c->embedding; a.part = part_base ? part_base : c->part;
  a.slices = part ? part->slices : 0; a.slice_stride = part ? part->slice_stride : 0;
  a.h_in = h_in; a.h_out = h_out; a.scale = scale; a.xn_out = xn_out; a.E = c->E; a.eps = c->cfg.normalization_layer_epsilon;
  const size_t lds = sizeof(float) * ((size_t)c->E + 64);
  if (mode == 0) hipLaunchKernelGGL((k_rownorm<0>), dim3(R), dim3(256), lds, st, a);
  else if (mode == 1) hipLaunchKernelGGL((k_rownorm<1>), dim3(R), dim3(256), lds, st, a);
  else hipLaunchKernelGGL((k_rownorm<2>), dim3(R), dim3(256), lds, st, a);
  LAUNCHCHK();
  return IXT_OK;
}

template <int HD, int GQ>
static void launch_front_pv(ixt_ctx* c, int layer, int R, int len_hint, const DotResult& dr, hipStream_t st) {
  const AttnGeom& g = c->geom;
  const int pages_hint = (len_hint + g.P - 1) / g.P;
  const int ppw = 64 / g.P;
  int nsplit = (pages_hint + 4 * ppw - 1) / (4 * ppw);
  if (nsplit < 1) nsplit = 1;
  FrontArgs fa;
  fa.part = c->part; fa.slices = dr.slices; fa.slice_stride = dr.slice_stride; fa.ncols = c->NQKV;
  fa.row_slot = c->row_slot; fa.row_pos = c->row_pos; fa.row_active = c->row_active; fa.page_map = c->ps.page_map;
  fa.rope_sin = c->rope_sin; fa.rope_cos = c->rope_cos;
  fa.kpages = c->kpages_l[layer]; fa.vpages = c->vpages_l[layer];
  fa.scores = c->scores; fa.g = g;
  hipLaunchKernelGGL((k_attn_front_decode<HD, GQ>), dim3(R, g.nkv, nsplit), dim3(256), 0, st, fa);
  const int nsb_hint = (pages_hint + IXT_SB - 1) / IXT_SB;
  int nchunk = (nsb_hint + 3) / 4;
  if (nchunk < 1) nchunk = 1;
  PvArgs pa;
  pa.scores = c->scores; pa.vpages = c->vpages_l[layer];
  pa.row_slot = c->row_slot; pa.row_pos = c->row_pos; pa.row_active = c->row_active; pa.page_map = c->ps.page_map;
  pa.pvpart = c->pvpart; pa.g = g; pa.cap = c->pv_cap; pa.window = c->pv_window;
  const size_t lds = sizeof(float) * ((size_t)g.mpg * g.P + g.mpg + 8);
  hipLaunchKernelGGL((k_attn_pv<HD, GQ>), dim3(R, g.nkv, nchunk), dim3(256), lds, st, pa);
}

template <int HD, int GQ>
static void launch_scores_pv(ixt_ctx* c, int layer, int R, int len_hint, hipStream_t st) {
  const AttnGeom& g = c->geom;
  const int pages_hint = (len_hint + g.P - 1) / g.P;
  const int ppw = 64 / g.P;
  int nsplit = (pages_hint + 4 * ppw - 1) / (4 * ppw);
  if (nsplit < 1) nsplit = 1;
  ScoreArgs sa;
  sa.qbuf = c->qbuf; sa.kpages = c->kpages_l[layer];
  sa.row_slot = c->row_slot; sa.row_pos = c->row_pos; sa.row_active = c->row_active; sa.page_map = c->ps.page_map;
  sa.scores = c->scores; sa.g = g;
  hipLaunchKernelGGL((k_attn_scores<HD, GQ>), dim3(R, g.nkv, nsplit), dim3(256), 0, st, sa);
  const int nsb_hint = (pages_hint + IXT_SB - 1) / IXT_SB;
  int nchunk = (nsb_hint + 3) / 4;
  if (nchunk < 1) nchunk = 1;
  PvArgs pa;
  pa.scores = c->scores; pa.vpages = c->vpages_l[layer];
  pa.row_slot = c->row_slot; pa.row_pos = c->row_pos; pa.row_active = c->row_active; pa.page_map = c->ps.page_map;
  pa.pvpart = c->pvpart; pa.g = g; pa.cap = c->pv_cap; pa.window = c->pv_window;
  const size_t lds = sizeof(float) * ((size_t)g.mpg * g.P + g.mpg + 8);
  hipLaunchKernelGGL((k_attn_pv<HD, GQ>), dim3(R, g.nkv, nchunk), dim3(256), lds, st, pa);
}

template <int HD, int GQ>
static void launch_decode_attn(ixt_ctx* c, int layer, int R, const DotResult& dr, bool newtok, hipStream_t st) {
  const AttnGeom& g = c->geom;
  DecAttnArgs da;
  da.part = c->part; da.slices = dr.slices; da.slice_stride = dr.slice_stride; da.ncols = c->NQKV;
  da.row_slot = c->row_slot; da.row_pos = c->row_pos; da.row_active = c->row_active; da.page_map = c->ps.page_map;
  da.rope_sin = c->rope_sin; da.rope_cos = c->rope_cos;
  da.kpages = c->kpages_l[layer]; da.vpages = c->vpages_l[layer];
  da.out = c->ao; da.g = g; da.qbuf = c->qbuf;
  const size_t lds_w = 4 * sizeof(float) * ((size_t)(GQ + 2) * HD + (size_t)GQ * g.Tmax + g.mpg);
  if (newtok && c->attn_wave_pairs && lds_w <= 64 * 1024) {   // one wave per (row, kv head), no block barriers
    const int npairs = R * g.nkv;
    hipLaunchKernelGGL((k_attn_decode_w<HD, GQ>), dim3((npairs + 3) / 4), dim3(256), lds_w, st, da, npairs);
    return;
  }
  const size_t lds = sizeof(float) * ((size_t)GQ * HD + 2 * HD + (size_t)GQ * ((g.Tmax + 3) & ~3) + ((g.mpg + 3) & ~3) + 8 + (size_t)g.nsb_max * HD);
  if (newtok) hipLaunchKernelGGL((k_attn_decode<HD, GQ, true>), dim3(g.nkv, R), dim3(256), lds, st, da);
  else hipLaunchKernelGGL((k_attn_decode<HD, GQ, false>), dim3(g.nkv, R), dim3(256), lds, st, da);
}

// prefill rows (q in qbuf, K/V already in the pages): NQ query vectors per block share every K / V load.  false: the score rows of
// len_hint keys do not fit the LDS (the caller takes the per-row kernels).
template <int HD, int GQ, int NQ>
static int tile_attn_t(ixt_ctx* c, int layer, int R, int lmax, size_t lds, hipStream_t st, float cap, int window) {
  static bool attr_done = false;
  if (!attr_done) {   // up to 160 KiB of dynamic LDS: the attribute once per kernel
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn_tile<HD, GQ, NQ>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
    attr_done = true;
  }
  TileAttnArgs ta;
  ta.row_slot = c->row_slot; ta.row_pos = c->row_pos; ta.row_active = c->row_active; ta.page_map = c->ps.page_map;
  ta.qbuf = c->qbuf; ta.kpages = c->kpages_l[layer]; ta.vpages = c->vpages_l[layer]; ta.out = c->ao; ta.g = c->geom; ta.R = R; ta.lmax = lmax; ta.cap = cap; ta.window = window;
  constexpr int TQ = NQ / GQ;
  hipLaunchKernelGGL((k_attn_tile<HD, GQ, NQ>), dim3(c->geom.nkv, (R + TQ - 1) / TQ), dim3(256), lds, st, ta);
  LAUNCHCHK();
  return IXT_OK;
}
static size_t tile_attn_lds(const ixt_ctx* c, int nq, int lmax) {
  const int ppb = 4 * (64 / (c->D / 4));   // pages of a PV pass
  return sizeof(float) * ((size_t)nq * lmax + (size_t)nq * ((c->geom.mpg + 3) & ~3) + (size_t)ppb * (nq / 2) * c->D);
}
static bool tile_attn_fits(const ixt_ctx* c, int R, int len_hint, bool decode) {
  if (decode || !c->attn_tile || R < c->attn_tile_min_rows) return false;
  const int lmax = (std::min(len_hint, c->Tmax) + 3) & ~3;
  return tile_attn_lds(c, 8, lmax) <= 150 * 1024;
}
static int launch_attention_tile(ixt_ctx* c, int layer, int R, int len_hint, hipStream_t st, float cap = 0.0f, int window = 0) {
  const int lmax = (std::min(len_hint, c->Tmax) + 3) & ~3;
  const bool wide = tile_attn_lds(c, 16, lmax) <= 150 * 1024 && c->D <= 128;   // (head_dim 256 with 16 vectors: 100+ bytes of scratch per lane)
  const size_t lds = tile_attn_lds(c, wide ? 16 : 8, lmax);
#define ATT(HD_, GQ_)                                                                             \
  if (c->D == HD_ && c->GQ == GQ_) return wide ? tile_attn_t<HD_, GQ_, 16>(c, layer, R, lmax, lds, st, cap, window) : tile_attn_t<HD_, GQ_, 8>(c, layer, R, lmax, lds, st, cap, window);
  ATT(128, 1) ATT(128, 2) ATT(128, 4) ATT(64, 1) ATT(64, 2) ATT(64, 4) ATT(256, 1) ATT(256, 2) ATT(256, 4)
#undef ATT
  return set_err(IXT_ERR_UNSUPPORTED, "tiled attention: head_dim %d with %d query heads per kv head", c->D, c->GQ);
}

// decode rows, enough (row, kv head) pairs to fill the chip: the whole attention in one launch (writes c->ao)
static int launch_attention_fused(ixt_ctx* c, int layer, int R, const DotResult& qkv, bool newtok, hipStream_t st) {
#define ATT(HD_, GQ_)                                                                             \
  if (c->D == HD_ && c->GQ == GQ_) {                                                              \
    launch_decode_attn<HD_, GQ_>(c, layer, R, qkv, newtok, st);                                   \
    LAUNCHCHK();                                                                                  \
    return IXT_OK;                                                                                \
  }
  ATT(128, 1) ATT(128, 2) ATT(128, 4) ATT(64, 1) ATT(64, 2) ATT(64, 4)
#undef ATT
  return set_err(IXT_ERR_UNSUPPORTED, "attention geometry hd=%d gq=%d", c->D, c->GQ);
}

// attention up to the per-super-block partials (c->pvpart); decode rows use the fused front kernel
static int launch_attention_partials(ixt_ctx* c, int layer, int R, int len_hint, const DotResult& qkv, bool decode, hipStream_t st) {
#define ATT(HD_, GQ_)                                                                             \
  if (c->D == HD_ && c->GQ == GQ_) {                                                              \
    if (decode) launch_front_pv<HD_, GQ_>(c, layer, R, len_hint, qkv, st);                        \
    else launch_scores_pv<HD_, GQ_>(c, layer, R, len_hint, st);                                   \
    LAUNCHCHK();                                                                                  \
    return IXT_OK;                                                                                \
  }
  ATT(128, 1) ATT(128, 2) ATT(128, 4) ATT(64, 1) ATT(64, 2) ATT(64, 4) ATT(256, 1) ATT(256, 2) ATT(256, 4)
#undef ATT
  return set_err(IXT_ERR_UNSUPPORTED, "attention geometry hd=%d gq=%d", c->D, c->GQ);
}

// ---- few rows: k_sdot / k_attn_small (ixt_small.h)
struct SdotPlan { int chains, tpb, rpl; size_t lds; bool ok; int pf_blocks; };
static SdotPlan sdot_plan(const ixt_ctx* c, int epi, bool norm, int K, int N, int R) {
  SdotPlan p{};
  p.chains = (K + IXT_CHAIN - 1) / IXT_CHAIN;
  p.rpl = 0;
  while ((1 << p.rpl) < R) ++p.rpl;
  const int ntile = N / 16;
  // tiles per block: the blocks of a norm-fused launch each norm the rows for themselves, two tiles share that (QKV at C2, 8 rows:
  // 11.3 us with two tiles per block against 14.3 with one; the plain forms are fastest with one: profiles/r03_b8_by_grid.txt)
  int tpb = c->small_tpb > 0 ? c->small_tpb : ((epi == GEPI_SWIGLU || norm) ? 2 : 1);
  if (epi == GEPI_SWIGLU && (tpb & 1)) ++tpb;
  while (tpb > 1 && p.chains * tpb > 16) tpb -= (epi == GEPI_SWIGLU ? 2 : 1);
  if (tpb > ntile) tpb = ntile;
  p.tpb = tpb;
  const int nblk = (K / 8 + 63) / 64;
  p.lds = ((size_t)(K / 8) << p.rpl) * 16 + (size_t)p.chains * tpb * 1024 + (norm ? 1024 + (size_t)(K / 8) * 16 : 0);
  p.ok = R >= 1 && R <= 16 && R * nblk <= 240 && K % IXT_MK == 0 && N % 16 == 0 && p.chains * tpb <= 16 && tpb >= 1 && (epi != GEPI_SWIGLU || tpb % 2 == 0) && p.lds <= 150 * 1024;
  return p;
}

template <int EPI, bool NORM>
static int sdot_t(const SdotArgs& a, const SdotPlan& p, hipStream_t st) {
  hipLaunchKernelGGL((k_sdot<EPI, NORM>), dim3(a.main_blocks + (a.pf_n ? p.pf_blocks : 0)), dim3(64 * p.chains * p.tpb), p.lds, st, a);
  LAUNCHCHK();
  return IXT_OK;
}

// up to 156 KiB of dynamic LDS (the down projection keeps R x mlp_dim activations on chip): the attribute once per kernel, at ixt_create
static int sdot_set_attributes() {
  static bool done = false;
  if (done) return IXT_OK;
#define SATTR(E_, N_) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sdot<E_, N_>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024))
  SATTR(GEPI_QKV, true); SATTR(GEPI_QKV, false); SATTR(GEPI_RESID, true); SATTR(GEPI_RESID, false);
  SATTR(GEPI_SWIGLU, true); SATTR(GEPI_SWIGLU, false); SATTR(GEPI_F32, true); SATTR(GEPI_F32, false);
#undef SATTR
  done = true;
  return IXT_OK;
}

// x: NORM: hidden rows [R][K] row-major (+ scale); else activations in xidx layout.  w: widx layout.
static int launch_sdot(ixt_ctx* c, int epi, bool norm, const bf16_t* x, const bf16_t* scale, const bf16_t* w, int K, int N, int R, const GemmEpi& e,
                       hipStream_t st, const bf16_t* pf = nullptr, size_t pf_elems = 0) {
  SdotPlan p = sdot_plan(c, epi, norm, K, N, R);
  if (!p.ok) return set_err(IXT_ERR_UNSUPPORTED, "sdot: %d rows x K %d x N %d does not fit the few-rows form", R, K, N);
  SdotArgs a;
  a.x = x; a.scale = scale; a.eps = c->cfg.normalization_layer_epsilon; a.w = reinterpret_cast<const u32x4*>(w);
  a.K = K; a.N = N; a.R = R; a.rpl = p.rpl; a.tpb = p.tpb; a.chains = p.chains; a.e = e;
  a.main_blocks = (N / 16 + p.tpb - 1) / p.tpb;
  const bool prefetch = c->small_prefetch && pf != nullptr && pf_elems > 0;
  a.pf = prefetch ? reinterpret_cast<const u32x4*>(pf) : nullptr;
  a.pf_n = prefetch ? pf_elems / 8 : 0;
  p.pf_blocks = c->small_prefetch_blocks;
  const int ti = timing_begin(c, st, (double)K * N * 2.0 + (double)R * K * 2.0);
  int rc = IXT_ERR_UNSUPPORTED;
  if (epi == GEPI_QKV) rc = norm ? sdot_t<GEPI_QKV, true>(a, p, st) : sdot_t<GEPI_QKV, false>(a, p, st);
  else if (epi == GEPI_RESID) rc = norm ? sdot_t<GEPI_RESID, true>(a, p, st) : sdot_t<GEPI_RESID, false>(a, p, st);
  else if (epi == GEPI_SWIGLU) rc = norm ? sdot_t<GEPI_SWIGLU, true>(a, p, st) : sdot_t<GEPI_SWIGLU, false>(a, p, st);
  else if (epi == GEPI_F32) rc = norm ? sdot_t<GEPI_F32, true>(a, p, st) : sdot_t<GEPI_F32, false>(a, p, st);
  timing_end(c, st, ti);
  return rc;
}

template <int HD, int GQ>
static void launch_attn_small_t(ixt_ctx* c, int layer, int R, hipStream_t st, const bf16_t* pf, size_t pf_elems) {
  const AttnGeom& g = c->geom;
  DecAttnArgs da{};
  const bool prefetch = c->small_prefetch && pf != nullptr && pf_elems > 0;
  da.rows = R; da.pf = prefetch ? reinterpret_cast<const u32x4*>(pf) : nullptr; da.pf_n = prefetch ? pf_elems / 8 : 0;
  const int pf_rows = prefetch ? (2 * c->small_prefetch_blocks + g.nkv - 1) / g.nkv : 0;   // (the attention leaves most of the chip idle: twice the blocks)
  da.row_slot = c->row_slot; da.row_pos = c->row_pos; da.row_active = c->row_active; da.page_map = c->ps.page_map;
  da.rope_sin = c->rope_sin; da.rope_cos = c->rope_cos;
  da.kpages = c->kpages_l[layer]; da.vpages = c->vpages_l[layer];
  da.out = c->ao; da.g = g; da.qbuf = c->qbuf;
  const size_t lds = sizeof(float) * ((size_t)GQ * HD + (size_t)GQ * ((g.Tmax + 3) & ~3) + 2 * ((g.mpg + 3) & ~3) + 16 + (size_t)g.nsb_max * HD);
  hipLaunchKernelGGL((k_attn_small<HD, GQ, 8>), dim3(g.nkv, R + pf_rows), dim3(512), lds, st, da);
}
static bool attn_small_fits(const ixt_ctx* c, int R) {
  const AttnGeom& g = c->geom;
  const size_t lds = sizeof(float) * ((size_t)c->GQ * c->D + (size_t)c->GQ * ((g.Tmax + 3) & ~3) + 2 * ((g.mpg + 3) & ~3) + 16 + (size_t)g.nsb_max * c->D);
  return c->small_attn_waves >= 8 && (long)R * c->KV <= c->small_attn_max_blocks && lds <= 64 * 1024 && (c->D == 128 || c->D == 64) && (c->GQ == 1 || c->GQ == 2);   // (four query heads per kv head: the buffers spill)
}
static int launch_attn_small(ixt_ctx* c, int layer, int R, hipStream_t st, const bf16_t* pf = nullptr, size_t pf_elems = 0) {
#define ATT(HD_, GQ_) if (c->D == HD_ && c->GQ == GQ_) { launch_attn_small_t<HD_, GQ_>(c, layer, R, st, pf, pf_elems); LAUNCHCHK(); return IXT_OK; }
  ATT(128, 1) ATT(128, 2) ATT(64, 1) ATT(64, 2)
#undef ATT
  return set_err(IXT_ERR_UNSUPPORTED, "attention geometry hd=%d gq=%d", c->D, c->GQ);
}

// Router: one wave per 512-k block over all experts when the expert count allows whole 16-byte gate loads and E fits the block table.
static void launch_moe_route(ixt_ctx* c, int R, const bf16_t* xn, const bf16_t* gate, int32_t* idx, float* wts, hipStream_t st) {
  const int nblk = (c->E / 8 + 63) / 64;
  if (c->moe_route_blocks && nblk <= 64 && (c->NE == 8 || c->NE == 16)) {
    const dim3 blk(64 * std::min(nblk, 16));
    if (c->NE == 8) hipLaunchKernelGGL((k_moe_route_blocks<1>), dim3(R), blk, 0, st, xn, gate, c->E, c->TK, idx, wts);
    else hipLaunchKernelGGL((k_moe_route_blocks<2>), dim3(R), blk, 0, st, xn, gate, c->E, c->TK, idx, wts);
    return;
  }
  hipLaunchKernelGGL(k_moe_route, dim3(R), dim3(64 * std::min(c->NE, 16)), 0, st, xn, gate, c->E, c->NE, c->TK, idx, wts);
}

// k_attn_split: S blocks per (row, kv head).  Every block of the launch must be resident at once (the waits need it): no more blocks than
// the occupancy of THIS kernel instantiation with its LDS says fit the chip, and at most two per CU.
template <int HD, int GQ>
static int attn_split_capacity_t(size_t lds) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_attn_split<HD, GQ>, 256, lds) != hipSuccess) return 0;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  return std::min(per_cu, 2) * cus;
}
static int attn_split_capacity(ixt_ctx* c, size_t lds) {
  if (c->split_capacity >= 0) return c->split_capacity;
  int cap = 0;
#define ATT(HD_, GQ_) if (c->D == HD_ && c->GQ == GQ_) cap = attn_split_capacity_t<HD_, GQ_>(lds);
  ATT(128, 1) ATT(128, 2) ATT(128, 4) ATT(64, 1) ATT(64, 2) ATT(64, 4) ATT(256, 1) ATT(256, 2) ATT(256, 4)
#undef ATT
  c->split_capacity = cap;
  return cap;
}
static int attn_split_factor(ixt_ctx* c, int R) {
  const AttnGeom& g = c->geom;
  if (c->small_attn_split < 2 || R > 16 || g.P > 32 || !(c->D == 256 || c->D == 128 || c->D == 64) || !(c->GQ == 1 || c->GQ == 2 || c->GQ == 4)) return 1;
  const size_t lds = sizeof(float) * ((size_t)c->GQ * c->D + (size_t)c->GQ * ((g.Tmax + 3) & ~3) + ((g.mpg + 3) & ~3) + 32 + (size_t)c->GQ * 4 * c->D + (size_t)c->GQ * ((g.mpg + 3) & ~3));
  if (lds > 64 * 1024) return 1;
  // Where k_attn_small applies (plain scores, <= 2 query heads per kv head, head_dim <= 128) the two are level at the benchmark's model -- one
  // utterance 1.069 against 1.046 ms per step, eight 1.318 against 1.229: the three exchanges cost ~3.5 us each, what the split saves in loads --,
  // so the split form serves the geometries that one-block form does not hold: four query heads per kv head, head_dim 256, soft cap, window.
  if (!c->gemma && c->small_attn_split < 100 && attn_small_fits(c, R)) return 1;
  int S = attn_split_capacity(c, lds) / (R * c->KV);
  const int smax = c->small_attn_split >= 100 ? c->small_attn_split - 100 : c->small_attn_split;   // 100 + S: also where k_attn_small would do (tests)
  if (S > smax) S = smax;
  if (S > 16) S = 16;
  return S < 2 ? 1 : S;
}
template <int HD, int GQ>
static void launch_attn_split_t(ixt_ctx* c, int layer, int R, int S, hipStream_t st, float cap, int window) {
  const AttnGeom& g = c->geom;
  DecAttnArgs da{};
  da.rows = R;
  da.row_slot = c->row_slot; da.row_pos = c->row_pos; da.row_active = c->row_active; da.page_map = c->ps.page_map;
  da.kpages = c->kpages_l[layer]; da.vpages = c->vpages_l[layer];
  da.out = c->ao; da.g = g; da.qbuf = c->qbuf;
  SplitWs w{};
  w.cnt = c->split_cnt; w.ws = c->split_ws; w.stride = c->split_stride; w.err = c->split_cnt + (size_t)16 * c->KV * 4; w.mpgp = (g.mpg + 3) & ~3;
  w.cap = cap; w.window = window;
  const size_t lds = sizeof(float) * ((size_t)GQ * HD + (size_t)GQ * ((g.Tmax + 3) & ~3) + w.mpgp + 32 + (size_t)GQ * 4 * HD + (size_t)GQ * w.mpgp);
  hipLaunchKernelGGL((k_attn_split<HD, GQ>), dim3(g.nkv, R, S), dim3(256), lds, st, da, w);
}
static int launch_attn_split(ixt_ctx* c, int layer, int R, int S, hipStream_t st, float cap = 0.0f, int window = 0) {
#define ATT(HD_, GQ_) if (c->D == HD_ && c->GQ == GQ_) { launch_attn_split_t<HD_, GQ_>(c, layer, R, S, st, cap, window); LAUNCHCHK(); return IXT_OK; }
  ATT(128, 1) ATT(128, 2) ATT(128, 4) ATT(64, 1) ATT(64, 2) ATT(64, 4) ATT(256, 1) ATT(256, 2) ATT(256, 4)
#undef ATT
  return set_err(IXT_ERR_UNSUPPORTED, "attention geometry hd=%d gq=%d", c->D, c->GQ);
}

// decoders.py:552-565 on nl rows of f32 logits in place (/ sqrt(emb), soft cap), then the greedy choice over the allowed columns of the
// post-processed values (the cap is not strictly monotone after rounding, so the maximum is taken after it).
static int launch_logits_post(ixt_ctx* c, float* lg, int nl, int normalize, float cap, int32_t* argmax_out, hipStream_t st) {
  const int nblk = (c->V + IXT_LPOST_COLS - 1) / IXT_LPOST_COLS;
  if ((size_t)nl * nblk > c->amax_cap) return set_err(IXT_ERR_INVALID, "argmax workspace too small");
  hipLaunchKernelGGL(k_logits_post, dim3(nblk, nl), dim3(256), 0, st, lg, c->V, normalize, sqrtf((float)c->E), cap, c->sample_cols, c->amax_val, c->amax_idx);
  LAUNCHCHK();
  if (argmax_out != nullptr) {
    hipLaunchKernelGGL(k_argmax_final, dim3(nl), dim3(256), 0, st, c->amax_val, c->amax_idx, nblk, argmax_out);
    LAUNCHCHK();
  }
  return IXT_OK;
}

// The Gemma-2 block over R rows (layers/gemma2.py:69-236): per (attention, MLP) sub-layer
//   QKV dot -> k_qkv_finalize (RoPE, q -> qbuf, K/V -> pages) -> scores / softmax + PV (soft cap; sliding window on the even, "local"
//   sub-layers) / combine -> O dot -> k_gnorm (post norm, residual, pre-MLP norm) -> wi dot -> GELU gate -> down dot -> k_gnorm (post norm,
//   residual, the next pre-attention norm or the final norm).  Prefill and decode rows take the same launches: every row's K/V is in its
//   page before any score is computed, and a row never sees a key beyond its own position.
// Logits: unembedding (dense kernel, or the transposed embedding when logits_via_embedding), then k_logits_post (/ sqrt(emb), soft cap,
// greedy sampling over the allowed columns -- the cap is not strictly monotone after rounding, so the maximum is taken after it).
static int forward_rows_gemma2(ixt_ctx* c, int R, const int32_t* tokens, int len_hint, int lrow0, int nl, const int32_t* lrows, int32_t* argmax_out,
                               float* logits_out, hipStream_t st) {
  const int E = c->E, F = c->F;
  const ixt_config& cf = c->cfg;
  DotResult dr;
  bf16_t* hcur = c->h_a;
  bf16_t* hnext = c->h_b;
  const bool all_rows = nl == R && lrow0 == 0 && lrows == nullptr;
  const size_t glds = sizeof(float) * ((size_t)E + 128);
  const bool attn_tiled = c->gemma_tile && tile_attn_fits(c, R, len_hint, false) && attn_split_factor(c, R) <= 1;
  TRY(launch_rownorm(c, 0, R, tokens, nullptr, nullptr, hcur, c->layers[0].ln1, c->xn, st));
  for (int l = 0; l < c->L; ++l) {
    const LayerW& w = c->layers[l];
    TRY(launch_dot(c, c->xn, w.wqkv, E, c->NQKV, R, c->part, c->part_floats, false, st, &dr));
    QkvArgs qa;
    qa.part = c->part; qa.slices = dr.slices; qa.slice_stride = dr.slice_stride; qa.ncols = c->NQKV;
    qa.row_slot = c->row_slot; qa.row_pos = c->row_pos; qa.row_active = c->row_active; qa.page_map = c->ps.page_map;
    // gemma2.py: even sub-layers are the local ones; gemma3.py:35-47: five local layers, then a global one
    const bool local = c->gemma3 ? (l % 6 != 5) : (l % 2 == 0);
    const bool lrope = local && c->rope_sin_local != nullptr;
    qa.rope_sin = lrope ? c->rope_sin_local : c->rope_sin; qa.rope_cos = lrope ? c->rope_cos_local : c->rope_cos; qa.qbuf = c->qbuf;
    qa.kpages = c->kpages_l[l]; qa.vpages = c->vpages_l[l];
    qa.g = c->geom;
    qa.qnorm = w.qn; qa.knorm = w.kn; qa.eps = cf.normalization_layer_epsilon;
    qa.q_scalar = (cf.query_pre_attn_scalar != 0.0f && cf.query_pre_attn_scalar != 1.0f) ? host_rbf(cf.query_pre_attn_scalar) : 0.0f;
    hipLaunchKernelGGL(k_qkv_finalize, dim3(R, c->Q + 2 * c->KV), dim3(64), 0, st, qa);
    LAUNCHCHK();
    const int attn_split = attn_split_factor(c, R);   // few rows (decode): the keys of a head over several blocks of ONE launch instead of scores / PV / combine
    if (attn_split > 1) {
      TRY(launch_attn_split(c, l, R, attn_split, st, cf.attn_logits_soft_cap, local ? cf.sliding_window_size : 0));
    } else if (attn_tiled) {   // prefill rows: consecutive rows share their page group's K and V loads (k_attn_tile with the cap and the window)
      TRY(launch_attention_tile(c, l, R, len_hint, st, cf.attn_logits_soft_cap, local ? cf.sliding_window_size : 0));
    } else {
      c->pv_cap = cf.attn_logits_soft_cap;
      c->pv_window = local ? cf.sliding_window_size : 0;
      int rc = launch_attention_partials(c, l, R, len_hint, DotResult{0, 0, 0}, false, st);
      c->pv_cap = 0.0f; c->pv_window = 0;
      TRY(rc);
      hipLaunchKernelGGL(k_attn_combine, dim3(R), dim3(256), 0, st, c->pvpart, c->row_pos, c->row_active, c->ao, c->geom, 1);
      LAUNCHCHK();
    }
    TRY(launch_dot(c, c->ao, w.wo, c->QN, E, R, c->part, c->part_floats, false, st, &dr));
    GNormArgs ga;
    ga.part = c->part; ga.slices = dr.slices; ga.slice_stride = dr.slice_stride; ga.h_in = hcur; ga.h_out = hnext;
    ga.post = w.pan; ga.next = w.ln2; ga.xn_out = c->xn; ga.E = E; ga.eps = cf.normalization_layer_epsilon;
    hipLaunchKernelGGL(k_gnorm, dim3(R), dim3(256), glds, st, ga);
    LAUNCHCHK();
    TRY(launch_dot(c, c->xn, w.wi, E, 2 * F, R, c->part, c->part_floats, false, st, &dr));
    hipLaunchKernelGGL(k_geglu_part, dim3((F + 2047) / 2048, R), dim3(256), 0, st, c->part, dr.slices, dr.slice_stride, F, c->hact);
    LAUNCHCHK();
    TRY(launch_dot(c, c->hact, w.wout, F, E, R, c->part, c->part_floats, false, st, &dr));
    ga.part = c->part; ga.slices = dr.slices; ga.slice_stride = dr.slice_stride; ga.h_in = hnext; ga.h_out = hcur;
    ga.post = w.pfn; ga.next = l + 1 < c->L ? c->layers[l + 1].ln1 : (all_rows ? c->final_norm : nullptr);
    hipLaunchKernelGGL(k_gnorm, dim3(R), dim3(256), glds, st, ga);
    LAUNCHCHK();
  }
  if (nl > 0) {
    if (nl > c->NLmax) return set_err(IXT_ERR_INVALID, "too many logits rows (%d > %d)", nl, c->NLmax);
    if (!all_rows) TRY(launch_rownorm(c, 2, nl, nullptr, nullptr, lrows ? hcur : hcur + (size_t)lrow0 * E, nullptr, c->final_norm, c->xn, st, 1, lrows));
    float* lg = logits_out ? logits_out : c->glogits;
    TRY(launch_unembed(c, c->xn, nl, lg, argmax_out, st));
    // decoders.py:552-565: the division and the cap belong to the tied head
    const int normalize = cf.logits_via_embedding && cf.normalize_embedding_logits;
    const float cap = cf.logits_via_embedding ? cf.final_logits_soft_cap : 0.0f;
    if (normalize || cap > 0.0f) TRY(launch_logits_post(c, lg, nl, normalize, cap, argmax_out, st));
  }
  c->h_final = hcur; c->h_final_rows = R;
  return IXT_OK;
}

// The model over R rows described by row_slot/row_pos/row_active (device) and `tokens`.
// Logits: nl rows -- rows [lrow0, lrow0 + nl) when lrows == nullptr, else the rows listed in lrows (device, nl entries).
// decode: one new token per row (rows never read each other's new K).
//
// Two launch sequences per layer, same arithmetic:
//   R >= ring_min_rows  LDS-ring GEMMs with fused epilogues (ixt_gemm.h): QKV (+ RoPE, q -> qbuf, K/V -> pages), attention from qbuf
//                       and the pages, O-proj (+ residual), norm, MLP-up (+ SwiGLU), MLP-down (+ residual), norm: 7 launches, no f32
//                       slice leaves the chip;
//   fewer rows          split-K dots into f32 slices: dot_qkv, [front | qkv_finalize + scores], pv, combine, dot_o, rownorm (slice sum
//                       + residual + norm), dot_wi, swiglu, dot_wo, rownorm (slice sum + residual + next norm).
// Routed evaluation of a layer's experts (moe.py:785-938; megablox gmm.py:305): copy p = row p / top_k with expert idx[p] (expert ids relative to
// `w`'s first resident expert).  x: rows in the dot kernels' x layout.  On return the down-projection of copy p is the sum over `res->slices`
// slices of c->part row c->moe_pos[p] (rows of E floats, slice stride res->slice_stride).  An expert nobody chose is never read.
//   at most 64 copies   split-K dots over 16-row blocks (decode: weight streaming), SwiGLU from the slices
//   more                LDS-ring GEMMs over 128-row groups with the SwiGLU epilogue / one f32 slice
static int moe_experts_routed(ixt_ctx* c, const LayerW& w, const bf16_t* x, const int32_t* idx, int npairs, int top_k, hipStream_t st, DotResult* res) {
  const int E = c->E, F = c->F;
  // copies per launch: few (decode) -> 16-row expert ranges, split-K / whole-K dots; a prompt's worth (up to moe_mid_pairs) -> 64-row ranges, four row
  // blocks per weight fragment (whole-K dots, the gate as epilogue); more -> 128-row ranges for the ring GEMM
  const int align = npairs <= 64 ? 16 : ((!c->fp8 && npairs <= c->moe_mid_pairs) ? 64 : 128);
  const int rows_g = (npairs + align - 1) / align * align + align * c->NEL;
  if (rows_g > c->moe_rows_cap) return set_err(IXT_ERR_INVALID, "moe: %d copies exceed the routed workspaces (%d rows)", npairs, c->moe_rows_cap);
  hipLaunchKernelGGL(k_moe_plan, dim3(1), dim3(1024), 0, st, idx, npairs, c->E0, c->NEL, align, rows_g / 16, c->moe_pos, c->moe_rbexp, c->moe_cnt);
  hipLaunchKernelGGL(k_moe_gather, dim3((E + 2047) / 2048, npairs), dim3(256), 0, st, x, c->moe_pos, top_k, E, c->moe_xg);
  LAUNCHCHK();
  const size_t wi_stride = c->moe_wi_stride / 8, wo_stride = c->moe_wo_stride / 8;   // 16-byte units
  if (c->fp8) {   // IXT-FP8-v1: the experts' e4m3 twins (contiguous per layer), activations quantised per copy row and 512-k block
    const ixt_ctx::F8W &ti = c->f8w.at(w.ewi[0]), &to = c->f8w.at(w.ewout[0]);
    auto dot8g = [&](const bf16_t* xin, const ixt_ctx::F8W& t, int K, int N, size_t sstride) -> int {
      const int chains = (K + IXT_CHAIN - 1) / IXT_CHAIN;
      hipLaunchKernelGGL(k_f8_quant_x, dim3(chains, rows_g / 16), dim3(64), 0, st, xin, K, rows_g, c->x8, c->xs8);
      const size_t w8stride = (size_t)K * N / 16;              // bytes / 16
      if (align == 16) {
        if ((size_t)chains * rows_g * N > c->part_floats) return set_err(IXT_ERR_INVALID, "moe: slice workspace too small");
        const int nwaves = N / 16, zblocks = rows_g / 16;
        const int ti_ = timing_begin(c, st, (double)K * N * c->NEL + (double)rows_g * K + (double)chains * rows_g * N * 4.0);
        dim3 grid((((nwaves + 3) / 4 + 7) / 8) * 8 * zblocks, chains, 1);
        hipLaunchKernelGGL((k_dotb8<1>), grid, dim3(256), (size_t)8192, st, reinterpret_cast<const u32x4*>(c->x8), reinterpret_cast<const u32x4*>(t.w8), c->xs8, t.s, K, N,
                           rows_g, c->part, nwaves, zblocks, c->moe_rbexp, w8stride, sstride);
        timing_end(c, st, ti_);
        LAUNCHCHK();
        return chains;
      }
      if ((size_t)rows_g * N > c->part_floats) return set_err(IXT_ERR_INVALID, "moe: output workspace too small");
      GemmEpi e{};
      e.out = c->part; e.xscale = c->xs8; e.wscale = t.s; e.sstride = sstride; e.rb_exp = c->moe_rbexp; e.wstride = w8stride;
      const int rc = ring_launch8_t<2, 4, 4, 2, 2, 4>(c->x8, t.w8, K, N, rows_g, e, st);
      return rc == IXT_OK ? 1 : rc;
    };
    int ch = dot8g(c->moe_xg, ti, E, 2 * F, (size_t)2 * F);
    if (ch < 0) return ch;
    hipLaunchKernelGGL(k_swiglu_part, dim3((F + 2047) / 2048, rows_g), dim3(256), 0, st, c->part, ch, (size_t)rows_g * 2 * F, F, c->moe_hact, c->moe_rbexp);
    LAUNCHCHK();
    ch = dot8g(c->moe_hact, to, F, E, (size_t)E);
    if (ch < 0) return ch;
    res->slices = ch; res->slice_stride = (size_t)rows_g * E; res->ntiles = E / 16;
    return IXT_OK;
  }
  if (align == 64) {
    const int zblocks = rows_g / 64;
    auto whole = [&](auto kern, const bf16_t* xin, const bf16_t* wbase, size_t wstride, int K, int N, float* out, double out_bytes) {
      const int nwaves = N / 16;
      const int ti = timing_begin(c, st, (double)K * N * 2.0 * c->NEL + (double)rows_g * K * 2.0 + out_bytes);
      dim3 grid((((nwaves + 3) / 4 + 7) / 8) * 8 * zblocks, 1, 1);
      hipLaunchKernelGGL(kern, grid, dim3(256), (size_t)64 * 1024, st, reinterpret_cast<const u32x4*>(xin), reinterpret_cast<const u32x4*>(wbase), K, N, rows_g, out, nwaves, zblocks,
                         c->moe_rbexp, wstride);
      timing_end(c, st, ti);
    };
    if ((size_t)rows_g * E > c->part_floats) return set_err(IXT_ERR_INVALID, "moe: output workspace too small");
    whole(k_dotb<4, false, true, 1>, c->moe_xg, w.ewi[0], wi_stride, E, 2 * F, reinterpret_cast<float*>(c->moe_hact), (double)rows_g * F * 2.0);
    LAUNCHCHK();
    whole(k_dotb<4, false, true, 0>, c->moe_hact, w.ewout[0], wo_stride, F, E, c->part, (double)rows_g * E * 4.0);
    LAUNCHCHK();
    res->slices = 1; res->slice_stride = (size_t)rows_g * E; res->ntiles = E / 16;
    return IXT_OK;
  }
  if (align == 16) {
    const int zblocks = rows_g / 16;
    auto dotg = [&](const bf16_t* xin, const bf16_t* wbase, size_t wstride, int K, int N) -> int {
      const int chains = (K + IXT_CHAIN - 1) / IXT_CHAIN, nwaves = N / 16;
      if ((size_t)chains * rows_g * N > c->part_floats) return set_err(IXT_ERR_INVALID, "moe: slice workspace too small");
      const int ti = timing_begin(c, st, (double)K * N * 2.0 * c->NEL + (double)rows_g * K * 2.0 + (double)chains * rows_g * N * 4.0);
      dim3 grid((((nwaves + 3) / 4 + 7) / 8) * 8 * zblocks, chains, 1);
      hipLaunchKernelGGL((k_dotb<1, true, true>), grid, dim3(256), (size_t)16 * 1024, st, reinterpret_cast<const u32x4*>(xin), reinterpret_cast<const u32x4*>(wbase), K, N,
                         rows_g, c->part, nwaves, zblocks, c->moe_rbexp, wstride);
      timing_end(c, st, ti);
      LAUNCHCHK();
      return chains;
    };
    int ch;
    if (c->moe_fused_gate) {   // whole-K blocks with the gate as their epilogue: no f32 slices of the 2 F columns leave the chip
      const int nwaves = 2 * F / 16;
      const int ti = timing_begin(c, st, (double)E * 2 * F * 2.0 * c->NEL + (double)rows_g * E * 2.0 + (double)rows_g * F * 2.0);
      dim3 grid((((nwaves + 3) / 4 + 7) / 8) * 8 * zblocks, 1, 1);
      hipLaunchKernelGGL((k_dotb<1, false, true, 1>), grid, dim3(256), (size_t)16 * 1024, st, reinterpret_cast<const u32x4*>(c->moe_xg), reinterpret_cast<const u32x4*>(w.ewi[0]), E, 2 * F,
                         rows_g, reinterpret_cast<float*>(c->moe_hact), nwaves, zblocks, c->moe_rbexp, wi_stride);
      timing_end(c, st, ti);
      LAUNCHCHK();
    } else {
      ch = dotg(c->moe_xg, w.ewi[0], wi_stride, E, 2 * F);
      if (ch < 0) return ch;
      hipLaunchKernelGGL(k_swiglu_part, dim3((F + 2047) / 2048, rows_g), dim3(256), 0, st, c->part, ch, (size_t)rows_g * 2 * F, F, c->moe_hact, c->moe_rbexp);
      LAUNCHCHK();
    }
    ch = dotg(c->moe_hact, w.ewout[0], wo_stride, F, E);
    if (ch < 0) return ch;
    res->slices = ch; res->slice_stride = (size_t)rows_g * E; res->ntiles = E / 16;
    return IXT_OK;
  }
  if ((size_t)rows_g * E > c->part_floats) return set_err(IXT_ERR_INVALID, "moe: output workspace too small");
  GemmEpi eu{};
  eu.act = c->moe_hact; eu.rb_exp = c->moe_rbexp; eu.wstride = wi_stride;
  TRY(launch_ring(c, GEPI_SWIGLU, c->moe_xg, w.ewi[0], E, 2 * F, rows_g, eu, false, st));
  GemmEpi ed{};
  ed.out = c->part; ed.rb_exp = c->moe_rbexp; ed.wstride = wo_stride;
  TRY(launch_ring(c, GEPI_F32, c->moe_hact, w.ewout[0], F, E, rows_g, ed, false, st));
  res->slices = 1; res->slice_stride = (size_t)rows_g * E; res->ntiles = E / 16;
  return IXT_OK;
}

// Few rows (R <= 16, llama2 block): five launches per layer, nothing but K / V pages, q, the attention output and the hidden rows leaves the chip
// between them (ixt_small.h).  Same results as the other forms bit for bit (tests/test_gpu_small.py compares the two paths of one engine).
static bool small_path_ok(const ixt_ctx* c, int R) {
  if (c->small_rows <= 0 || R > c->small_rows || R > 16 || c->moe || c->gemma || c->fp8) return false;
  return sdot_plan(c, GEPI_QKV, true, c->E, c->NQKV, R).ok && sdot_plan(c, GEPI_RESID, false, c->QN, c->E, R).ok &&
         sdot_plan(c, GEPI_SWIGLU, true, c->E, 2 * c->F, R).ok;
}
static int forward_rows_small(ixt_ctx* c, int R, const int32_t* tokens, int len_hint, int lrow0, int nl, const int32_t* lrows, int32_t* argmax_out,
                              float* logits_out, bool decode, hipStream_t st) {
  const int E = c->E, F = c->F;
  bf16_t* hcur = c->h_a;
  bf16_t* hnext = c->h_b;
  const bool down_small = sdot_plan(c, GEPI_RESID, false, F, E, R).ok;
  const bool attn_small = attn_small_fits(c, R);
  const int attn_split = decode ? attn_split_factor(c, R) : 1;
  const size_t fused_lds = sizeof(float) * ((size_t)c->GQ * c->D + 2 * c->D + (size_t)c->GQ * (c->geom.Tmax + 3) + c->geom.mpg + 12 + (size_t)c->geom.nsb_max * c->D);
  TRY(launch_rownorm(c, 0, R, tokens, nullptr, nullptr, hcur, nullptr, nullptr, st));   // hidden rows = embedding rows (no norm: the QKV kernel norms)
  for (int l = 0; l < c->L; ++l) {
    const LayerW& w = c->layers[l];
    GemmEpi e{};
    e.row_slot = c->row_slot; e.row_pos = c->row_pos; e.row_active = c->row_active; e.page_map = c->ps.page_map;
    e.rope_sin = c->rope_sin; e.rope_cos = c->rope_cos; e.qbuf = c->qbuf; e.kpages = c->kpages_l[l]; e.vpages = c->vpages_l[l]; e.g = c->geom;
    const bf16_t* next_qkv = l + 1 < c->L ? c->layers[l + 1].wqkv : nullptr;
    if (c->small_norm) {
      TRY(launch_sdot(c, GEPI_QKV, true, hcur, w.ln1, w.wqkv, E, c->NQKV, R, e, st, w.wo, (size_t)c->QN * E));   // while QKV runs: the O kernel
    } else {
      TRY(launch_rownorm(c, 2, R, nullptr, nullptr, hcur, nullptr, w.ln1, c->xn, st));
      TRY(launch_sdot(c, GEPI_QKV, false, c->xn, nullptr, w.wqkv, E, c->NQKV, R, e, st));
    }
    {
      const int ta = decode ? timing_begin(c, st, 0.0, 1) : -1;
      if (attn_split > 1) {
        TRY(launch_attn_split(c, l, R, attn_split, st));
      } else if (attn_small) {
        TRY(launch_attn_small(c, l, R, st, w.wi, (size_t)E * 2 * F));                                             // while the attention runs: the wi kernel
      } else if (fused_lds <= 64 * 1024) {
        TRY(launch_attention_fused(c, l, R, DotResult{0, 0, 0}, false, st));
      } else {
        TRY(launch_attention_partials(c, l, R, len_hint, DotResult{0, 0, 0}, false, st));
        hipLaunchKernelGGL(k_attn_combine, dim3(R), dim3(256), 0, st, c->pvpart, c->row_pos, c->row_active, c->ao, c->geom, 1);
        LAUNCHCHK();
      }
      timing_end(c, st, ta);
    }
    GemmEpi eo{};
    eo.h_in = hcur; eo.h_out = hnext;
    TRY(launch_sdot(c, GEPI_RESID, false, c->ao, nullptr, w.wo, c->QN, E, R, eo, st));
    GemmEpi eu{};
    eu.act = c->hact;
    if (c->small_norm) {
      TRY(launch_sdot(c, GEPI_SWIGLU, true, hnext, w.ln2, w.wi, E, 2 * F, R, eu, st, w.wout, (size_t)F * E));      // while wi runs: the wo kernel
    } else {
      TRY(launch_rownorm(c, 2, R, nullptr, nullptr, hnext, nullptr, w.ln2, c->xn, st));
      TRY(launch_sdot(c, GEPI_SWIGLU, false, c->xn, nullptr, w.wi, E, 2 * F, R, eu, st));
    }
    if (down_small) {
      GemmEpi ed{};
      ed.h_in = hnext; ed.h_out = hcur;
      TRY(launch_sdot(c, GEPI_RESID, false, c->hact, nullptr, w.wout, F, E, R, ed, st, next_qkv, (size_t)E * c->NQKV));   // while wo runs: the next layer's QKV kernel
    } else {   // R x mlp_dim activations beyond the LDS: split-K slices + slice-sum / residual
      DotResult dr;
      TRY(launch_dot(c, c->hact, w.wout, F, E, R, c->part, c->part_floats, false, st, &dr, true));
      hipLaunchKernelGGL(k_sumres, dim3((E + 255) / 256, R), dim3(128), 0, st, c->part, dr.slices, dr.slice_stride, hnext, hcur, E);
      LAUNCHCHK();
    }
  }
  if (nl > 0) {
    if (nl > c->NLmax) return set_err(IXT_ERR_INVALID, "too many logits rows (%d > %d)", nl, c->NLmax);
    TRY(launch_rownorm(c, 2, nl, nullptr, nullptr, lrows ? hcur : hcur + (size_t)lrow0 * E, nullptr, c->final_norm, c->xn, st, 1, lrows));
    TRY(launch_unembed(c, c->xn, nl, logits_out, argmax_out, st));
  }
  c->h_final = hcur; c->h_final_rows = R;
  return IXT_OK;
}

static int forward_rows(ixt_ctx* c, int R, const int32_t* tokens, int len_hint, int lrow0, int nl, const int32_t* lrows, int32_t* argmax_out,
                        float* logits_out, bool decode, hipStream_t st) {
  if (!ixt_weights_complete(c)) return set_err(IXT_ERR_NOT_LOADED, "weights incomplete");
  TRY(fp8_prepare(c, st));
  if (c->gemma) return forward_rows_gemma2(c, R, tokens, len_hint, lrow0, nl, lrows, argmax_out, logits_out, st);
  if (c->moe && c->NEL != c->NE) return set_err(IXT_ERR_UNSUPPORTED, "this context holds experts %d..%d of %d: the fused engine needs every expert (expert-parallel contexts serve ixt_op_moe_route / ixt_op_moe_experts)", c->E0, c->E0 + c->NEL - 1, c->NE);
  if (small_path_ok(c, R)) return forward_rows_small(c, R, tokens, len_hint, lrow0, nl, lrows, argmax_out, logits_out, decode, st);
  const int E = c->E, F = c->F;
  DotResult dr;
  bf16_t* hcur = c->h_a;
  bf16_t* hnext = c->h_b;
  const bool final_fused = nl == R && lrow0 == 0 && lrows == nullptr;   // decode: every row is unembedded
  const bool ring = R >= c->ring_min_rows && !c->moe && !c->fp8;   // fp8: the dots write scaled f32 slices, the bf16 path's row kernels take it from there   // Mixtral: the split-K dots (every expert is a launch sequence of its own)
  const int fused_min = c->attn_fused_min_blocks;
  const size_t fused_lds = sizeof(float) * ((size_t)c->GQ * c->D + 2 * c->D + (size_t)c->GQ * (c->geom.Tmax + 3) + c->geom.mpg + 12 + (size_t)c->geom.nsb_max * c->D);
  const bool attn_one_launch = R * c->KV >= fused_min && fused_lds <= 64 * 1024;
  const bool attn_tiled = tile_attn_fits(c, R, len_hint, decode);   // prefill rows: consecutive rows share their page group's K and V loads
  TRY(launch_rownorm(c, 0, R, tokens, nullptr, nullptr, hcur, c->layers[0].ln1, c->xn, st));
  for (int l = 0; l < c->L; ++l) {
    const LayerW& w = c->layers[l];
    const bf16_t* next_scale = l + 1 < c->L ? c->layers[l + 1].ln1 : (final_fused ? c->final_norm : nullptr);
    if (ring) {
      GemmEpi e{};
      e.row_slot = c->row_slot; e.row_pos = c->row_pos; e.row_active = c->row_active; e.page_map = c->ps.page_map;
      e.rope_sin = c->rope_sin; e.rope_cos = c->rope_cos; e.qbuf = c->qbuf; e.kpages = c->kpages_l[l]; e.vpages = c->vpages_l[l]; e.g = c->geom;
      TRY(launch_ring(c, GEPI_QKV, c->xn, w.wqkv, E, c->NQKV, R, e, false, st));
      if (attn_tiled) {
        TRY(launch_attention_tile(c, l, R, len_hint, st));
      } else if (attn_one_launch) {
        const DotResult none{0, 0, 0};
        const int ta = decode ? timing_begin(c, st, 0.0, 1) : -1;
        if (decode && attn_small_fits(c, R)) TRY(launch_attn_small(c, l, R, st));   // (small_attn_max_blocks raised: the 8-wave form with everything in flight)
        else TRY(launch_attention_fused(c, l, R, none, false, st));   // q from qbuf, every key / value (the new ones too) from the pages
        timing_end(c, st, ta);
      } else {
        TRY(launch_attention_partials(c, l, R, len_hint, DotResult{0, 0, 0}, false, st));
        hipLaunchKernelGGL(k_attn_combine, dim3(R), dim3(256), 0, st, c->pvpart, c->row_pos, c->row_active, c->ao, c->geom, 1);
        LAUNCHCHK();
      }
      GemmEpi eo{};
      eo.h_in = hcur; eo.h_out = hnext;
      TRY(launch_ring(c, GEPI_RESID, c->ao, w.wo, c->QN, E, R, eo, false, st));
      TRY(launch_rownorm(c, 2, R, nullptr, nullptr, hnext, nullptr, w.ln2, c->xn, st));
      GemmEpi eu{};
      eu.act = c->hact;
      TRY(launch_ring(c, GEPI_SWIGLU, c->xn, w.wi, E, 2 * F, R, eu, false, st));
      // down-projection: few columns (E) x a long K.  With enough whole-K blocks it is a ring GEMM with the residual epilogue; at 256 rows
      // (32 such blocks at C2) the tiles are split over the 512-k chains instead (one f32 slice per chain) and the next kernel adds them up.
      const long down_blocks = (long)((E + 63) / 64) * ((R + 127) / 128);
      const bool down_ring = c->ring_down == 1 || (c->ring_down < 0 && down_blocks >= 192);
      if (down_ring) {
        GemmEpi ed{};
        ed.h_in = hnext; ed.h_out = hcur;
        TRY(launch_ring(c, GEPI_RESID, c->hact, w.wout, F, E, R, ed, false, st));
        if (next_scale != nullptr) TRY(launch_rownorm(c, 2, R, nullptr, nullptr, hcur, nullptr, next_scale, c->xn, st));
      } else {
        TRY(launch_dot(c, c->hact, w.wout, F, E, R, c->part, c->part_floats, false, st, &dr, true));
        if (next_scale != nullptr) {
          TRY(launch_rownorm(c, 1, R, nullptr, &dr, hnext, hcur, next_scale, c->xn, st));
        } else {
          hipLaunchKernelGGL(k_sumres, dim3((E + 255) / 256, R), dim3(128), 0, st, c->part, dr.slices, dr.slice_stride, hnext, hcur, E);
          LAUNCHCHK();
        }
      }
      continue;
    }
    TRY(launch_dot(c, c->xn, w.wqkv, E, c->NQKV, R, c->part, c->part_floats, false, st, &dr));
    const int attn_split = decode && !attn_tiled ? attn_split_factor(c, R) : 1;   // few decode rows: finalise first, then the keys of a head over several blocks
    if (!decode || attn_split > 1) {
      QkvArgs qa;
      qa.part = c->part; qa.slices = dr.slices; qa.slice_stride = dr.slice_stride; qa.ncols = c->NQKV;
      qa.row_slot = c->row_slot; qa.row_pos = c->row_pos; qa.row_active = c->row_active; qa.page_map = c->ps.page_map;
      qa.rope_sin = c->rope_sin; qa.rope_cos = c->rope_cos; qa.qbuf = c->qbuf;
      qa.kpages = c->kpages_l[l]; qa.vpages = c->vpages_l[l];
      qa.g = c->geom;
      hipLaunchKernelGGL(k_qkv_finalize, dim3(R, c->Q + 2 * c->KV), dim3(64), 0, st, qa);
      LAUNCHCHK();
    }
    // one block per (row, kv head) once that fills the chip; below that the key range is split over blocks (3 launches)
    if (attn_tiled) {
      TRY(launch_attention_tile(c, l, R, len_hint, st));
    } else if (attn_split > 1) {
      const int ta = timing_begin(c, st, 0.0, 1);
      TRY(launch_attn_split(c, l, R, attn_split, st));
      timing_end(c, st, ta);
    } else if (attn_one_launch) {   // decode rows, and prefill rows after k_qkv_finalize
      const int ta = decode ? timing_begin(c, st, 0.0, 1) : -1;   // K/V bytes depend on the rows' lengths: the caller prices them
      TRY(launch_attention_fused(c, l, R, dr, decode, st));
      timing_end(c, st, ta);
    } else {
      TRY(launch_attention_partials(c, l, R, len_hint, dr, decode, st));
      hipLaunchKernelGGL(k_attn_combine, dim3(R), dim3(256), 0, st, c->pvpart, c->row_pos, c->row_active, c->ao, c->geom, 1);
      LAUNCHCHK();
    }
    TRY(launch_dot(c, c->ao, w.wo, c->QN, E, R, c->part, c->part_floats, false, st, &dr));
    TRY(launch_rownorm(c, 1, R, nullptr, &dr, hcur, hnext, w.ln2, c->xn, st));
    if (c->moe) {
      // mixtral.py:123-160 / moe.py:855-942.  Every expert runs over ALL rows and only the rows that selected it keep the result: a row's
      // output does not depend on which other rows share its launch, so this equals the routed (permute / gmm / unpermute) form bit for
      // bit; it reads each expert's kernels once per pass, which is what decode with >= ~16 rows needs anyway, and spends
      // num_experts / num_experts_per_tok times the matrix-core work of the routed form on prefill (not yet built as a grouped GEMM).
      launch_moe_route(c, R, c->xn, w.gate, c->moe_idx, c->moe_w, st);
      LAUNCHCHK();
      if (c->moe_grouped) {   // only the routed work: every copy in the rows of its expert, unselected experts never read
        DotResult mr;
        TRY(moe_experts_routed(c, w, c->xn, c->moe_idx, R * c->TK, c->TK, st, &mr));
        hipLaunchKernelGGL(k_moe_combine_rows, dim3((E + 255) / 256, R), dim3(256), 0, st, c->part, mr.slices, mr.slice_stride, c->moe_pos, c->moe_w, E, c->TK, c->moe_comb);
        LAUNCHCHK();
        dr.slices = 1; dr.slice_stride = (size_t)R * E;
        if (next_scale != nullptr) {
          TRY(launch_rownorm(c, 1, R, nullptr, &dr, hnext, hcur, next_scale, c->xn, st, 1, nullptr, c->moe_comb));
        } else {
          hipLaunchKernelGGL(k_sumres, dim3((E + 255) / 256, R), dim3(128), 0, st, c->moe_comb, dr.slices, dr.slice_stride, hnext, hcur, E);
          LAUNCHCHK();
        }
        continue;
      }
      for (int e = 0; e < c->NE; ++e) {
        TRY(launch_dot(c, c->xn, w.ewi[e], E, 2 * F, R, c->part, c->part_floats, false, st, &dr));
        hipLaunchKernelGGL(k_swiglu_part, dim3((F + 2047) / 2048, R), dim3(256), 0, st, c->part, dr.slices, dr.slice_stride, F, c->hact);
        LAUNCHCHK();
        TRY(launch_dot(c, c->hact, w.ewout[e], F, E, R, c->part, c->part_floats, false, st, &dr));
        hipLaunchKernelGGL(k_moe_scatter, dim3((E + 255) / 256, R), dim3(256), 0, st, c->part, dr.slices, dr.slice_stride, E, e, c->TK, c->moe_idx, c->moe_y);
        LAUNCHCHK();
      }
      hipLaunchKernelGGL(k_moe_combine, dim3((E + 255) / 256, R), dim3(256), 0, st, c->moe_y, c->moe_w, E, c->TK, c->part);
      LAUNCHCHK();
      dr.slices = 1; dr.slice_stride = (size_t)R * E;
      if (next_scale != nullptr) {
        TRY(launch_rownorm(c, 1, R, nullptr, &dr, hnext, hcur, next_scale, c->xn, st));
      } else {
        hipLaunchKernelGGL(k_sumres, dim3((E + 255) / 256, R), dim3(128), 0, st, c->part, dr.slices, dr.slice_stride, hnext, hcur, E);
        LAUNCHCHK();
      }
      continue;
    }
    TRY(launch_dot(c, c->xn, w.wi, E, 2 * F, R, c->part, c->part_floats, false, st, &dr));
    hipLaunchKernelGGL(k_swiglu_part, dim3((F + 2047) / 2048, R), dim3(256), 0, st, c->part, dr.slices, dr.slice_stride, F, c->hact);
    LAUNCHCHK();
    TRY(launch_dot(c, c->hact, w.wout, F, E, R, c->part, c->part_floats, false, st, &dr));
    // residual + the next norm (the next layer's pre-attention norm, or the final norm when every row gets logits) in one launch
    if (next_scale != nullptr) {
      TRY(launch_rownorm(c, 1, R, nullptr, &dr, hnext, hcur, next_scale, c->xn, st));
    } else {
      hipLaunchKernelGGL(k_sumres, dim3((E + 255) / 256, R), dim3(128), 0, st, c->part, dr.slices, dr.slice_stride, hnext, hcur, E);
      LAUNCHCHK();
    }
  }
  if (nl > 0) {
    if (nl > c->NLmax) return set_err(IXT_ERR_INVALID, "too many logits rows (%d > %d)", nl, c->NLmax);
    if (!final_fused) TRY(launch_rownorm(c, 2, nl, nullptr, nullptr, lrows ? hcur : hcur + (size_t)lrow0 * E, nullptr, c->final_norm, c->xn, st, 1, lrows));
    TRY(launch_unembed(c, c->xn, nl, logits_out, argmax_out, st));
  }
  c->h_final = hcur; c->h_final_rows = R;
  return IXT_OK;
}

extern "C" int ixt_prefill_chunk(ixt_ctx* c, int slot, const int32_t* tokens_dev, int n_tokens, int start_pos, int32_t* first_token_dev,
                                 float* last_logits_dev, void* stream) {
  if (!c || !tokens_dev || !first_token_dev) return set_err(IXT_ERR_INVALID, "null argument");
  TRY(check_group(c, slot));
  if (n_tokens <= 0 || n_tokens > c->Rmax) return set_err(IXT_ERR_RANGE, "prefill chunk of %d rows (max %d)", n_tokens, c->Rmax);
  if (start_pos < 0 || start_pos + n_tokens > c->Tmax) return set_err(IXT_ERR_RANGE, "prefill beyond max_target_length");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_rows_prefill, dim3((n_tokens + 255) / 256), dim3(256), 0, st, c->row_slot, c->row_pos, c->row_active, slot, start_pos, n_tokens,
                     c->ps.num_pages_used, c->P);
  LAUNCHCHK();
  return forward_rows(c, n_tokens, tokens_dev, start_pos + n_tokens, n_tokens - 1, 1, nullptr, first_token_dev, last_logits_dev, false, st);
}

// maxengine.py:477-478 (`return_prompt_logp`): log_softmax(logits[t])[tokens[t]] of every row t of the chunk the last ixt_prefill_chunk call
// ran.  The pass leaves the rows' final hidden states behind; they are normalised and unembedded here NLmax rows at a time (the logits of a
// 256-row chunk at V = 264 192 are 270 MB: they never exist as a whole).
extern "C" int ixt_prefill_prompt_logp(ixt_ctx* c, const int32_t* tokens_dev, int n_tokens, float* prompt_logp_dev, void* stream) {
  if (!c || !tokens_dev || !prompt_logp_dev) return set_err(IXT_ERR_INVALID, "null argument");
  if (c->h_final == nullptr || n_tokens <= 0 || n_tokens != c->h_final_rows)
    return set_err(IXT_ERR_INVALID, "prompt log-probabilities: call right after ixt_prefill_chunk with the same n_tokens (%d rows are held)", c->h_final_rows);
  hipStream_t st = (hipStream_t)stream;
  if (c->plogits == nullptr) {
    TRY(dmalloc(c, &c->plogits, (size_t)c->NLmax * c->V));
    TRY(dmalloc(c, &c->pargmax, (size_t)c->NLmax));
  }
  const ixt_config& cf = c->cfg;
  for (int row0 = 0; row0 < n_tokens; row0 += c->NLmax) {
    const int g = n_tokens - row0 < c->NLmax ? n_tokens - row0 : c->NLmax;
    TRY(launch_rownorm(c, 2, g, nullptr, nullptr, c->h_final + (size_t)row0 * c->E, nullptr, c->final_norm, c->xn, st));
    TRY(launch_unembed(c, c->xn, g, c->plogits, c->pargmax, st));
    if (c->gemma) {   // decoders.py:552-565
      const int normalize = cf.logits_via_embedding && cf.normalize_embedding_logits;
      const float cap = cf.logits_via_embedding ? cf.final_logits_soft_cap : 0.0f;
      if (normalize || cap > 0.0f) TRY(launch_logits_post(c, c->plogits, g, normalize, cap, c->pargmax, st));
    }
    TRY(ixt_op_logprob(c, c->plogits, g, tokens_dev + row0, prompt_logp_dev + row0, stream));
  }
  return IXT_OK;
}

extern "C" int ixt_prefill_rows(ixt_ctx* c, int n_rows, const int32_t* tokens_dev, const int32_t* row_group_dev, const int32_t* row_pos_dev, int max_len,
                                int n_logits, const int32_t* logits_rows_dev, int32_t* first_tokens_dev, float* logits_dev, void* stream) {
  if (!c || !tokens_dev || !row_group_dev || !row_pos_dev || !logits_rows_dev || !first_tokens_dev) return set_err(IXT_ERR_INVALID, "null argument");
  if (n_rows <= 0 || n_rows > c->Rmax) return set_err(IXT_ERR_RANGE, "packed prefill of %d rows (max %d: ixt_config.max_packed_rows)", n_rows, c->Rmax);
  if (n_logits <= 0 || n_logits > c->NLmax || n_logits > n_rows) return set_err(IXT_ERR_RANGE, "packed prefill with %d prompts (max %d)", n_logits, c->NLmax);
  if (max_len <= 0 || max_len > c->Tmax) max_len = c->Tmax;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_rows_packed, dim3((n_rows + 255) / 256), dim3(256), 0, st, c->row_slot, c->row_pos, c->row_active, row_group_dev, row_pos_dev, n_rows,
                     c->ps.num_pages_used, c->P, c->G, c->Tmax, max_len, c->ps.status);
  LAUNCHCHK();
  HIPCHK(hipMemcpyAsync(c->lrows, logits_rows_dev, sizeof(int32_t) * n_logits, hipMemcpyDeviceToDevice, st));
  return forward_rows(c, n_rows, tokens_dev, max_len, 0, n_logits, c->lrows, first_tokens_dev, logits_dev, false, st);
}

static int decode_rows(ixt_ctx* c, const int32_t* tokens_dev, const int32_t* pos_dev, int32_t* next_tokens_dev, float* logits_dev, int check_len,
                       hipStream_t st) {
  hipLaunchKernelGGL(k_rows_decode, dim3(1), dim3(c->B <= 1024 ? ((c->B + 63) / 64) * 64 : 1024), 0, st, c->row_slot, c->row_pos, c->row_active, c->row_tok,
                     tokens_dev, pos_dev, c->ps.has_active_page, c->ps.num_pages_used, c->ps.sequence_lengths, check_len, c->P, c->Tmax, c->B);
  LAUNCHCHK();
  return forward_rows(c, c->B, c->row_tok, c->Tmax, 0, c->B, nullptr, next_tokens_dev, logits_dev, true, st);
}

extern "C" int ixt_decode_step(ixt_ctx* c, const int32_t* tokens_dev, const int32_t* pos_dev, int32_t* next_tokens_dev, float* logits_dev, void* stream) {
  if (!c || !tokens_dev || !pos_dev || !next_tokens_dev) return set_err(IXT_ERR_INVALID, "null argument");
  if (c->B > 1024) return set_err(IXT_ERR_UNSUPPORTED, "max_slots > 1024");
  return decode_rows(c, tokens_dev, pos_dev, next_tokens_dev, logits_dev, 0, (hipStream_t)stream);
}

static int generate_body(ixt_ctx* c, int32_t* tokens_dev, int32_t* next_pos_dev, int32_t* generated_dev, int32_t* result_data_dev, float* logits_dev,
                         hipStream_t st) {
  hipLaunchKernelGGL(k_pages_advance, dim3(1), dim3(64), 0, st, c->ps);
  LAUNCHCHK();
  TRY(decode_rows(c, tokens_dev, next_pos_dev, c->new_tok, logits_dev, 1, st));
  hipLaunchKernelGGL(k_generate_epilogue, dim3(1), dim3(((c->B + 63) / 64) * 64), 0, st, c->new_tok, tokens_dev, next_pos_dev, generated_dev, result_data_dev, c->B);
  LAUNCHCHK();
  return IXT_OK;
}

extern "C" int ixt_generate(ixt_ctx* c, int32_t* tokens_dev, int32_t* next_pos_dev, int32_t* generated_dev, int32_t* result_data_dev, float* logits_dev,
                            void* stream) {
  if (!c || !tokens_dev || !next_pos_dev || !generated_dev || !result_data_dev) return set_err(IXT_ERR_INVALID, "null argument");
  if (c->B > 1024) return set_err(IXT_ERR_UNSUPPORTED, "max_slots > 1024");
  hipStream_t st = (hipStream_t)stream;
  if (!c->cfg.use_hip_graph || c->timing) return generate_body(c, tokens_dev, next_pos_dev, generated_dev, result_data_dev, logits_dev, st);
  if (!ixt_weights_complete(c)) return set_err(IXT_ERR_NOT_LOADED, "weights incomplete");
  char key[160];
  snprintf(key, sizeof(key), "%p/%p/%p/%p/%p", (void*)tokens_dev, (void*)next_pos_dev, (void*)generated_dev, (void*)result_data_dev, (void*)logits_dev);
  auto it = c->graphs.find(key);
  if (it == c->graphs.end()) {
    // capture on a private stream so the caller's stream state is untouched
    hipStream_t cs;
    HIPCHK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    hipGraph_t graph = nullptr;
    HIPCHK(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
    c->capturing = true;
    int rc = generate_body(c, tokens_dev, next_pos_dev, generated_dev, result_data_dev, logits_dev, cs);
    c->capturing = false;
    hipError_t e = hipStreamEndCapture(cs, &graph);
    if (rc != IXT_OK || e != hipSuccess) {
      if (graph) (void)hipGraphDestroy(graph);
      (void)hipStreamDestroy(cs);
      return rc != IXT_OK ? rc : set_err(IXT_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
    }
    GraphEntry ge;
    e = hipGraphInstantiate(&ge.exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    (void)hipStreamDestroy(cs);
    if (e != hipSuccess) return set_err(IXT_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
    if (c->graphs.size() >= 8) {
      for (auto& kv : c->graphs) (void)hipGraphExecDestroy(kv.second.exec);
      c->graphs.clear();
    }
    it = c->graphs.emplace(std::string(key), ge).first;
  }
  HIPCHK(hipGraphLaunch(it->second.exec, st));
  return IXT_OK;
}

// --------------------------------------------------------------------------------------- op-level
__global__ void k_sum_slices(const float* part, int slices, size_t stride, float* out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float t = 0.0f;
  for (int s = 0; s < slices; ++s) t = t + part[(size_t)s * stride + i];
  out[i] = t;
}

static int transpose_rows(ixt_ctx* c, const bf16_t* x, int R, int K, hipStream_t st) {
  if ((size_t)((R + 15) / 16) * 16 * K > c->xt_cap) return set_err(IXT_ERR_INVALID, "op: %d rows x %d exceed the scratch of this ctx", R, K);
  hipLaunchKernelGGL(k_transpose_x, dim3((K + 255) / 256, R), dim3(256), 0, st, x, c->xt_op, R, K);
  LAUNCHCHK();
  return IXT_OK;
}

extern "C" int ixt_op_dot(ixt_ctx* c, const uint16_t* x, const uint16_t* w, float* out, int R, int K, int N, void* stream) {
  if (!c || !x || !w || !out || R <= 0) return set_err(IXT_ERR_INVALID, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (K % IXT_MK || N % 16) return set_err(IXT_ERR_UNSUPPORTED, "dot: K %% 32 or N %% 16");
  if ((size_t)((K + IXT_CHAIN - 1) / IXT_CHAIN) * R * N > c->part_floats && (size_t)R * N > c->part_floats)
    return set_err(IXT_ERR_INVALID, "op: %d x %d outputs exceed the scratch of this ctx", R, N);
  TRY(transpose_rows(c, x, R, K, st));
  const size_t Npad = (size_t)((N + 127) / 128) * 128;   // the ring layout pads the columns to whole 128-column groups
  if ((size_t)K * Npad > c->wt_cap) {
    if (c->wt_op) { HIPCHK(hipDeviceSynchronize()); HIPCHK(hipFree(c->wt_op)); c->wt_op = nullptr; c->wt_cap = 0; }
    HIPCHK(hipMalloc((void**)&c->wt_op, (size_t)K * Npad * sizeof(bf16_t)));
    c->wt_cap = (size_t)K * Npad;
  }
  const bool wring = c->ring_wlayout && R >= c->ring_min_rows && K % IXT_KALIGN == 0 && (size_t)R * N <= c->part_floats;
  if (wring) HIPCHK(hipMemsetAsync(c->wt_op, 0, (size_t)K * Npad * sizeof(bf16_t), st));
  TRY(tile_param(w, IXT_BF16, c->wt_op, K, N, 0, st, 0, 0, wring ? 1 : 0));
  DotResult dr;
  if (wring) {
    GemmEpi e{};
    e.out = c->part;
    TRY(launch_ring(c, GEPI_F32, c->xt_op, c->wt_op, K, N, R, e, false, st, 1));
    dr.slices = 1; dr.slice_stride = (size_t)R * N; dr.ntiles = N / 16;
  } else if (c->dot_small && sdot_plan(c, GEPI_F32, false, K, N, R).ok && (size_t)R * N <= c->part_floats) {   // the few-rows form, f32 epilogue (tests)
    GemmEpi e{};
    e.out = c->part;
    TRY(launch_sdot(c, GEPI_F32, false, c->xt_op, nullptr, c->wt_op, K, N, R, e, st));
    dr.slices = 1; dr.slice_stride = (size_t)R * N; dr.ntiles = N / 16;
  } else
  TRY(launch_dot(c, c->xt_op, c->wt_op, K, N, R, c->part, c->part_floats, false, st, &dr));
  const size_t n = (size_t)R * N;
  hipLaunchKernelGGL(k_sum_slices, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, c->part, dr.slices, dr.slice_stride, out, n);
  LAUNCHCHK();
  return IXT_OK;
}

// ---- Mixtral block piece by piece (layers/moe.py): what an expert-parallel caller (moe_ep.py) runs on each side of its exchange
static int check_moe_layer(ixt_ctx* c, int layer) {
  if (!c) return set_err(IXT_ERR_INVALID, "null ctx");
  if (!c->moe) return set_err(IXT_ERR_UNSUPPORTED, "decoder_block is not mixtral");
  if (layer < 0 || layer >= c->L) return set_err(IXT_ERR_RANGE, "layer %d out of range [0, %d)", layer, c->L);
  const uint32_t need = (1u << W_GATE) | (1u << W_WI0) | (1u << W_WI1) | (1u << W_WOUT);
  if ((c->loaded[layer] & need) != need) return set_err(IXT_ERR_NOT_LOADED, "MoeBlock_0 of layer %d is not loaded", layer);
  return IXT_OK;
}

extern "C" int ixt_op_moe_route(ixt_ctx* c, int layer, const uint16_t* x, int R, int32_t* experts_out, float* weights_out, void* stream) {
  TRY(check_moe_layer(c, layer));
  if (!x || !experts_out || !weights_out || R <= 0) return set_err(IXT_ERR_INVALID, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  TRY(transpose_rows(c, x, R, c->E, st));
  launch_moe_route(c, R, c->xt_op, c->layers[layer].gate, experts_out, weights_out, st);
  LAUNCHCHK();
  return IXT_OK;
}

extern "C" int ixt_op_moe_expert(ixt_ctx* c, int layer, int expert, const uint16_t* x, int R, uint16_t* out, void* stream) {
  TRY(check_moe_layer(c, layer));
  if (!x || !out || R <= 0 || R > c->Rmax) return set_err(IXT_ERR_INVALID, "bad argument (rows in (0, %d])", c->Rmax);
  if (expert < c->E0 || expert >= c->E0 + c->NEL) return set_err(IXT_ERR_RANGE, "expert %d is not one of this context's experts [%d, %d)", expert, c->E0, c->E0 + c->NEL);
  expert -= c->E0;
  hipStream_t st = (hipStream_t)stream;
  const int E = c->E, F = c->F;
  const LayerW& w = c->layers[layer];
  DotResult dr;
  TRY(transpose_rows(c, x, R, E, st));
  TRY(launch_dot(c, c->xt_op, w.ewi[expert], E, 2 * F, R, c->part, c->part_floats, false, st, &dr));
  hipLaunchKernelGGL(k_swiglu_part, dim3((F + 2047) / 2048, R), dim3(256), 0, st, c->part, dr.slices, dr.slice_stride, F, c->hact);
  LAUNCHCHK();
  TRY(launch_dot(c, c->hact, w.ewout[expert], F, E, R, c->part, c->part_floats, false, st, &dr));
  const size_t n = (size_t)R * E;
  hipLaunchKernelGGL(k_slices_to_bf16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, c->part, dr.slices, dr.slice_stride, n, out);
  LAUNCHCHK();
  return IXT_OK;
}

extern "C" int ixt_op_moe_experts(ixt_ctx* c, int layer, const uint16_t* x, const int32_t* expert_ids, int n, uint16_t* out, void* stream) {
  TRY(check_moe_layer(c, layer));
  if (!x || !expert_ids || !out || n <= 0) return set_err(IXT_ERR_INVALID, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  // the workspaces hold Rmax rows: more rows (the receive side of an exchange holds ranks x capacity of them, mostly padding) go through in pieces
  for (int r0 = 0; r0 < n; r0 += c->Rmax) {
    const int m = std::min(c->Rmax, n - r0);
    TRY(transpose_rows(c, x + (size_t)r0 * c->E, m, c->E, st));
    DotResult mr;
    TRY(moe_experts_routed(c, c->layers[layer], c->xt_op, expert_ids + r0, m, 1, st, &mr));
    hipLaunchKernelGGL(k_moe_rows_out, dim3((c->E + 255) / 256, m), dim3(256), 0, st, c->part, mr.slices, mr.slice_stride, c->moe_pos, c->E, out + (size_t)r0 * c->E);
    LAUNCHCHK();
  }
  return IXT_OK;
}

extern "C" int ixt_op_moe_combine(ixt_ctx* c, const uint16_t* y, const float* weights, int R, uint16_t* out, void* stream) {
  if (!c || !c->moe) return set_err(IXT_ERR_UNSUPPORTED, "decoder_block is not mixtral");
  if (!y || !weights || !out || R <= 0) return set_err(IXT_ERR_INVALID, "bad argument");
  hipLaunchKernelGGL(k_moe_combine, dim3((c->E + 255) / 256, R), dim3(256), 0, (hipStream_t)stream, y, weights, c->E, c->TK, (float*)nullptr, out);
  LAUNCHCHK();
  return IXT_OK;
}

extern "C" int ixt_op_rmsnorm(ixt_ctx* c, const uint16_t* x, const uint16_t* scale, uint16_t* out, int R, int E, void* stream) {
  if (!c || !x || !scale || !out || R <= 0 || E % 8) return set_err(IXT_ERR_INVALID, "bad argument");
  RowNormArgs a{};
  a.h_in = x; a.h_out = nullptr; a.scale = scale; a.xn_out = out; a.E = E; a.eps = c->cfg.normalization_layer_epsilon; a.xn_kmajor = 0;
  hipLaunchKernelGGL((k_rownorm<2>), dim3(R), dim3(256), sizeof(float) * ((size_t)E + 64), (hipStream_t)stream, a);
  LAUNCHCHK();
  return IXT_OK;
}

extern "C" int ixt_op_rope(ixt_ctx* c, uint16_t* x, const int32_t* pos, int R, int nheads, void* stream) {
  if (!c || !x || !pos || R <= 0) return set_err(IXT_ERR_INVALID, "bad argument");
  hipLaunchKernelGGL(k_rope, dim3(R, nheads), dim3(c->D / 2), 0, (hipStream_t)stream, x, pos, c->rope_sin, c->rope_cos, nheads, c->D, c->rope32 ? 1 : 0);
  LAUNCHCHK();
  return IXT_OK;
}

extern "C" int ixt_op_swiglu(ixt_ctx* c, const uint16_t* a, const uint16_t* b, uint16_t* out, int64_t n, void* stream) {
  if (!c || !a || !b || !out || n <= 0) return set_err(IXT_ERR_INVALID, "bad argument");
  hipLaunchKernelGGL(k_swiglu, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
  LAUNCHCHK();
  return IXT_OK;
}

extern "C" int ixt_op_geglu(ixt_ctx* c, const uint16_t* a, const uint16_t* b, uint16_t* out, int64_t n, void* stream) {
  if (!c || !a || !b || !out || n < 0) return set_err(IXT_ERR_INVALID, "bad argument");
  hipLaunchKernelGGL(k_geglu, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
  LAUNCHCHK();
  return IXT_OK;
}

extern "C" int ixt_op_residual(ixt_ctx* c, const uint16_t* a, const uint16_t* b, uint16_t* out, int64_t n, void* stream) {
  if (!c || !a || !b || !out || n <= 0) return set_err(IXT_ERR_INVALID, "bad argument");
  hipLaunchKernelGGL(k_residual, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
  LAUNCHCHK();
  return IXT_OK;
}

extern "C" int ixt_op_embed(ixt_ctx* c, const uint16_t* table, const int32_t* ids, uint16_t* out, int R, int E, int V, void* stream) {
  if (!c || !table || !ids || !out || R <= 0 || E <= 0 || E % 8 || V <= 0) return set_err(IXT_ERR_INVALID, "bad argument");
  hipLaunchKernelGGL(k_embed, dim3((E / 8 + 255) / 256, R), dim3(256), 0, (hipStream_t)stream, table, ids, out, E, V);
  LAUNCHCHK();
  return IXT_OK;
}

extern "C" int ixt_op_kv_write(ixt_ctx* c, int layer, const uint16_t* k, const uint16_t* v, const int32_t* row_slot, const int32_t* row_pos,
                               const int32_t* row_active, int R, void* stream) {
  if (!c || !k || !v || layer < 0 || layer >= c->L || R <= 0) return set_err(IXT_ERR_INVALID, "bad argument");
  hipLaunchKernelGGL(k_kv_write, dim3(R, c->KV), dim3(c->D), 0, (hipStream_t)stream, k, v, row_slot, row_pos, row_active, c->ps.page_map,
                     c->kpages_l[layer], c->vpages_l[layer], c->geom);
  LAUNCHCHK();
  return IXT_OK;
}

extern "C" int ixt_op_paged_attention(ixt_ctx* c, int layer, const uint16_t* q, const int32_t* row_slot, const int32_t* row_pos,
                                      const int32_t* row_active, int R, uint16_t* out, void* stream) {
  if (!c || !q || !out || layer < 0 || layer >= c->L || R <= 0 || R > c->Rmax) return set_err(IXT_ERR_INVALID, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(hipMemcpyAsync(c->row_slot, row_slot, sizeof(int32_t) * R, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(c->row_pos, row_pos, sizeof(int32_t) * R, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(c->row_active, row_active, sizeof(int32_t) * R, hipMemcpyDeviceToDevice, st));
  return launch_attention(c, layer, R, c->Tmax, q, out, st);
}

extern "C" int ixt_op_unembed_argmax(ixt_ctx* c, const uint16_t* y, int R, float* logits, int32_t* argmax, void* stream) {
  if (!c || !y || !argmax || R <= 0 || R > (c->B > 1 ? c->B : 1)) return set_err(IXT_ERR_INVALID, "bad argument");
  if (!(c->loaded[c->L] & 4)) return set_err(IXT_ERR_NOT_LOADED, "logits_dense/kernel not loaded");
  hipStream_t st = (hipStream_t)stream;
  TRY(transpose_rows(c, y, R, c->E, st));
  return launch_unembed(c, c->xt_op, R, logits, argmax, st);
}

extern "C" int ixt_kv_gather(ixt_ctx* c, int layer, int slot, int length, uint16_t* k_out, uint16_t* v_out, void* stream) {
  if (!c || !k_out || !v_out || layer < 0 || layer >= c->L || length <= 0 || length > c->Tmax) return set_err(IXT_ERR_INVALID, "bad argument");
  TRY(check_slot(c, slot));
  hipLaunchKernelGGL(k_kv_gather, dim3(length, c->KV), dim3(c->D), 0, (hipStream_t)stream, c->kpages_l[layer],
                     c->vpages_l[layer], c->ps.page_map, slot, c->geom, k_out, v_out);
  LAUNCHCHK();
  return IXT_OK;
}

extern "C" int ixt_rope_tables(ixt_ctx* c, uint16_t* sin_out, uint16_t* cos_out) {
  if (!c || !sin_out || !cos_out) return set_err(IXT_ERR_INVALID, "null argument");
  memcpy(sin_out, c->rope_sin_h.data(), c->rope_sin_h.size() * 2);
  memcpy(cos_out, c->rope_cos_h.data(), c->rope_cos_h.size() * 2);
  return IXT_OK;
}

extern "C" int ixt_rope_tables_f32(ixt_ctx* c, float* sin_out, float* cos_out) {
  if (!c || !sin_out || !cos_out) return set_err(IXT_ERR_INVALID, "null argument");
  if (!c->rope32) return set_err(IXT_ERR_UNSUPPORTED, "this context was not created with rope_llama31");
  memcpy(sin_out, c->rope_sin32_h.data(), c->rope_sin32_h.size() * 4);
  memcpy(cos_out, c->rope_cos32_h.data(), c->rope_cos32_h.size() * 4);
  return IXT_OK;
}

// --------------------------------------------------------------------------------------- sampling columns, options, timing
extern "C" int ixt_set_sampling_ranges(ixt_ctx* c, const int32_t* lo_hi, int nranges) {
  if (!c || nranges < 0 || nranges > 4 || (nranges > 0 && !lo_hi)) return set_err(IXT_ERR_INVALID, "0..4 ranges");
  ColRanges cr{};
  for (int i = 0; i < nranges; ++i) {
    const int lo = lo_hi[2 * i], hi = lo_hi[2 * i + 1];
    if (lo < 0 || hi > c->V || lo >= hi) return set_err(IXT_ERR_RANGE, "sampling range [%d, %d) outside [0, %d)", lo, hi, c->V);
    cr.lo[i] = lo; cr.hi[i] = hi;
  }
  cr.n = nranges;
  HIPCHK(hipDeviceSynchronize());
  for (auto& kv : c->graphs) (void)hipGraphExecDestroy(kv.second.exec);   // captured launches carry the old ranges as kernel arguments
  c->graphs.clear();
  c->sample_cols = cr;
  return IXT_OK;
}

extern "C" int ixt_set_option(ixt_ctx* c, const char* name, int value) {
  if (!c || !name) return set_err(IXT_ERR_INVALID, "null argument");
  struct Opt { const char* name; int* field; int lo; };
  const Opt opts[] = {
      {"attn_fused_min_blocks", &c->attn_fused_min_blocks, 1}, {"gemm_prefix_halves", &c->gemm_prefix_halves, 0},
      {"gemm_split_min_blocks", &c->gemm_split_min_blocks, 1}, {"gemm_min_blocks", &c->gemm_min_blocks, 1}, {"gemm_min_rows", &c->gemm_min_rows, 1},
      {"unembed_gemm", &c->unembed_gemm, 0}, {"attn_wave_pairs", &c->attn_wave_pairs, 0}, {"attn_tile", &c->attn_tile, 0}, {"attn_tile_min_rows", &c->attn_tile_min_rows, 1}, {"dot_stream_min_blocks", &c->dot_stream_min_blocks, 1},
      {"ring_min_rows", &c->ring_min_rows, 1}, {"ring_cfg", &c->ring_cfg, 0}, {"ring_split", &c->ring_split, 0}, {"ring_wlayout", &c->ring_wlayout, 0}, {"ring_down", &c->ring_down, -1}, {"unembed_tall", &c->unembed_tall, 0}, {"ring_two_per_cu", &c->ring_two_per_cu, 0},
      {"small_rows", &c->small_rows, 0}, {"small_tpb", &c->small_tpb, 0}, {"small_attn_waves", &c->small_attn_waves, 4}, {"small_attn_max_blocks", &c->small_attn_max_blocks, 0}, {"dot_small", &c->dot_small, 0}, {"small_norm", &c->small_norm, 0}, {"small_prefetch", &c->small_prefetch, 0}, {"small_prefetch_blocks", &c->small_prefetch_blocks, 1}, {"moe_grouped", &c->moe_grouped, 0}, {"small_attn_split", &c->small_attn_split, 0}, {"moe_route_blocks", &c->moe_route_blocks, 0}, {"gemma_tile", &c->gemma_tile, 0}, {"moe_fused_gate", &c->moe_fused_gate, 0}, {"moe_mid_pairs", &c->moe_mid_pairs, 0},
  };
  for (const Opt& o : opts) {
    if (strcmp(name, o.name) != 0) continue;
    if (value < o.lo) return set_err(IXT_ERR_RANGE, "%s must be >= %d", name, o.lo);
    HIPCHK(hipDeviceSynchronize());
    for (auto& kv : c->graphs) (void)hipGraphExecDestroy(kv.second.exec);   // captured launch sequences embed the old choice
    c->graphs.clear();
    *o.field = value;
    return IXT_OK;
  }
  return set_err(IXT_ERR_INVALID, "unknown option %s", name);
}

// f2bf (hardware conversion) against the integer round-to-nearest-even formula over all 2^32 f32 bit patterns: the number of non-NaN
// inputs on which they differ, plus the number of NaN inputs that do not stay NaN (tests/test_gpu_ops.py).
__global__ __launch_bounds__(256) void k_cvt_check(unsigned long long* bad) {
  unsigned long long n = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < (1ull << 32); i += (unsigned long long)gridDim.x * 256) {
    const uint32_t u = (uint32_t)i;
    const float f = __uint_as_float(u);
    const bf16_t hw = f2bf(f), sw = f2bf_int(f);
    const bool nan = (u & 0x7fffffffu) > 0x7f800000u;
    if (nan ? ((hw & 0x7fffu) <= 0x7f80u) : (hw != sw)) ++n;
  }
  if (n) atomicAdd(bad, n);
}
// IXT-FP8-v1 dot of caller-supplied operands (tests): x bf16 [R][K] row-major, w bf16 [K][N] row-major -> out f32 [R][N]
extern "C" int ixt_op_dot_fp8(ixt_ctx* c, const uint16_t* x, const uint16_t* w, float* out, int R, int K, int N, void* stream) {
  if (!c || !x || !w || !out || R <= 0) return set_err(IXT_ERR_INVALID, "bad argument");
  if (!c->fp8) return set_err(IXT_ERR_UNSUPPORTED, "this context was not created with fp8_weights");
  hipStream_t st = (hipStream_t)stream;
  if (K % 64 || N % 16) return set_err(IXT_ERR_UNSUPPORTED, "fp8 dot: K %% 64 or N %% 16");
  if ((size_t)((K + IXT_CHAIN - 1) / IXT_CHAIN) * R * N > c->part_floats) return set_err(IXT_ERR_INVALID, "op: %d x %d outputs exceed the scratch of this ctx", R, N);
  TRY(transpose_rows(c, x, R, K, st));
  bf16_t* wt = nullptr;
  ixt_ctx::F8W t{nullptr, nullptr, K, N};
  HIPCHK(hipMalloc((void**)&wt, (size_t)K * N * sizeof(bf16_t)));
  HIPCHK(hipMalloc((void**)&t.w8, (size_t)K * N));
  HIPCHK(hipMalloc((void**)&t.s, (size_t)N * sizeof(float)));
  int rc = tile_param(w, IXT_BF16, wt, K, N, 0, st);
  if (rc == IXT_OK) {
    hipLaunchKernelGGL(k_f8_quant_w, dim3(N), dim3(256), 0, st, wt, K, N, t.w8, t.s);
    DotResult dr;
    // the activation workspace is sized for the model's widest K: the op's K must fit
    const int kmax8 = std::max(std::max(c->F, c->QN), c->E);
    if (K > kmax8 || R > c->Rmax) rc = set_err(IXT_ERR_INVALID, "op: %d rows x K %d exceed the fp8 workspace", R, K);
    else rc = launch_dot8(c, c->xt_op, t, K, N, R, c->part, c->part_floats, st, &dr);
    if (rc == IXT_OK) {
      const size_t n = (size_t)R * N;
      hipLaunchKernelGGL(k_sum_slices, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, c->part, dr.slices, dr.slice_stride, out, n);
    }
  }
  (void)hipStreamSynchronize(st);
  (void)hipFree(wt); (void)hipFree(t.w8); (void)hipFree(t.s);
  return rc;
}

extern "C" int ixt_op_fp8_roundtrip(ixt_ctx* c, const float* in_dev, float* out_dev, int64_t n, void* stream) {
  if (!c || !in_dev || !out_dev || n <= 0) return set_err(IXT_ERR_INVALID, "bad argument");
  hipLaunchKernelGGL(k_f8_roundtrip, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in_dev, out_dev, n);
  LAUNCHCHK();
  return IXT_OK;
}
extern "C" int ixt_op_cvt_check(ixt_ctx* c, int64_t* mismatches_out, void* stream) {
  if (!c || !mismatches_out) return set_err(IXT_ERR_INVALID, "null argument");
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* d = nullptr;
  HIPCHK(hipMalloc((void**)&d, sizeof(unsigned long long)));
  HIPCHK(hipMemsetAsync(d, 0, sizeof(unsigned long long), st));
  hipLaunchKernelGGL(k_cvt_check, dim3(4096), dim3(256), 0, st, d);
  unsigned long long h = 0;
  HIPCHK(hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipFree(d));
  *mismatches_out = (int64_t)h;
  return IXT_OK;
}

extern "C" int ixt_timing_enable(ixt_ctx* c, int enable) {
  if (!c) return set_err(IXT_ERR_INVALID, "null ctx");
  c->timing = enable != 0;
  return IXT_OK;
}
extern "C" int ixt_timing_reset(ixt_ctx* c) {
  if (!c) return set_err(IXT_ERR_INVALID, "null ctx");
  (void)hipDeviceSynchronize();
  c->tused = 0;
  return IXT_OK;
}
extern "C" int ixt_timing_get(ixt_ctx* c, const char* family, double* avg_us, int64_t* launches, double* bytes_per_launch) {
  if (!c || !family) return set_err(IXT_ERR_INVALID, "null argument");
  int fam;
  if (strcmp(family, "dot") == 0 || strcmp(family, "dot_decode") == 0) fam = 0;
  else if (strcmp(family, "attn_decode") == 0) fam = 1;
  else return set_err(IXT_ERR_INVALID, "unknown family %s", family);
  HIPCHK(hipDeviceSynchronize());
  double tot_ms = 0.0, tot_bytes = 0.0;
  size_t cnt = 0;
  for (size_t i = 0; i < c->tused; ++i) {
    if (c->tpairs[i].family != fam) continue;
    float ms = 0.0f;
    HIPCHK(hipEventElapsedTime(&ms, c->tpairs[i].a, c->tpairs[i].b));
    tot_ms += ms;
    tot_bytes += c->tpairs[i].bytes;
    ++cnt;
  }
  const double n = cnt ? (double)cnt : 1.0;
  if (avg_us) *avg_us = tot_ms * 1000.0 / n;
  if (launches) *launches = (int64_t)cnt;
  if (bytes_per_launch) *bytes_per_launch = tot_bytes / n;
  return IXT_OK;
}

// the training step (include/ixt_train.h) shares this translation unit: the kernels of ixt_kernels.h are defined once
#include "ixt_train.hip"
