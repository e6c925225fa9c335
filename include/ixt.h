/*
 * ixt.h -- C ABI of libixt.so: the MI355X (gfx950) hot path of an IndexTTS2-style
 * semantic-token LM: chunked prompt prefill + per-step audio-token decode over a paged KV cache.
 *
 * Boundary rules: plain C, `int` status return (0 = ok, negative = ixt_status,
 * never throws), device pointers are plain `void*`/typed pointers owned by the caller unless stated,
 * every compute call takes an explicit hipStream_t (passed as void*), no allocation after
 * ixt_create/ixt_load_weight, one ctx per GPU, calls on one ctx externally serialised (the reference
 * engine has no locks either: maxengine.py:550,849,1190-1191,1325-1328).
 *
 * Each entry point names the reference interface it replaces (paths relative to the reference repo).
 */
#ifndef IXT_H_
#define IXT_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum ixt_status {
  IXT_OK = 0,
  IXT_ERR_INVALID = -1,      /* bad argument / shape (reference: ValueError)                    */
  IXT_ERR_HIP = -2,          /* a HIP runtime call failed; see ixt_last_error()                  */
  IXT_ERR_UNSUPPORTED = -3,  /* configuration outside the built path (reference: NotImplementedError) */
  IXT_ERR_NOT_LOADED = -4,   /* a weight needed by the call was never loaded                     */
  IXT_ERR_RANGE = -5         /* slot / length out of range (page_manager.py:529-532)             */
} ixt_status;

typedef enum ixt_dtype { IXT_BF16 = 0, IXT_F32 = 1 } ixt_dtype;

/* Derived hyper-parameters of MaxText/pyconfig.py:551-644 that shape the path. */
typedef struct ixt_config {
  int32_t vocab_size;              /* base.yml:430 */
  int32_t emb_dim;                 /* base.yml:132 (x scale) */
  int32_t num_query_heads;         /* base.yml:133 */
  int32_t num_kv_heads;            /* base.yml:134 */
  int32_t head_dim;                /* base.yml:137; 64 or 128 */
  int32_t mlp_dim;                 /* base.yml:135 */
  int32_t num_decoder_layers;      /* base.yml:136 */
  float normalization_layer_epsilon; /* base.yml:21 */
  float rope_min_timescale;        /* base.yml:601 */
  float rope_max_timescale;        /* base.yml:602 */
  int32_t max_prefill_predict_length; /* base.yml:528: max rows of one prefill chunk */
  int32_t max_target_length;       /* base.yml:527 */
  int32_t max_slots;               /* decode batch = per_device_batch_size (maxengine.py:1455-1458) */
  int32_t pagedattn_num_pages;     /* base.yml:740 */
  int32_t pagedattn_tokens_per_page; /* base.yml:741; power of two, 8..64 */
  int32_t pagedattn_max_pages_per_group; /* base.yml:743 (already derived, > 0) */
  int32_t use_hip_graph;           /* build knob: capture ixt_generate into one hipGraph */
  int32_t max_packed_rows;         /* build knob: most rows of one ixt_prefill_rows call (0: max_prefill_predict_length) */
  int32_t staging_groups;          /* build knob: page groups behind the decode slots that a packed prefill (MaxEngine.prefill_concat,
                                      maxengine.py:719-836) fills before insert_partial hands their pages to slots; 0 = none */
  /* Decoder block variant.  All zero = the llama2 block (llama2.py:54-169).
   * IXT_BLOCK_GEMMA2 (layers/gemma2.py:69-236, configs/models/gemma2-2b.yml): `num_decoder_layers` counts the reference's merged
   * [local, global] layers; the context then holds 2 x num_decoder_layers (attention, MLP) sub-layers -- sub-layer 2 i is layer i's
   * "local" half, 2 i + 1 its "global" half -- and every `layer` argument of this header (ixt_kv_gather, ixt_op_kv_write,
   * ixt_op_paged_attention) indexes sub-layers.  The fields below are read for gemma2 / gemma3 only and must be zero otherwise. */
  int32_t decoder_block;           /* base.yml decoder_block: IXT_BLOCK_LLAMA2 | IXT_BLOCK_GEMMA2 | IXT_BLOCK_GEMMA3 | IXT_BLOCK_MIXTRAL */
  float attn_logits_soft_cap;      /* attentions.py:1257-1259: score = tanh(score / cap) * cap in f32 before the softmax; 0 = none */
  float final_logits_soft_cap;     /* decoders.py:563-565 (inside the logits_via_embedding branch, as in the reference); 0 = none */
  int32_t sliding_window_size;     /* attentions.py:625-631, local sub-layers: key j visible iff pos - window < j <= pos; 0 = none */
  int32_t use_post_attn_norm;      /* gemma2.py:107-116, :195-203: RMSNorm of the attention output before the residual add */
  int32_t use_post_ffw_norm;       /* gemma2.py:135-143, :227-235: RMSNorm of the MLP output before the residual add */
  int32_t logits_via_embedding;    /* decoders.py:552-558: logits = y . embedding^T (no logits_dense kernel is loaded) */
  int32_t normalize_embedding_logits; /* decoders.py:560-562: logits / sqrt(emb_dim), in f32 (the divisor is an f32 array) */
  /* IXT_BLOCK_GEMMA3 (layers/gemma3.py:35-200, configs/models/gemma3-4b.yml): one (attention, MLP) pair per layer with the Gemma-2 fields
   * above (post norms, GELU gate, caps, tied head); layer i is local sliding unless i % 6 == 5 (GEMMA3_ATTENTION_PATTERN, gemma3.py:35-47);
   * every query / key head is RMS-normalised over head_dim before RoPE (`self_attention/{query,key}_norm/scale`, attentions.py:1872-1888,
   * :2246-2248).  Read for gemma3 only: */
  float local_rope_max_timescale;  /* base.yml:603, attentions.py:2085-2088: RoPE base of the local layers; <= 0: rope_max_timescale */
  float query_pre_attn_scalar;     /* gemma3.py:50-58, attentions.py:2262-2264: q = bf16(q * bf16(scalar)) after RoPE; 0 or 1: none */
  /* IXT_BLOCK_MIXTRAL (layers/mixtral.py:52-170, layers/moe.py:205-1490, configs/models/mixtral-8x7b.yml): the
   * llama2 wiring with the MLP replaced by a routed mixture -- gate logits (moe.py:170-204), jax.lax.top_k + softmax over the k logits
   * (:326-345), the selected experts' silu MLPs, sum_k weight_k y_k in f32 (:455-484).  Weights: `MoeBlock_0/gate/kernel` [emb, experts],
   * `MoeBlock_0/wi_0`, `MoeBlock_0/wi_1` [experts, emb, mlp], `MoeBlock_0/wo` [experts, mlp, emb]; the pre-MoE norm is
   * `post_self_attention_layer_norm/scale`.  Read for mixtral only (0 or 1 otherwise): */
  int32_t num_experts;             /* base.yml:160; <= 32 */
  int32_t num_experts_per_tok;     /* base.yml:161; <= min(num_experts, 8) */
  /* Expert parallelism (moe.py:785-938: the experts of a layer split over the ranks of the `expert` mesh axis): this context keeps the
   * kernels of experts [ep_rank E/ep_size, (ep_rank + 1) E/ep_size) only -- `MoeBlock_0/wi_0`, `wi_1`, `wo` are still handed over whole
   * and the other experts' slices are skipped.  ep_size <= 1: every expert.  A sharded context serves ixt_op_moe_route / ixt_op_moe_experts
   * (the two sides of the caller's exchange); the fused engine entry points need every expert and refuse. */
  int32_t ep_rank, ep_size;
  /* base.yml:93 `quantization: "fp8"` (BASELINE.json configs[3]).  The reference installs fp8 for training only (quantizations.py:289-298) and refuses
   * it in decode (decode.py:127-128), so the scheme is this build's own, "IXT-FP8-v1" (csrc/ixt_fp8.h): the four projection
   * kernels of every layer (and every expert) are kept as e4m3 with one f32 scale per output column, activations are quantised to e4m3 per row
   * and 512-k block on the way in, the products run on v_mfma_f32_16x16x32_fp8_fp8 with f32 accumulation; embedding, norms, attention and the
   * logits head stay bf16.  Not bit-comparable with the bf16 path: tests/test_gpu_fp8.py states the error. */
  int32_t fp8_weights;
  /* base.yml:599-600 `rope_type: "llama3.1"` (layers/embeddings.py:354-488 LLaMARotaryEmbedding): the rotation pairs are the NEIGHBOURING features
   * (2 i, 2 i + 1), sin / cos are f32 and the result rounds to bf16 once; rope_use_scale applies the llama-3.1 frequency scaling
   * (factor 8, low / high frequency factors 1 / 4, original context 8192; :417-443).  0: the default RotaryEmbedding (:216-351).  llama2 block only. */
  int32_t rope_llama31, rope_use_scale;
} ixt_config;

enum { IXT_BLOCK_LLAMA2 = 0, IXT_BLOCK_GEMMA2 = 1, IXT_BLOCK_GEMMA3 = 2, IXT_BLOCK_MIXTRAL = 3 };

typedef struct ixt_ctx ixt_ctx;

/* Human-readable text of the last error on this thread. */
const char* ixt_last_error(void);

/* MaxEngine.__init__ (maxengine.py:107-136): allocates weights storage, page pools, PageState
 * (page_manager.py:93-126: page 0 permanently reserved), RoPE tables and all workspaces on the
 * current HIP device. */
int ixt_create(const ixt_config* cfg, ixt_ctx** out);
int ixt_destroy(ixt_ctx* ctx);

/* MaxEngine.load_params (maxengine.py:218-271): copy one parameter into the ctx's HBM layout.
 * `maxtext_path` is the unscanned tree path, e.g. "params/decoder/layers_3/self_attention/query/kernel";
 * both spellings of the MLP pre-norm are accepted ("mlp/mlp_layer_norm/scale" of linears.py:347-354 and
 * "post_self_attention_layer_norm/scale" of llama_or_mistral_ckpt.py:449-450).  `dev_ptr` is a device
 * pointer to a dense row-major array of `shape[ndim]`; IXT_F32 sources are rounded to bf16
 * (round-to-nearest-even) as linears.py:216 / embeddings.py:154 do at use.  Projection kernels are re-tiled into the
 * matrix-fragment order of the GEMM kernels while they are copied (the caller's array is not kept). */
int ixt_load_weight(ixt_ctx* ctx, const char* maxtext_path, const void* dev_ptr, int dtype, const int64_t* shape, int ndim, void* stream);
/* 1 when every parameter of the configured model has been loaded. */
int ixt_weights_complete(ixt_ctx* ctx);

/* ---- PageState / PageManager (inference/page_manager.py), state lives in HBM ---------------------- */
/* PageManager.get_initial_page_state (page_manager.py:600-621). */
int ixt_pages_reset(ixt_ctx* ctx, void* stream);
/* PageManager.update_prefill_pages (page_manager.py:493-536): release the group, then first-fit reserve
 * ceil(true_length/page) pages; silently leaves the group empty when resources are short (:305-311). */
int ixt_pages_reserve(ixt_ctx* ctx, int slot, int true_length, void* stream);
/* Build extension for chunked prefill over pages (no reference counterpart): keep the
 * group's pages and grow the allocation to total_length tokens. */
int ixt_pages_extend(ixt_ctx* ctx, int slot, int total_length, void* stream);
/* PageManager.update_decode_pages (page_manager.py:331-412,538-566). */
int ixt_pages_advance(ixt_ctx* ctx, void* stream);
/* PageManager.release_pages (page_manager.py:568-598). */
int ixt_pages_release(ixt_ctx* ctx, int slot, void* stream);
/* MaxEngine.insert's `has_active_page.at[slot].set(True)` (maxengine.py:1189-1191). */
int ixt_pages_set_active(ixt_ctx* ctx, int slot, int active, void* stream);
/* ---- build extensions around MaxEngine.insert / prefill_concat (no function of the reference's page manager does these; each states the
 * engine code it serves) ---- */
/* MaxEngine.insert (maxengine.py:1166-1192) of a `length`-token prefix into the slot that already holds its pages: has_active_page = 1
 * (:1189-1191) and sequence_lengths / active_page / active_page_position back to their post-prefill values (a generate() between
 * prefill and insert has advanced them). */
int ixt_pages_resync(ixt_ctx* ctx, int slot, int length, void* stream);
/* MaxEngine.insert_partial (maxengine.py:1206-1318) over pages: slot `dst_slot` releases its pages and takes over the page list of
 * group `src_group` (a staging group a packed prefill has filled); no K/V byte moves; src_group ends empty. */
int ixt_pages_move(ixt_ctx* ctx, int src_group, int dst_slot, void* stream);
/* MaxEngine.insert / bulk_insert into a slot other than the one prefilled (the page copy loop, maxengine.py:1106-1125): release + reserve
 * `length` tokens for dst_slot, then copy the K and V pages of src_group, all layers.  src_group stays valid. */
int ixt_pages_copy(ixt_ctx* ctx, int src_group, int dst_slot, int length, void* stream);
/* Page-pool shortage is silent in the reference (page_manager.py:305-311).  out4 (HOST) = {reservations that found no room since the
 * last reset, the last group that failed, decode rows left without a page, packed prefill rows at or beyond their call's max_len};
 * synchronises `stream`; reset != 0 clears the counters. */
int ixt_pages_status(ixt_ctx* ctx, int32_t* out4, int reset, void* stream);
/* The same four words copied to PINNED host memory in stream order, without waiting: the caller records an event behind the call and reads
 * the words once it has completed (a throughput path checks the reservations of packed prefill N while pass N already runs). */
int ixt_pages_status_async(ixt_ctx* ctx, int32_t* host_out4, void* stream);
/* Few decode rows: the keys of a (row, kv head) are split over several thread blocks that exchange their softmax statistics through HBM
 * inside one launch (k_attn_split; option small_attn_split).  Their waits are bounded; out1 (HOST) = 1 if a wait ever gave up since the last
 * reset (the affected step's numbers are then wrong -- the reference has no counterpart: its generate step, maxengine.py:838-935, is one XLA
 * program), 0 otherwise.  Synchronises `stream`; reset != 0 re-arms the exchange. */
int ixt_exchange_status(ixt_ctx* ctx, int32_t* out1, int reset, void* stream);
/* Copy the seven PageState fields to HOST arrays (synchronises `stream`); the decode slots' groups only:
 * page_status[num_pages], page_map[max_slots*max_pages_per_group], num_pages_used[max_slots],
 * sequence_lengths[max_slots], active_page[max_slots], has_active_page[max_slots] (0/1),
 * active_page_position[max_slots]. */
int ixt_pages_export(ixt_ctx* ctx, int32_t* page_status, int32_t* page_map, int32_t* num_pages_used,
                     int32_t* sequence_lengths, int32_t* active_page, uint8_t* has_active_page,
                     int32_t* active_page_position, void* stream);
/* Overwrite page_status from a HOST array (test hook for page_manager_test.py:143-160 "no space"). */
int ixt_pages_import_status(ixt_ctx* ctx, const int32_t* page_status, void* stream);

/* ---- engine-level compute ------------------------------------------------------------------------- */
/* MaxEngine._prefill_jit body (maxengine.py:400-530) for one chunk of one slot, pages already reserved:
 * rows = tokens_dev[0..n_tokens) at positions start_pos.., K/V written in place to the slot's pages,
 * first_token_dev[0] = argmax of the last row's logits (:483-499).  last_logits_dev (f32[vocab]) may be
 * NULL.  n_tokens is the TRUE length of the chunk (padding rows are never computed). */
/* `return_prompt_logp` of MaxEngine._prefill_jit (maxengine.py:477-478): prompt_logp_dev f32[n_tokens] = log_softmax(logits[t])[tokens[t]] for
 * every row t of the chunk that the LAST ixt_prefill_chunk call on this ctx ran (same tokens_dev / n_tokens, nothing in between); the rows
 * are unembedded max(max_slots, staging_groups) at a time into a workspace allocated on the first call. */
int ixt_prefill_prompt_logp(ixt_ctx* ctx, const int32_t* tokens_dev, int n_tokens, float* prompt_logp_dev, void* stream);
int ixt_prefill_chunk(ixt_ctx* ctx, int slot, const int32_t* tokens_dev, int n_tokens, int start_pos,
                      int32_t* first_token_dev, float* last_logits_dev, void* stream);
/* MaxEngine.prefill_concat's model pass (maxengine.py:719-836) over pages: n_rows rows of several prompts as one ragged batch -- row i =
 * token tokens_dev[i] at position row_pos_dev[i] of page group row_group_dev[i] (pages already reserved; decode slots or staging groups);
 * rows of different groups never see each other.  first_tokens_dev[j] = argmax of the logits of row logits_rows_dev[j] (the last row of
 * prompt j, :797-812), logits_dev f32[n_logits][vocab] may be NULL.  All arrays are device arrays.  max_len: the caller's bound on
 * position + 1 over the rows (the longest prompt end; it sizes the on-chip score rows of the tiled attention; <= 0: max_target_length);
 * a row at or beyond it is left idle and counted in ixt_pages_status()[3]. */
int ixt_prefill_rows(ixt_ctx* ctx, int n_rows, const int32_t* tokens_dev, const int32_t* row_group_dev, const int32_t* row_pos_dev, int max_len,
                     int n_logits, const int32_t* logits_rows_dev, int32_t* first_tokens_dev, float* logits_dev, void* stream);
/* MaxEngine._generate_jit model step (maxengine.py:868-936) for all slots: row b = token tokens_dev[b] at
 * position pos_dev[b]; rows without an active page, outside max_target_length or beyond the pages granted to the slot write no K/V
 * and attend to nothing (ixt_generate also requires pos_dev[b] == sequence_lengths[b] - 1: a slot prefilled but not inserted idles).
 * next_tokens_dev[b] = argmax; logits_dev (f32[max_slots*vocab]) may be NULL. */
int ixt_decode_step(ixt_ctx* ctx, const int32_t* tokens_dev, const int32_t* pos_dev, int32_t* next_tokens_dev,
                    float* logits_dev, void* stream);
/* MaxEngine.generate (maxengine.py:838-866): ixt_pages_advance + ixt_decode_step + the state update of
 * :913-935 in one call (one hipGraph launch when cfg.use_hip_graph): tokens <- new token, next_pos += 1,
 * generated += 1, result_data[b] = {token, 1, generated} (ResultTokens layout, :915-926). All int32[max_slots]
 * device arrays except result_data int32[max_slots*3]. */
int ixt_generate(ixt_ctx* ctx, int32_t* tokens_dev, int32_t* next_pos_dev, int32_t* generated_dev,
                 int32_t* result_data_dev, float* logits_dev, void* stream);

/* ---- op-level entry points (unit parity against the oracle) ---------------------------------------- */
/* DenseGeneral.__call__ (linears.py:188-232) in canonical order: out f32[R*N] = x bf16[R*K] . W bf16[K*N] (both row-major;
 * K % 32 == 0, N % 16 == 0; W is re-tiled into a scratch copy first). */
int ixt_op_dot(ixt_ctx* ctx, const uint16_t* x, const uint16_t* w, float* out, int R, int K, int N, void* stream);
/* RMSNorm.__call__ (normalizations.py:57-70): out bf16[R*E]. */
int ixt_op_rmsnorm(ixt_ctx* ctx, const uint16_t* x, const uint16_t* scale, uint16_t* out, int R, int E, void* stream);
/* RotaryEmbedding.__call__ (embeddings.py:277-315) in place on x bf16[R][nheads][head_dim], positions int32[R]. */
int ixt_op_rope(ixt_ctx* ctx, uint16_t* x, const int32_t* pos, int R, int nheads, void* stream);
/* The Mixtral block piece by piece (decoder_block mixtral, weights of `layer` loaded), for a caller that places an exchange between the
 * pieces (expert parallelism, layers/moe.py:785-938): x bf16 [R][emb] row-major (the post_self_attention_layer_norm output).
 *   route   (moe.py:170-204, :326-345): experts_out int32 [R][num_experts_per_tok] in top_k order, weights_out f32 [R][k] (bf16 values);
 *   expert  (moe.py:862-874): out bf16 [R][emb] = wo_e(bf16(bf16(silu(wi_0_e x)) * wi_1_e x)) of one expert over all R rows (R <= the ctx's
 *           row capacity);
 *   combine (moe.py:455-484): y bf16 [R][k][emb] -> out bf16 [R][emb] = bf16(fma chain over j of y[r][j] * weights[r][j]). */
int ixt_op_moe_route(ixt_ctx* ctx, int layer, const uint16_t* x, int R, int32_t* experts_out, float* weights_out, void* stream);
int ixt_op_moe_expert(ixt_ctx* ctx, int layer, int expert, const uint16_t* x, int R, uint16_t* out, void* stream);
int ixt_op_moe_combine(ixt_ctx* ctx, const uint16_t* y, const float* weights, int R, uint16_t* out, void* stream);
/* MlpBlock gate (linears.py:459-464): out = bf16(bf16(silu(a)) * b), n elements. */
int ixt_op_swiglu(ixt_ctx* ctx, const uint16_t* a, const uint16_t* b, uint16_t* out, int64_t n, void* stream);
/* The Gemma gate (mlp_activations ["gelu", "linear"], linears.py:449-464 with jax.nn.gelu approximate=True on bf16 values, every primitive
 * rounded to bf16): out = bf16(bf16(a * cdf(a)) * b), cdf = 0.5 (1 + tanh(sqrt(2/pi) (a + 0.044715 a^3))). */
int ixt_op_geglu(ixt_ctx* ctx, const uint16_t* a, const uint16_t* b, uint16_t* out, int64_t n, void* stream);
/* Residual add of the decoder block (llama2.py:127, :157): out = bf16(a + b), n elements. */
int ixt_op_residual(ixt_ctx* ctx, const uint16_t* a, const uint16_t* b, uint16_t* out, int64_t n, void* stream);
/* Embed.__call__ (embeddings.py:131-163): out bf16[R][E] = table bf16[V][E] rows ids[R] (ids clamped to the table, as a jnp gather is). */
int ixt_op_embed(ixt_ctx* ctx, const uint16_t* table, const int32_t* ids, uint16_t* out, int R, int E, int V, void* stream);
/* PagedAttentionOp page writes (paged_attention.py:403-471): k,v bf16[R][nkv][hd] (post-RoPE) into `layer`'s
 * pages of row_slot[r] at row_pos[r] (rows with row_active[r]==0 skipped). */
int ixt_op_kv_write(ixt_ctx* ctx, int layer, const uint16_t* k, const uint16_t* v, const int32_t* row_slot,
                    const int32_t* row_pos, const int32_t* row_active, int R, void* stream);
/* ref_ragged_paged_attention (paged_attention_kernel_v2.py:71-109): q bf16[R][nq][hd] -> out bf16[R][nq][hd];
 * row r attends to positions [0, row_pos[r]] of slot row_slot[r] (kv_len - q_len causal offset rule). */
int ixt_op_paged_attention(ixt_ctx* ctx, int layer, const uint16_t* q, const int32_t* row_slot,
                           const int32_t* row_pos, const int32_t* row_active, int R, uint16_t* out, void* stream);
/* Decoder._apply_output_head logits + greedy sampling (decoders.py:566-587, inference_utils.py:74-75):
 * y bf16[R][emb] (already normed) -> logits f32[R][vocab] (may be NULL), argmax int32[R]. */
int ixt_op_unembed_argmax(ixt_ctx* ctx, const uint16_t* y, int R, float* logits, int32_t* argmax, void* stream);
/* inference_utils.log_prob_of_chosen_token (inference_utils.py:56-63; the engine's return_log_prob outputs, maxengine.py:502-516, 907-926):
 * logits f32[rows][vocab] (the rounded logits ixt_prefill_chunk / ixt_generate / ixt_op_unembed_argmax write), chosen int32[rows] ->
 * out f32[rows] = log_softmax(logits[r])[chosen[r]]. */
int ixt_op_logprob(ixt_ctx* ctx, const float* logits_dev, int rows, const int32_t* chosen_dev, float* out_dev, void* stream);
/* Routed evaluation of THIS context's experts on rows that were sent here (moe.py:855-938 between the two all-to-alls; kernels/megablox/gmm.py:305):
 * x bf16 [n][emb] row-major, expert_ids int32 [n] = the global expert of every row (device).  out bf16 [n][emb]: row i = expert_ids[i]'s MLP of
 * x[i]; a row whose id is negative or belongs to another rank's experts is padding and comes back as zeros.  One launch sequence for all
 * experts (rows are grouped by expert on the device, an expert without rows is never read); no host synchronisation. */
int ixt_op_moe_experts(ixt_ctx* ctx, int layer, const uint16_t* x, const int32_t* expert_ids, int n, uint16_t* out, void* stream);
/* The IXT-FP8-v1 dot on caller-supplied operands (a context created with fp8_weights): x bf16 [R][K], w bf16 [K][N] (row-major) -> out f32 [R][N];
 * w is quantised per column, x per row and 512-k block, exactly as the engine does with a layer's projection kernels (tests). */
int ixt_op_dot_fp8(ixt_ctx* ctx, const uint16_t* x, const uint16_t* w, float* out, int R, int K, int N, void* stream);
/* f32 -> e4m3 (OCP, round to nearest even, saturating at 448) -> f32 on the device, n values: the conversion IXT-FP8-v1 quantises with. */
int ixt_op_fp8_roundtrip(ixt_ctx* ctx, const float* in_dev, float* out_dev, int64_t n, void* stream);
/* Self-check of the f32 -> bf16 rounding every kernel uses (jax's astype(bfloat16): round to nearest even; the reference's dtype choice,
 * base.yml:86): the hardware conversion against the integer formula over all 2^32 inputs -> number of differing non-NaN inputs plus
 * NaN inputs that do not stay NaN (0 expected). */
int ixt_op_cvt_check(ixt_ctx* ctx, int64_t* mismatches_out, void* stream);
/* Gather a slot's K and V of one layer to dense bf16[length][nkv][hd] device arrays (cache export). */
int ixt_kv_gather(ixt_ctx* ctx, int layer, int slot, int length, uint16_t* k_out, uint16_t* v_out, void* stream);
/* RoPE tables (bf16 bit patterns) to HOST arrays [max_target_length][head_dim/2]. */
int ixt_rope_tables(ixt_ctx* ctx, uint16_t* sin_out, uint16_t* cos_out);
/* rope_llama31 contexts: the f32 tables of LLaMARotaryEmbedding to HOST arrays [max_target_length][head_dim/2]. */
int ixt_rope_tables_f32(ixt_ctx* ctx, float* sin_out, float* cos_out);

/* Live measurement hook for bench.py: average duration in microseconds of the named kernel family over the
 * launches made by this ctx since the last ixt_timing_reset, measured with HIP events on the launch stream.
 * Families: "dot_decode" (canonical dot launches, bytes priced by the library), "attn_decode" (one-launch decode attention;
 * bytes_per_launch is 0: the K/V bytes depend on the rows' lengths, the caller prices them). */
int ixt_timing_enable(ixt_ctx* ctx, int enable);
int ixt_timing_reset(ixt_ctx* ctx);
int ixt_timing_get(ixt_ctx* ctx, const char* family, double* avg_us, int64_t* launches, double* bytes_per_launch);


/* Greedy sampling restricted to the union of `nranges` (0..4) half-open column ranges lo_hi[2i], lo_hi[2i+1] (host array);
 * nranges == 0 lifts the restriction.  SURVEY.md section 8(f) item 1: the expanded vocabulary keeps the audio codes in rows
 * 6..104 and 256001..264093 (audio_token_mapping_adjusted.json, generator vocab_expansion/extend_tokenizer.py:29-46); with
 * these two ranges every generated id maps back to an audio code.  Logits are returned unmasked. */
int ixt_set_sampling_ranges(ixt_ctx* ctx, const int32_t* lo_hi, int nranges);

/* Launch-shape knobs of this build (no counterpart in the reference; results never depend on them, tests use them to drive
 * every kernel variant on small configurations):
 *   "attn_fused_min_blocks"  decode attention runs as one launch per layer once rows x kv_heads reaches this (default 128)
 *   "gemm_min_rows", "gemm_min_blocks"  a dot of at least this many rows (default 128) and 128 x 128 blocks (default 150) runs as
 *                            the tiled GEMM kernel (all of K per block, one f32 slice) instead of split-K
 *   "gemm_split_min_blocks"  below gemm_min_blocks: the same tiles split over the 512-k chains (one slice per chain) once there are
 *                            this many (tile, chain) blocks (default 300); otherwise the 64-row split-K kernel
 *   "gemm_prefix_halves"     1: the full-K GEMM form runs as two blocks per tile (first half of the chains as a running total, second half
 *                            chain by chain); 0 (default, measured faster): one block per tile
 *   "unembed_gemm"           1 (default): from gemm_min_rows rows the unembedding runs on the tiled GEMM kernel with the sampling epilogue
 *   "attn_wave_pairs"        1: the one-launch decode attention gives each (row, kv head) pair to one wave, no block barriers;
 *                            0 (default, measured faster): one 256-thread block per pair
 *   "dot_stream_min_blocks"  a dot of > 64 rows and 2..4 chains uses the streaming kernel (one slice) once it has this many
 *                            128-column x 64-row blocks (default: never)
 *   "ring_min_rows"          from this many rows (default 128) every dot of the model runs as the LDS-ring GEMM with fused epilogues
 *                            (QKV + RoPE + page write, residual, SwiGLU, sampling; csrc/ixt_gemm.h); ixt_op_dot uses its f32 epilogue
 *   "ring_cfg"               0 (default): tile chosen per shape; 1: 128 x 128, 2: 128 rows x 64 columns, 3: 64 x 32
 *   "ring_split"             1: ixt_op_dot's ring GEMM runs split over the 512-k chains (one f32 slice per chain)
 *   "ring_wlayout"           1: ixt_op_dot stores its weights in the ring layout ([n/128][k/64][16 fragments]: one sequential stream per block)
 *   "attn_tile"              1 (default): prefill calls of at least "attn_tile_min_rows" rows (default 64) run the tiled attention: 16 (8 past
 *                            ~1500 keys) consecutive rows per block share every K / V load; 0: one block per (row, kv head) as for decode rows
 *   "unembed_tall"           1 (default): the unembedding at 129..256 rows uses 256 x 128 tiles (weights enter the chip once); 0: 128 x 128
 *   "ring_two_per_cu"        1 (default): ring GEMMs (and the unembedding) of 1024 rows and more with at least 320 tiles of 128 x 128 use 32-k chunks
 *                            in 4 slots (64 KiB of LDS: two blocks per CU overlap each other's prologue and epilogue); 0: the 64-k chunk form
 *   "ring_down"              MLP down-projection on the ring path: 1 ring GEMM + residual epilogue, 0 split-K tiles + slice-sum kernel,
 *                            -1 (default) by shape */
int ixt_set_option(ixt_ctx* ctx, const char* name, int value);

/* Hash of the sources this library was built from (maxtext-indextts2_amd/build.py); the Python binding refuses a stale binary. */
const char* ixt_build_id(void);

#ifdef __cplusplus
}
#endif
#endif /* IXT_H_ */
