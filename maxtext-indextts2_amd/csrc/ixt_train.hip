// ixt_train.hip -- C ABI of the training step (include/ixt_train.h) over ixt_train_kernels.h.  Host code: buffer ownership, launch order.
// Reference: MaxText/train.py:255-472 (loss_fn, train_step), layers/* in MODEL_MODE_TRAIN, max_utils.py:563-654, maxtext_utils.py:825-846,
// optimizers.py:28-38.  Parameters, gradients and AdamW state live in FLAT f32 device buffers owned by the caller (so that the data-parallel
// exchange -- RCCL reduce-scatter + all-gather through torch.distributed -- works on them directly); this file owns activations and workspaces.
#include "../../include/ixt_train.h"
#include "ixt_kernels.h"
#include "ixt_train_kernels.h"
// (compiled as part of ixt_api.hip: one translation unit, one libixt.so)

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

static thread_local char t_err[512] = "";
static int t_set_err(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(t_err, sizeof(t_err), fmt, ap);
  va_end(ap);
  return code;
}
extern "C" const char* ixt_train_last_error(void) { return t_err; }
#define TCHK(expr)                                                                                                          \
  do {                                                                                                                      \
    hipError_t e_ = (expr);                                                                                                 \
    if (e_ != hipSuccess) return t_set_err(IXT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)
#define TLAUNCH() TCHK(hipGetLastError())
#define TTRY(x) do { int rc_ = (x); if (rc_ != IXT_OK) return rc_; } while (0)

struct ParamInfo { std::string name; long offset, numel; };

struct LayerAct {   // saved for the backward pass
  bf16_t *h_in, *xn1, *q, *k, *v, *P, *ao, *h_mid, *xn2, *a, *b, *act;
  float *r1, *r2;
};

struct ixt_trainer {
  ixt_train_config cfg;
  int E, H, D, F, V, L, B, S, T, QN, Tc;
  int KVH, G, KN;           // grouped-query attention: kv heads, query heads per kv head, KVH * D
  std::vector<ParamInfo> params;
  long total = 0;
  // caller-owned flat buffers
  float *p = nullptr, *g = nullptr, *mu = nullptr, *nu = nullptr;
  bf16_t* pb = nullptr;   // bf16 copies of the parameters (owned here)
  // offsets
  long o_emb, o_fnorm, o_logits;
  struct LOff { long ln1, wq, wk, wv, wo, ln2, wi0, wi1, wout; };
  std::vector<LOff> lo;
  std::vector<LayerAct> act;
  bf16_t *h_final = nullptr, *xn_f = nullptr, *rope_sin = nullptr, *rope_cos = nullptr;
  float* r_f = nullptr;
  // workspaces
  float *scores = nullptr, *dh = nullptr, *dxn = nullptr, *logits = nullptr, *row_loss = nullptr, *scal = nullptr, *part = nullptr;
  bf16_t *tmp_bf = nullptr, *dS = nullptr, *d_ao = nullptr, *dq = nullptr, *dk = nullptr, *dv = nullptr, *dact = nullptr, *da = nullptr, *db = nullptr, *dlog = nullptr,
         *dy_bf = nullptr;
  float *dkf = nullptr, *dvf = nullptr;   // grouped-query attention: dK / dV summed over a group's query heads in f32 before they round to bf16
  std::vector<void*> allocs;
};

template <typename T_>
static int tmalloc(ixt_trainer* t, T_** p, size_t n) {
  void* q = nullptr;
  TCHK(hipMalloc(&q, n * sizeof(T_) + 256));
  t->allocs.push_back(q);
  *p = (T_*)q;
  return IXT_OK;
}

static void add_param(ixt_trainer* t, const std::string& name, long numel, long* off) {
  *off = t->total;
  t->params.push_back({name, t->total, numel});
  t->total += (numel + 63) / 64 * 64;   // 256-byte aligned segments
}

extern "C" int ixt_train_create(const ixt_train_config* c, ixt_trainer** out) {
  if (!c || !out) return t_set_err(IXT_ERR_INVALID, "null argument");
  if (c->num_kv_heads <= 0 || c->num_query_heads % c->num_kv_heads) return t_set_err(IXT_ERR_INVALID, "train step: num_query_heads %% num_kv_heads != 0");
  if (c->head_dim % 8 || c->emb_dim % 8 || c->mlp_dim % 8 || c->vocab_size % 8 || c->seq_len % 8)
    return t_set_err(IXT_ERR_UNSUPPORTED, "train step: every dimension must be a multiple of 8");
  if (c->emb_dim > 256 * TN_FPT) return t_set_err(IXT_ERR_UNSUPPORTED, "train step: emb_dim %d above %d (k_t_rmsnorm_bwd keeps a row block's scale gradients in registers)", c->emb_dim, 256 * TN_FPT);
  if (c->batch <= 0 || c->seq_len <= 0 || c->num_decoder_layers <= 0) return t_set_err(IXT_ERR_INVALID, "bad sizes");
  ixt_trainer* t = new ixt_trainer();
  t->cfg = *c;
  t->E = c->emb_dim; t->H = c->num_query_heads; t->D = c->head_dim; t->F = c->mlp_dim; t->V = c->vocab_size; t->L = c->num_decoder_layers;
  t->B = c->batch; t->S = c->seq_len; t->T = t->B * t->S; t->QN = t->H * t->D;
  t->KVH = c->num_kv_heads; t->G = t->H / t->KVH; t->KN = t->KVH * t->D;
  t->Tc = c->logits_chunk > 0 ? c->logits_chunk : 2048;
  if (t->Tc > t->T) t->Tc = t->T;
  const long E = t->E, F = t->F, V = t->V, QN = t->QN, KN = t->KN;
  add_param(t, "params/token_embedder/embedding", V * E, &t->o_emb);
  t->lo.resize(t->L);
  for (int i = 0; i < t->L; ++i) {
    const std::string p = "params/decoder/layers_" + std::to_string(i) + "/";
    add_param(t, p + "pre_self_attention_layer_norm/scale", E, &t->lo[i].ln1);
    add_param(t, p + "self_attention/query/kernel", E * QN, &t->lo[i].wq);
    add_param(t, p + "self_attention/key/kernel", E * KN, &t->lo[i].wk);
    add_param(t, p + "self_attention/value/kernel", E * KN, &t->lo[i].wv);
    add_param(t, p + "self_attention/out/kernel", QN * E, &t->lo[i].wo);
    add_param(t, p + "mlp/mlp_layer_norm/scale", E, &t->lo[i].ln2);
    add_param(t, p + "mlp/wi_0/kernel", E * F, &t->lo[i].wi0);
    add_param(t, p + "mlp/wi_1/kernel", E * F, &t->lo[i].wi1);
    add_param(t, p + "mlp/wo/kernel", F * E, &t->lo[i].wout);
  }
  add_param(t, "params/decoder/decoder_norm/scale", E, &t->o_fnorm);
  add_param(t, "params/decoder/logits_dense/kernel", E * V, &t->o_logits);
  *out = nullptr;
  int rc = IXT_OK;
  auto fail = [&](int code) { ixt_train_destroy(t); return code; };
#define A(expr) do { rc = (expr); if (rc != IXT_OK) return fail(rc); } while (0)
  const size_t T = t->T, S = t->S, BH = (size_t)t->B * t->H;
  A(tmalloc(t, &t->pb, (size_t)t->total));
  t->act.resize(t->L);
  for (int i = 0; i < t->L; ++i) {
    LayerAct& a = t->act[i];
    A(tmalloc(t, &a.h_in, T * E)); A(tmalloc(t, &a.xn1, T * E)); A(tmalloc(t, &a.q, T * QN)); A(tmalloc(t, &a.k, T * KN)); A(tmalloc(t, &a.v, T * KN));
    A(tmalloc(t, &a.P, BH * S * S)); A(tmalloc(t, &a.ao, T * QN)); A(tmalloc(t, &a.h_mid, T * E)); A(tmalloc(t, &a.xn2, T * E));
    A(tmalloc(t, &a.a, T * F)); A(tmalloc(t, &a.b, T * F)); A(tmalloc(t, &a.act, T * F)); A(tmalloc(t, &a.r1, T)); A(tmalloc(t, &a.r2, T));
  }
  A(tmalloc(t, &t->h_final, T * E)); A(tmalloc(t, &t->xn_f, T * E)); A(tmalloc(t, &t->r_f, T));
  A(tmalloc(t, &t->scores, BH * S * S)); A(tmalloc(t, &t->dh, T * E)); A(tmalloc(t, &t->dxn, T * E));
  A(tmalloc(t, &t->logits, (size_t)t->Tc * V)); A(tmalloc(t, &t->dlog, (size_t)t->Tc * V)); A(tmalloc(t, &t->row_loss, T)); A(tmalloc(t, &t->scal, (size_t)16));
  A(tmalloc(t, &t->part, (size_t)4096));
  const size_t big = T * (size_t)(F > QN ? (F > E ? F : E) : (QN > E ? QN : E));
  A(tmalloc(t, &t->tmp_bf, big)); A(tmalloc(t, &t->dS, BH * S * S)); A(tmalloc(t, &t->d_ao, T * QN)); A(tmalloc(t, &t->dq, T * QN)); A(tmalloc(t, &t->dk, T * KN));
  A(tmalloc(t, &t->dv, T * KN));
  if (t->G > 1) { A(tmalloc(t, &t->dkf, T * KN)); A(tmalloc(t, &t->dvf, T * KN)); } A(tmalloc(t, &t->dact, T * F)); A(tmalloc(t, &t->da, T * F)); A(tmalloc(t, &t->db, T * F)); A(tmalloc(t, &t->dy_bf, T * E));
#undef A
  // RoPE tables: the same host recipe as the decode path (embeddings.py:270-307)
  {
    const int half = t->D / 2;
    std::vector<uint16_t> sn((size_t)S * half), cs((size_t)S * half);
    auto f2bf_h = [](float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); };
    for (int i = 0; i < half; ++i) {
      const float frac = (float)(2 * i) / (float)t->D;
      const float ts = (float)((double)c->rope_min_timescale * pow((double)c->rope_max_timescale / (double)c->rope_min_timescale, (double)frac));
      for (size_t p = 0; p < S; ++p) {
        const float ang = (float)p / ts;
        sn[p * half + i] = f2bf_h((float)sin((double)ang));
        cs[p * half + i] = f2bf_h((float)cos((double)ang));
      }
    }
    if (tmalloc(t, &t->rope_sin, sn.size()) != IXT_OK || tmalloc(t, &t->rope_cos, cs.size()) != IXT_OK ||
        hipMemcpy(t->rope_sin, sn.data(), sn.size() * 2, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(t->rope_cos, cs.data(), cs.size() * 2, hipMemcpyHostToDevice) != hipSuccess)
      return fail(t_set_err(IXT_ERR_HIP, "rope table upload failed"));
  }
  *out = t;
  return IXT_OK;
}

extern "C" int ixt_train_destroy(ixt_trainer* t) {
  if (!t) return IXT_OK;
  (void)hipDeviceSynchronize();
  for (void* p : t->allocs) (void)hipFree(p);
  delete t;
  return IXT_OK;
}

extern "C" int64_t ixt_train_total_numel(ixt_trainer* t) { return t ? t->total : 0; }
extern "C" int ixt_train_param_count(ixt_trainer* t) { return t ? (int)t->params.size() : 0; }
extern "C" int ixt_train_param_info(ixt_trainer* t, int index, char* name_out, int name_cap, int64_t* offset, int64_t* numel) {
  if (!t || index < 0 || index >= (int)t->params.size() || !name_out || !offset || !numel) return t_set_err(IXT_ERR_INVALID, "bad argument");
  snprintf(name_out, name_cap, "%s", t->params[index].name.c_str());
  *offset = t->params[index].offset;
  *numel = t->params[index].numel;
  return IXT_OK;
}
extern "C" int ixt_train_bind(ixt_trainer* t, float* params, float* grads, float* adam_mu, float* adam_nu) {
  if (!t || !params || !grads || !adam_mu || !adam_nu) return t_set_err(IXT_ERR_INVALID, "null argument");
  t->p = params; t->g = grads; t->mu = adam_mu; t->nu = adam_nu;
  return IXT_OK;
}
extern "C" int ixt_train_refresh_bf16(ixt_trainer* t, void* stream) {
  if (!t || !t->p) return t_set_err(IXT_ERR_INVALID, "bind the parameter buffers first");
  hipLaunchKernelGGL(k_t_cast_bf16, dim3((unsigned)((t->total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, t->p, t->pb, t->total);
  TLAUNCH();
  return IXT_OK;
}

// ---------------------------------------------------------------------------------------------------------------- launches
// Scratch for split reductions (set by the step: a buffer that is idle while the layers' backward GEMMs run).
static float* g_splitk_buf = nullptr;
static size_t g_splitk_floats = 0;
static int gemm(hipStream_t st, const bf16_t* A, long sam, long sak, const bf16_t* B, long sbk, long sbn, void* C, long scm, int M, int N, int K, int out_bf16,
                int accumulate, int nb1 = 1, int nb2 = 1, long sab1 = 0, long sab2 = 0, long sbb1 = 0, long sbb2 = 0, long scb1 = 0, long scb2 = 0) {
  // A weight gradient of few output tiles over a long reduction (2048 x 2048 over 16384 tokens: 256 tiles, one block per CU, 287 TFLOP/s): the
  // reduction is cut in four batch parts whose products are added up in order (k_t_sum_parts).
  constexpr int KS = 4;
  if (!out_bf16 && !accumulate && nb1 * nb2 == 1 && scm == N && K >= 2048 && K % (KS * TG_BK) == 0 && (long)((N + 127) / 128) * ((M + 127) / 128) <= 256 &&
      g_splitk_buf != nullptr && (size_t)KS * M * N <= g_splitk_floats) {
    const int rc = gemm(st, A, sam, sak, B, sbk, sbn, g_splitk_buf, N, M, N, K / KS, 0, 0, KS, 1, (long)(K / KS) * sak, 0, (long)(K / KS) * sbk, 0, (long)M * N, 0);
    if (rc != IXT_OK) return rc;
    hipLaunchKernelGGL(k_t_sum_parts, dim3((unsigned)(((long)M * N + 255) / 256)), dim3(256), 0, st, g_splitk_buf, KS, (long)M * N, reinterpret_cast<float*>(C));
    TLAUNCH();
    return IXT_OK;
  }
  TGemm g;
  g.A = A; g.sam = sam; g.sak = sak; g.sab1 = sab1; g.sab2 = sab2;
  g.B = B; g.sbk = sbk; g.sbn = sbn; g.sbb1 = sbb1; g.sbb2 = sbb2;
  g.C = C; g.scm = scm; g.scb1 = scb1; g.scb2 = scb2;
  g.M = M; g.N = N; g.K = K; g.nb2 = nb2; g.out_bf16 = out_bf16; g.accumulate = accumulate;
  if ((sam != 1 && sak != 1) || (sbk != 1 && sbn != 1)) return t_set_err(IXT_ERR_INVALID, "gemm: an operand has no unit stride");
  const dim3 grid((N + 127) / 128, (M + 127) / 128, nb1 * nb2);
  const bool ta = sak != 1, tb = sbk != 1;     // the operand's rows, not k, are contiguous: its tile is read transposed from LDS
  if (ta && tb) hipLaunchKernelGGL((k_tgemm<true, true>), grid, dim3(256), 0, st, g);
  else if (ta) hipLaunchKernelGGL((k_tgemm<true, false>), grid, dim3(256), 0, st, g);
  else if (tb) hipLaunchKernelGGL((k_tgemm<false, true>), grid, dim3(256), 0, st, g);
  else hipLaunchKernelGGL((k_tgemm<false, false>), grid, dim3(256), 0, st, g);
  TLAUNCH();
  return IXT_OK;
}
#define EW(n) dim3((unsigned)(((long)(n) + 255) / 256)), dim3(256), 0, st

extern "C" int ixt_train_forward_backward(ixt_trainer* t, const int32_t* inputs, const int32_t* positions, const int32_t* segmentation,
                                          const int32_t* targets, const int32_t* targets_segmentation, float* loss_and_weights_dev, void* stream) {
  if (!t || !inputs || !positions || !segmentation || !targets || !targets_segmentation || !loss_and_weights_dev)
    return t_set_err(IXT_ERR_INVALID, "null argument");
  if (!t->p) return t_set_err(IXT_ERR_INVALID, "bind the parameter buffers first");
  g_splitk_buf = nullptr; g_splitk_floats = 0;
  hipStream_t st = (hipStream_t)stream;
  const int E = t->E, H = t->H, D = t->D, F = t->F, V = t->V, S = t->S, T = t->T, QN = t->QN, B = t->B, KVH = t->KVH, G = t->G, KN = t->KN;
  const long HD = (long)H * D;
  const bf16_t* pb = t->pb;
  const float eps = t->cfg.normalization_layer_epsilon;
  TCHK(hipMemsetAsync(t->g, 0, sizeof(float) * t->total, st));
  // ------------------------------------------------------------------ forward (layers/* in train mode)
  bf16_t* h = t->act[0].h_in;
  hipLaunchKernelGGL(k_embed, dim3((E / 8 + 255) / 256, T), dim3(256), 0, st, pb + t->o_emb, inputs, h, E, V);
  TLAUNCH();
  for (int l = 0; l < t->L; ++l) {
    LayerAct& a = t->act[l];
    const ixt_trainer::LOff& o = t->lo[l];
    hipLaunchKernelGGL(k_t_rmsnorm_fwd, dim3(T), dim3(256), 0, st, a.h_in, pb + o.ln1, a.xn1, a.r1, E, eps);
    TTRY(gemm(st, a.xn1, E, 1, pb + o.wq, QN, 1, a.q, QN, T, QN, E, 1, 0));
    TTRY(gemm(st, a.xn1, E, 1, pb + o.wk, KN, 1, a.k, KN, T, KN, E, 1, 0));
    TTRY(gemm(st, a.xn1, E, 1, pb + o.wv, KN, 1, a.v, KN, T, KN, E, 1, 0));
    hipLaunchKernelGGL((k_t_rope<false>), dim3(T, H), dim3(D / 2), 0, st, a.q, positions, t->rope_sin, t->rope_cos, H, D);
    hipLaunchKernelGGL((k_t_rope<false>), dim3(T, KVH), dim3(D / 2), 0, st, a.k, positions, t->rope_sin, t->rope_cos, KVH, D);
    // scores[b,h] = Q_h K_(h / G)^T (sm_scale 1: the query kernel carries 1/sqrt(d), attentions.py:1900-1904).  Grouped-query attention
    // (attentions.py:1206-1271: query heads reshaped to [kv heads, G]): query head kvh G + g of every kv head in launch g, batches (b, kvh).
    for (int g = 0; g < G; ++g)
      TTRY(gemm(st, a.q + (long)g * D, HD, 1, a.k, 1, KN, t->scores + (long)g * S * S, S, S, S, D, 0, 0, B, KVH, (long)S * HD, (long)G * D, (long)S * KN, D,
                (long)H * S * S, (long)G * S * S));
    if (S <= 256 * TS_MAXV) hipLaunchKernelGGL(k_t_softmax_fwd_reg, dim3(S, B * H), dim3(256), 0, st, t->scores, segmentation, a.P, S, H);
    else hipLaunchKernelGGL(k_t_softmax_fwd, dim3(S, B * H), dim3(256), 0, st, t->scores, segmentation, a.P, S, H);
    for (int g = 0; g < G; ++g)
      TTRY(gemm(st, a.P + (long)g * S * S, S, 1, a.v, KN, 1, a.ao + (long)g * D, HD, S, D, S, 1, 0, B, KVH, (long)H * S * S, (long)G * S * S, (long)S * KN, D,
                (long)S * HD, (long)G * D));
    TTRY(gemm(st, a.ao, QN, 1, pb + o.wo, E, 1, t->tmp_bf, E, T, E, QN, 1, 0));
    hipLaunchKernelGGL(k_t_add_bf16, EW((long)T * E), a.h_in, t->tmp_bf, a.h_mid, (long)T * E);
    hipLaunchKernelGGL(k_t_rmsnorm_fwd, dim3(T), dim3(256), 0, st, a.h_mid, pb + o.ln2, a.xn2, a.r2, E, eps);
    TTRY(gemm(st, a.xn2, E, 1, pb + o.wi0, F, 1, a.a, F, T, F, E, 1, 0));
    TTRY(gemm(st, a.xn2, E, 1, pb + o.wi1, F, 1, a.b, F, T, F, E, 1, 0));
    hipLaunchKernelGGL(k_t_swiglu_fwd, EW((long)T * F), a.a, a.b, a.act, (long)T * F);
    TTRY(gemm(st, a.act, F, 1, pb + o.wout, E, 1, t->tmp_bf, E, T, E, F, 1, 0));
    bf16_t* h_next = l + 1 < t->L ? t->act[l + 1].h_in : t->h_final;
    hipLaunchKernelGGL(k_t_add_bf16, EW((long)T * E), a.h_mid, t->tmp_bf, h_next, (long)T * E);
    TLAUNCH();
  }
  hipLaunchKernelGGL(k_t_rmsnorm_fwd, dim3(T), dim3(256), 0, st, t->h_final, pb + t->o_fnorm, t->xn_f, t->r_f, E, eps);
  g_splitk_buf = nullptr; g_splitk_floats = 0;          // (the logits buffer is in use below)
  // ------------------------------------------------------------------ logits, loss and their gradients, a chunk of rows at a time
  TCHK(hipMemsetAsync(t->dxn, 0, sizeof(float) * (size_t)T * E, st));
  for (int t0 = 0; t0 < T; t0 += t->Tc) {
    const int n = T - t0 < t->Tc ? T - t0 : t->Tc;
    TTRY(gemm(st, t->xn_f + (long)t0 * E, E, 1, pb + t->o_logits, V, 1, t->logits, V, n, V, E, 0, 0));
    hipLaunchKernelGGL(k_t_xent, dim3(n), dim3(256), 0, st, t->logits, targets + t0, targets_segmentation + t0, t->dlog, t->row_loss + t0, V);
    // d xn_f = dlogits . W^T ; dW += xn_f^T . dlogits
    // (n x E outputs are only 256 tiles of 128 x 128 over K = V: the reduction is cut in KS parts -- batch dimension of the GEMM -- whose products
    //  land in the idle score buffer and are added up in order; one part per CU took 8.8 ms per chunk at 253 TFLOP/s)
    const int KS = (V % (4 * TG_BK) == 0 && V >= 65536 && (size_t)4 * n * E <= (size_t)B * H * S * S) ? 4 : 1;
    if (KS > 1) {
      TTRY(gemm(st, t->dlog, V, 1, pb + t->o_logits, 1, V, t->scores, E, n, E, V / KS, 0, 0, KS, 1, V / KS, 0, V / KS, 0, (long)n * E, 0));
      hipLaunchKernelGGL(k_t_sum_parts, EW((long)n * E), t->scores, KS, (long)n * E, t->dxn + (long)t0 * E);
      TLAUNCH();
    } else {
      TTRY(gemm(st, t->dlog, V, 1, pb + t->o_logits, 1, V, t->dxn + (long)t0 * E, E, n, E, V, 0, 0));
    }
    TTRY(gemm(st, t->xn_f + (long)t0 * E, 1, E, t->dlog, V, 1, t->g + t->o_logits, V, E, V, n, 0, 1));
  }
  hipLaunchKernelGGL(k_t_reduce, dim3(1), dim3(256), 0, st, t->row_loss, (long)T, 0, loss_and_weights_dev, 0);
  // total_weights = number of target positions with a non-zero segment id (train.py:309)
  hipLaunchKernelGGL(k_t_nonzero, EW(T), targets_segmentation, t->row_loss, (long)T);          // row_loss is free again after the reduction above
  hipLaunchKernelGGL(k_t_reduce, dim3(1), dim3(256), 0, st, t->row_loss, (long)T, 0, loss_and_weights_dev + 1, 0);
  TLAUNCH();
  // ------------------------------------------------------------------ backward
  g_splitk_buf = t->logits; g_splitk_floats = (size_t)t->Tc * V;   // idle from here on: scratch of the split weight-gradient GEMMs
  TCHK(hipMemsetAsync(t->dh, 0, sizeof(float) * (size_t)T * E, st));
  hipLaunchKernelGGL(k_t_rmsnorm_bwd, dim3((T + TN_RPB - 1) / TN_RPB), dim3(256), 0, st, t->dxn, t->h_final, pb + t->o_fnorm, t->r_f, t->dh, t->g + t->o_fnorm, E, (long)T);
  for (int l = t->L - 1; l >= 0; --l) {
    LayerAct& a = t->act[l];
    const ixt_trainer::LOff& o = t->lo[l];
    // MLP: h_out = h_mid + wo(act)
    hipLaunchKernelGGL(k_t_f32_to_bf16, EW((long)T * E), t->dh, t->dy_bf, (long)T * E);
    TTRY(gemm(st, t->dy_bf, E, 1, pb + o.wout, 1, E, t->dact, F, T, F, E, 1, 0));                       // d act = dy . Wout^T
    TTRY(gemm(st, a.act, 1, F, t->dy_bf, E, 1, t->g + o.wout, E, F, E, T, 0, 0));                        // d Wout = act^T . dy
    hipLaunchKernelGGL(k_t_swiglu_bwd, EW((long)T * F), t->dact, a.a, a.b, t->da, t->db, (long)T * F);
    TTRY(gemm(st, t->da, F, 1, pb + o.wi0, 1, F, t->dxn, E, T, E, F, 0, 0));                             // d xn2 = da . Wi0^T + db . Wi1^T
    TTRY(gemm(st, t->db, F, 1, pb + o.wi1, 1, F, t->dxn, E, T, E, F, 0, 1));
    TTRY(gemm(st, a.xn2, 1, E, t->da, F, 1, t->g + o.wi0, F, E, F, T, 0, 0));
    TTRY(gemm(st, a.xn2, 1, E, t->db, F, 1, t->g + o.wi1, F, E, F, T, 0, 0));
    hipLaunchKernelGGL(k_t_rmsnorm_bwd, dim3((T + TN_RPB - 1) / TN_RPB), dim3(256), 0, st, t->dxn, a.h_mid, pb + o.ln2, a.r2, t->dh, t->g + o.ln2, E, (long)T);
    // attention: h_mid = h_in + out(ao)
    hipLaunchKernelGGL(k_t_f32_to_bf16, EW((long)T * E), t->dh, t->dy_bf, (long)T * E);
    TTRY(gemm(st, t->dy_bf, E, 1, pb + o.wo, 1, E, t->d_ao, QN, T, QN, E, 1, 0));                        // d ao = dy . Wo^T
    TTRY(gemm(st, a.ao, 1, QN, t->dy_bf, E, 1, t->g + o.wo, E, QN, E, T, 0, 0));                          // d Wo = ao^T . dy
    // dP = d_ao V^T (f32) ; dV = P^T d_ao ; dS = P o (dP - rowsum(dP o P)) ; dQ = dS K ; dK = dS^T Q.  With G query heads per kv head, dV and dK
    // are the sums over the group: accumulated in f32 over the G launches, rounded to bf16 once.
    for (int g = 0; g < G; ++g)
      TTRY(gemm(st, t->d_ao + (long)g * D, HD, 1, a.v, 1, KN, t->scores + (long)g * S * S, S, S, S, D, 0, 0, B, KVH, (long)S * HD, (long)G * D, (long)S * KN, D,
                (long)H * S * S, (long)G * S * S));
    if (G == 1) {
      TTRY(gemm(st, a.P, 1, S, t->d_ao, HD, 1, t->dv, HD, S, D, S, 1, 0, B, H, (long)H * S * S, (long)S * S, (long)S * HD, D, (long)S * HD, D));
    } else {
      for (int g = 0; g < G; ++g)
        TTRY(gemm(st, a.P + (long)g * S * S, 1, S, t->d_ao + (long)g * D, HD, 1, t->dvf, KN, S, D, S, 0, g > 0, B, KVH, (long)H * S * S, (long)G * S * S,
                  (long)S * HD, (long)G * D, (long)S * KN, D));
      hipLaunchKernelGGL(k_t_f32_to_bf16, EW((long)T * KN), t->dvf, t->dv, (long)T * KN);
    }
    if (S <= 256 * TS_MAXV) hipLaunchKernelGGL(k_t_softmax_bwd_reg, dim3(S, B * H), dim3(256), 0, st, t->scores, a.P, t->dS, S);
    else hipLaunchKernelGGL(k_t_softmax_bwd, dim3(S, B * H), dim3(256), 0, st, t->scores, a.P, t->dS, S);
    for (int g = 0; g < G; ++g)
      TTRY(gemm(st, t->dS + (long)g * S * S, S, 1, a.k, KN, 1, t->dq + (long)g * D, HD, S, D, S, 1, 0, B, KVH, (long)H * S * S, (long)G * S * S, (long)S * KN, D,
                (long)S * HD, (long)G * D));
    if (G == 1) {
      TTRY(gemm(st, t->dS, 1, S, a.q, HD, 1, t->dk, HD, S, D, S, 1, 0, B, H, (long)H * S * S, (long)S * S, (long)S * HD, D, (long)S * HD, D));
    } else {
      for (int g = 0; g < G; ++g)
        TTRY(gemm(st, t->dS + (long)g * S * S, 1, S, a.q + (long)g * D, HD, 1, t->dkf, KN, S, D, S, 0, g > 0, B, KVH, (long)H * S * S, (long)G * S * S,
                  (long)S * HD, (long)G * D, (long)S * KN, D));
      hipLaunchKernelGGL(k_t_f32_to_bf16, EW((long)T * KN), t->dkf, t->dk, (long)T * KN);
    }
    hipLaunchKernelGGL((k_t_rope<true>), dim3(T, H), dim3(D / 2), 0, st, t->dq, positions, t->rope_sin, t->rope_cos, H, D);
    hipLaunchKernelGGL((k_t_rope<true>), dim3(T, KVH), dim3(D / 2), 0, st, t->dk, positions, t->rope_sin, t->rope_cos, KVH, D);
    TTRY(gemm(st, t->dq, QN, 1, pb + o.wq, 1, QN, t->dxn, E, T, E, QN, 0, 0));                           // d xn1 = dq Wq^T + dk Wk^T + dv Wv^T
    TTRY(gemm(st, t->dk, KN, 1, pb + o.wk, 1, KN, t->dxn, E, T, E, KN, 0, 1));
    TTRY(gemm(st, t->dv, KN, 1, pb + o.wv, 1, KN, t->dxn, E, T, E, KN, 0, 1));
    TTRY(gemm(st, a.xn1, 1, E, t->dq, QN, 1, t->g + o.wq, QN, E, QN, T, 0, 0));
    TTRY(gemm(st, a.xn1, 1, E, t->dk, KN, 1, t->g + o.wk, KN, E, KN, T, 0, 0));
    TTRY(gemm(st, a.xn1, 1, E, t->dv, KN, 1, t->g + o.wv, KN, E, KN, T, 0, 0));
    hipLaunchKernelGGL(k_t_rmsnorm_bwd, dim3((T + TN_RPB - 1) / TN_RPB), dim3(256), 0, st, t->dxn, a.h_in, pb + o.ln1, a.r1, t->dh, t->g + o.ln1, E, (long)T);
    TLAUNCH();
  }
  hipLaunchKernelGGL(k_t_embed_bwd, dim3((E + 255) / 256, T), dim3(256), 0, st, t->dh, inputs, t->g + t->o_emb, E, V);
  TLAUNCH();
  return IXT_OK;
}

extern "C" int ixt_train_apply(ixt_trainer* t, float total_weights, float clip_threshold, float lr, float b1, float b2, float eps, float eps_root,
                               float weight_decay, int count, float* norms_out_dev, void* stream) {
  if (!t || !t->p || !norms_out_dev || count < 1) return t_set_err(IXT_ERR_INVALID, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int nblk = 2048;
  hipLaunchKernelGGL(k_t_sumsq_partial, dim3(nblk), dim3(256), 0, st, t->g, t->total, t->part);
  hipLaunchKernelGGL(k_t_reduce, dim3(1), dim3(256), 0, st, t->part, (long)nblk, 0, t->scal, 0);
  hipLaunchKernelGGL(k_t_clip_scale, dim3(1), dim3(1), 0, st, t->scal, total_weights + 1e-8f, clip_threshold, t->scal + 1, norms_out_dev);   // EPS = 1e-8 (globals.py:20)
  AdamArgs a;
  a.lr = lr; a.b1 = b1; a.b2 = b2; a.eps = eps; a.eps_root = eps_root; a.wd = weight_decay;
  a.bc1 = 1.0f - powf(b1, (float)count);
  a.bc2 = 1.0f - powf(b2, (float)count);
  hipLaunchKernelGGL(k_t_adamw, EW(t->total), t->p, t->g, t->mu, t->nu, t->pb, t->scal + 1, t->total, a);
  TLAUNCH();
  return IXT_OK;
}
