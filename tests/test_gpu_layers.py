"""GPU parity of the `layers.*` / `inference.*` module surface (the reference's layer names over the op-level C ABI): each module against the
oracle, and a whole Transformer assembled from the modules -- prefill of one prompt, then decode steps through PagedAttentionOp and the
device PageManager -- against the oracle's logits, bit for bit (the same arithmetic as the engine's fused kernels, one op at a time)."""
import ctypes
import importlib

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def surface():
  import torch
  runtime = importlib.import_module("maxtext-indextts2_amd.runtime")
  layers = importlib.import_module("maxtext-indextts2_amd.layers")
  ct = importlib.import_module("maxtext-indextts2_amd.common_types")
  cfg = H.make_config(per_device_batch_size=3.0, max_prefill_predict_length=32, max_target_length=64, pagedattn_num_pages=64)
  w = H.synth.synth_weights(cfg, 21)
  ctx = runtime.set_context(runtime.OpContext(cfg))
  yield {"cfg": cfg, "w": w, "ctx": ctx, "layers": layers, "ct": ct, "torch": torch, "runtime": runtime, "orc": H.make_oracle(cfg, w)}
  runtime._default = None


def test_rmsnorm_dense_mlp_modules(surface):
  torch, L, cfg, O = surface["torch"], surface["layers"], surface["cfg"], importlib.import_module("oracle.oracle")
  rng = np.random.default_rng(3)
  E, F = cfg.emb_dim, cfg.mlp_dim
  x = H.rand_bf16(rng, (2, 5, E), 2.0)
  norm = L.normalizations.rms_norm(num_features=E, epsilon=cfg.normalization_layer_epsilon, name="n")
  norm.scale = H.dev_bf16(H.rand_bf16(rng, (E,)))
  ref = np.zeros((10, E), np.uint16)
  for r in range(10):
    O.lib().orc_rmsnorm(O._p(x.reshape(10, E)[r], O._u16p), O._p(H.bits(norm.scale), O._u16p), O._p(ref[r], O._u16p), E, ctypes.c_float(cfg.normalization_layer_epsilon))
  assert np.array_equal(H.bits(norm(H.dev_bf16(x))).reshape(10, E), ref)
  # DenseGeneral over (heads, head_dim), the out-projection form (axis (-2, -1))
  dg = L.linears.DenseGeneral((cfg.num_query_heads, cfg.head_dim), E, axis=(-2, -1), name="out")
  wk = H.rand_bf16(rng, (cfg.num_query_heads, cfg.head_dim, E), 0.05)
  dg.kernel = H.dev_bf16(wk)
  a = H.rand_bf16(rng, (1, 7, cfg.num_query_heads, cfg.head_dim))
  K = cfg.num_query_heads * cfg.head_dim
  t = np.zeros((7, E), np.float32)
  O.lib().orc_dot_rows(O._p(a.reshape(7, K), O._u16p), K, 7, O._p(wk.reshape(K, E), O._u16p), K, E, O._p(t, O._f32p), E)
  assert np.array_equal(H.bits(dg(H.dev_bf16(a))).reshape(7, E), O.f32_to_bf16(t))
  # MlpBlock with its pre-norm (linears.py:430-476)
  mlp = L.linears.mlp_block(config=cfg, in_features=E, intermediate_dim=F, activations=cfg.mlp_activations, use_pre_norm=True, name="mlp")
  w0, w1, wo = H.rand_bf16(rng, (E, F), 0.06), H.rand_bf16(rng, (E, F), 0.06), H.rand_bf16(rng, (F, E), 0.04)
  mlp.wi_0.kernel, mlp.wi_1.kernel, mlp.wo.kernel = H.dev_bf16(w0), H.dev_bf16(w1), H.dev_bf16(wo)
  xin = x.reshape(10, E)
  xn = np.zeros_like(xin)
  one = H.synth.f32_to_bf16_bits(np.ones((E,), np.float32))
  for r in range(10):
    O.lib().orc_rmsnorm(O._p(xin[r], O._u16p), O._p(one, O._u16p), O._p(xn[r], O._u16p), E, ctypes.c_float(cfg.normalization_layer_epsilon))
  def dot_bf16(xx, ww, Kd, Nd):
    t = np.zeros((xx.shape[0], Nd), np.float32)
    O.lib().orc_dot_rows(O._p(np.ascontiguousarray(xx), O._u16p), Kd, xx.shape[0], O._p(ww, O._u16p), Kd, Nd, O._p(t, O._f32p), Nd)
    return O.f32_to_bf16(t)
  ga, gb = dot_bf16(xn, w0, E, F), dot_bf16(xn, w1, E, F)
  act = np.zeros_like(ga)
  O.lib().orc_swiglu(O._p(ga.reshape(-1), O._u16p), O._p(gb.reshape(-1), O._u16p), O._p(act.reshape(-1), O._u16p), ga.size)
  assert np.array_equal(H.bits(mlp(H.dev_bf16(x))).reshape(10, E), dot_bf16(act, wo, F, E))


def test_embed_and_rope_modules(surface):
  torch, L, cfg, O = surface["torch"], surface["layers"], surface["cfg"], importlib.import_module("oracle.oracle")
  emb = L.embeddings.embed_as_linen(num_embeddings=cfg.vocab_size, num_features=cfg.emb_dim, config=cfg, name="token_embedder")
  table = surface["w"]["params/token_embedder/embedding"]
  emb.embedding = H.dev_bf16(table)
  ids = torch.tensor([[3, 1023, 0, 77]], dtype=torch.int32, device="cuda")
  assert np.array_equal(H.bits(emb(ids))[0], table[[3, 1023, 0, 77]])
  rope = L.embeddings.RotaryEmbedding(cfg.rope_min_timescale, cfg.rope_max_timescale, cfg.head_dim)
  rng = np.random.default_rng(8)
  x = H.rand_bf16(rng, (2, 3, 4, cfg.head_dim), 2.0)
  pos = np.array([[0, 5, 63], [1, 2, 40]], np.int32)
  rs, rc = O.rope_tables(cfg.head_dim, cfg.max_target_length)
  ref = x.copy()
  for b in range(2):
    for s in range(3):
      O.lib().orc_rope(O._p(ref[b, s], O._u16p), 4, cfg.head_dim, O._p(rs[pos[b, s]], O._u16p), O._p(rc[pos[b, s]], O._u16p))
  assert np.array_equal(H.bits(rope(H.dev_bf16(x), torch.from_numpy(pos).cuda())), ref)
  with pytest.raises(ValueError):
    L.embeddings.RotaryEmbedding(1, 10000, cfg.head_dim + 1)


def test_transformer_from_modules_matches_oracle(surface):
  """models.Transformer / decoders.Decoder / llama2.LlamaDecoderLayer / attentions.Attention / inference.paged_attention.PagedAttentionOp:
  prefill logits for EVERY row of two prompts (the reference computes [1, P, V] too) and four decode steps of both slots."""
  torch, L, cfg, ct, ctx, orc = surface["torch"], surface["layers"], surface["cfg"], surface["ct"], surface["ctx"], surface["orc"]
  model = L.models.Transformer(cfg, model_mode=ct.MODEL_MODE_PREFILL)
  model.load_params(surface["w"])
  pm = ctx.page_manager
  prompts = [H.synth.synth_prompt(cfg.vocab_size, 32, 900), (H.synth.synth_prompt(cfg.vocab_size, 32, 901)[0], 9)]
  V = cfg.vocab_size
  nxt = []
  for slot, (toks, tl) in enumerate(prompts):
    ps = pm.update_prefill_pages(None, slot, tl)
    t = torch.from_numpy(toks[:tl][None]).cuda()
    pos = torch.arange(tl, dtype=torch.int32, device="cuda")[None]
    logits = model(t, pos, decoder_segment_ids=torch.ones((1, tl), dtype=torch.int32, device="cuda"), model_mode=ct.MODEL_MODE_PREFILL, slot=slot, page_state=ps)
    assert logits.shape == (1, tl, V) and logits.dtype == torch.float32
    orc.page_state = orc.pm.update_prefill_pages(orc.page_state, slot, tl)
    o = orc.model.forward(toks[:tl], np.full((tl,), slot, np.int32), np.arange(tl, dtype=np.int32), np.ones((tl,), np.int32), orc.page_state.page_map,
                          logits_rows=list(range(tl)), want_logits=True)
    assert np.array_equal(logits[0].cpu().numpy().view(np.uint32), o["logits"].view(np.uint32)), slot
    nxt.append(int(o["argmax"][-1]))
  B = int(cfg.per_device_batch_size)
  tok = np.zeros((B,), np.int32)
  tok[:2] = nxt
  for step in range(4):
    ps = pm.update_decode_pages(None)
    orc.page_state = orc.pm.update_decode_pages(orc.page_state)
    pos = np.maximum(orc.page_state.sequence_lengths - 1, 0).astype(np.int32)
    logits = model(torch.from_numpy(tok[:, None]).cuda(), torch.from_numpy(pos[:, None]).cuda(), model_mode=ct.MODEL_MODE_AUTOREGRESSIVE, page_state=ps)
    act = orc.page_state.has_active_page.astype(np.int32)
    o = orc.model.forward(tok, np.arange(B, dtype=np.int32), pos, act, orc.page_state.page_map, logits_rows=list(range(B)), want_logits=True)
    got = logits[:, 0].cpu().numpy()
    assert np.array_equal(got[:2].view(np.uint32), o["logits"][:2].view(np.uint32)), step
    tok[:2] = o["argmax"][:2]


def test_attention_surface_errors(surface):
  L, cfg = surface["layers"], surface["cfg"]
  with pytest.raises(NotImplementedError):
    L.attentions.Attention(cfg, cfg.num_query_heads, cfg.num_kv_heads, cfg.head_dim, cfg.max_target_length, attention_kernel="dot_product")
  with pytest.raises(NotImplementedError):
    L.attentions.AttentionOp()(None)
  with pytest.raises(ValueError):      # built (tests/test_gpu_rope_llama31.py), but this context carries the default RotaryEmbedding's tables
    L.embeddings.LLaMARotaryEmbedding(1, 10000, cfg.head_dim)
