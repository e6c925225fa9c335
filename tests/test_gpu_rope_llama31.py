"""rope_type "llama3.1" (layers/embeddings.py:354-488 LLaMARotaryEmbedding) on the GPU against the oracle, bit for bit:
every kernel that rotates -- the few-rows QKV launch, the split-K path's finalize / decode prologues, the ring GEMM's QKV epilogue, the op-level
rope -- and the K cache export (the kernels rotate de-interleaved Q / K columns and store the results back in the model's feature order)."""
import ctypes
import importlib

import numpy as np
import pytest

import helpers as H
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _cfg(**kw):
  base = dict(rope_type="llama3.1", max_prefill_predict_length=32, max_target_length=96, pagedattn_num_pages=64)
  base.update(kw)
  return H.make_config(**base)


def _pair(cfg, seed):
  w = H.synth.synth_weights(cfg, seed)
  eng, params = H.make_engine(cfg, w)
  return eng, params, H.make_oracle(cfg, w)


def test_tables_match_the_oracle():
  cfg = _cfg()
  eng, _, orc = _pair(cfg, 1)
  n = cfg.max_target_length * cfg.head_dim // 2
  sn, cs = np.zeros((n,), np.float32), np.zeros((n,), np.float32)
  H.lib.check(eng.lib.ixt_rope_tables_f32(eng._ctx, sn.ctypes.data_as(ctypes.c_void_p), cs.ctypes.data_as(ctypes.c_void_p)))
  assert np.array_equal(sn.view(np.uint32), orc.model.rope_sin32.reshape(-1).view(np.uint32))
  assert np.array_equal(cs.view(np.uint32), orc.model.rope_cos32.reshape(-1).view(np.uint32))


@pytest.mark.parametrize("slots,use_scale", [(2, True), (3, False), (20, True)])
def test_prefill_and_decode_match_the_oracle(slots, use_scale):
  """2 / 3 slots: the few-rows path (k_sdot QKV epilogue) for prefill chunks <= 16 rows and for decode; 32-row prompts and 20 slots: the split-K
  path (k_qkv_finalize, the decode attention's prologue)."""
  import torch
  cfg = _cfg(per_device_batch_size=float(slots), rope_use_scale=use_scale, pagedattn_num_pages=max(64, slots * 12 + 1))
  eng, params, orc = _pair(cfg, 2)
  ds, ods = eng.init_decode_state(materialize_logits=True), orc.init_decode_state()
  for s in range(slots):
    toks, tl = H.synth.synth_prompt(cfg.vocab_size, 32, 50 + s)
    tl = [tl, 9, 16][s % 3]
    pre, res = eng.prefill(params=params, padded_tokens=toks, true_length=tl, slot=s)
    opre, ores = orc.prefill(toks, tl, slot=s, want_logits=True)
    assert np.array_equal(res.data.cpu().numpy(), ores)
    assert np.array_equal(pre["logits"].cpu().numpy().reshape(-1).view(np.uint32), opre["logits"].reshape(-1).view(np.uint32))
    ds, ods = eng.insert(pre, ds, s), orc.insert(opre, ods, s)
  for step in range(6):
    ds, r = eng.generate(params, ds)
    ods, orr = orc.generate(ods, want_logits=True)
    assert np.array_equal(r.data.cpu().numpy(), orr), step
    assert np.array_equal(ds["logits"].reshape(slots, -1).cpu().numpy().view(np.uint32), np.asarray(ods["logits"]).reshape(slots, -1).view(np.uint32)), step
  for layer in range(cfg.num_decoder_layers):            # the cache export undoes the engine's feature order
    k, v = eng.gather_kv(layer, 0, int(ods["next_pos"][0, 0]))
    ok, ov = orc.model.gather_kv(layer, orc.page_state.page_map[0], int(ods["next_pos"][0, 0]))
    assert np.array_equal(H.bits(k), ok) and np.array_equal(H.bits(v), ov)
  torch.cuda.synchronize()


def test_long_prompt_through_the_ring_gemm_epilogue():
  """A 160-token prompt in one chunk (>= 128 rows: the LDS-ring GEMM with the QKV epilogue) and chunked 5 x 32 (split-K + k_qkv_finalize)."""
  import torch
  maxengine = H.maxengine_module()
  toks = H.synth.synth_prompt(1024, 160, 5)[0]
  outs = []
  for chunk in (160, 32):
    cfg = _cfg(max_prefill_predict_length=chunk, max_target_length=192, use_chunked_prefill=True, prefill_chunk_size=chunk)
    eng, params, orc = _pair(cfg, 3)
    done, pre, res = 0, None, None
    while done < 160:
      ep = None if done == 0 else maxengine.ExistingPrefix(cache=pre["cache"], common_prefix_tokens=toks[:done])
      pre, res = eng.prefill(params=params, padded_tokens=toks[done:done + chunk], true_length=chunk, slot=0, existing_prefix=ep)
      done += chunk
    if chunk == 160:
      opre, ores = orc.prefill(toks, 160, slot=0, want_logits=True)
      assert np.array_equal(res.data.cpu().numpy(), ores)
      assert np.array_equal(pre["logits"].cpu().numpy().reshape(-1).view(np.uint32), opre["logits"].reshape(-1).view(np.uint32))
    outs.append(pre["logits"].cpu().numpy().reshape(-1).view(np.uint32).copy())
    torch.cuda.synchronize()
  assert np.array_equal(outs[0], outs[1])


def test_module_level_llama_rotary_embedding():
  """layers.embeddings.LLaMARotaryEmbedding over ixt_op_rope: neighbouring pairs of the caller's rows, against orc_rope_llama31."""
  import torch
  runtime = importlib.import_module("maxtext-indextts2_amd.runtime")
  emb = importlib.import_module("maxtext-indextts2_amd.layers.embeddings")
  cfg = _cfg()
  saved = runtime._default
  try:
    runtime.set_context(runtime.OpContext(cfg))
    rope = emb.LLaMARotaryEmbedding(cfg.rope_min_timescale, cfg.rope_max_timescale, cfg.head_dim, use_scale=True, config=cfg)
    rng = np.random.default_rng(0)
    B, S, N, D = 2, 7, 4, cfg.head_dim
    xb = H.rand_bf16(rng, (B, S, N, D))
    pos = rng.integers(0, cfg.max_target_length, size=(B, S)).astype(np.int32)
    got = H.bits(rope(H.dev_bf16(xb), torch.from_numpy(pos).cuda()))
    sn, cs = O.rope_tables_llama31(D, cfg.max_target_length, float(cfg.rope_min_timescale), float(cfg.rope_max_timescale), True)
    want = xb.copy().reshape(B * S, N, D)
    lib = O.lib()
    lib.orc_rope_llama31.argtypes = [O._u16p, ctypes.c_int, ctypes.c_int, O._f32p, O._f32p]
    for r in range(B * S):
      row = np.ascontiguousarray(want[r])
      p = int(pos.reshape(-1)[r])
      lib.orc_rope_llama31(O._p(row, O._u16p), N, D, O._p(np.ascontiguousarray(sn[p]), O._f32p), O._p(np.ascontiguousarray(cs[p]), O._f32p))
      want[r] = row
    assert np.array_equal(got.reshape(B * S, N, D), want)
    with pytest.raises(ValueError):
      emb.LLaMARotaryEmbedding(cfg.rope_min_timescale, cfg.rope_max_timescale, cfg.head_dim, use_scale=False, config=cfg)
  finally:
    runtime._default = saved
