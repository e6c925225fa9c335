"""The oracle's Gemma-3 block (layers/gemma3.py:35-200, attentions.py:1872-1888, :2085-2088, :2246-2265), checked
on the CPU: the parameter names, the decoder invariants of the reference's tests (chunked == unchunked, prefill + autoregress == longer
prefill) over a full [5 local, 1 global] pattern, and what the Gemma-3 features add (the head norm against its float64 formula, the local
layers' own RoPE base, the layer pattern, q * query_pre_attn_scalar)."""
import numpy as np

import helpers as H
from oracle import oracle as O


def _cfg(**kw):
  base = dict(max_prefill_predict_length=48, max_target_length=128)
  base.update(kw)
  return H.make_config("gemma3-tiny", **base)


def test_weight_tree_and_config():
  cfg = _cfg()
  w = H.synth.synth_weights(cfg, 0)
  for part in ("pre_self_attention_norm/scale", "self_attention/query_norm/scale", "self_attention/key_norm/scale", "self_attention/out/kernel",
               "post_self_attention_norm/scale", "mlp/mlp_layer_norm/scale", "mlp/wi_1/kernel", "post_ffw_norm/scale"):
    assert f"params/decoder/layers_6/{part}" in w
  assert w["params/decoder/layers_0/self_attention/query_norm/scale"].shape == (cfg.head_dim,)
  assert "params/decoder/logits_dense/kernel" not in w
  assert H.sublayers(cfg) == cfg.num_decoder_layers == 7
  oc = H.oracle_config(cfg)
  assert oc.decoder_block == "gemma3" and oc.query_pre_attn_scalar == cfg.head_dim ** -0.5 and oc.local_rope_max_timescale == 10000.0
  # gemma3.py:50-58
  me = H.maxengine_module()
  big = H.make_config("gemma3-4b", base_num_decoder_layers=1, max_prefill_predict_length=16, max_target_length=32)
  assert me.get_query_pre_attn_scalar(big) == 256 ** -0.5


def test_head_norm_restatement_against_float64():
  lib = O.lib()
  rng = np.random.default_rng(0)
  for hd in (64, 128, 256):
    x = H.synth.f32_to_bf16_bits(rng.standard_normal(hd).astype(np.float32) * 3)
    sc = H.synth.f32_to_bf16_bits(1.0 + 0.25 * rng.standard_normal(hd).astype(np.float32))
    out = np.zeros_like(x)
    lib.orc_rmsnorm(x.ctypes.data_as(O._u16p), sc.ctypes.data_as(O._u16p), out.ctypes.data_as(O._u16p), hd, O.ctypes.c_float(1e-6))
    xf, sf = H.synth.bf16_bits_to_f32(x).astype(np.float64), H.synth.bf16_bits_to_f32(sc).astype(np.float64)
    ref = xf / np.sqrt(np.mean(xf * xf) + 1e-6) * sf           # normalizations.py:57-70
    got = H.synth.bf16_bits_to_f32(out).astype(np.float64)
    assert np.all(np.abs(got - ref) <= 2.0 ** -7 * np.abs(ref) + 1e-6)   # two bf16 roundings


def _prefill_logits(cfg, w, toks, n):
  orc = H.make_oracle(cfg, w)
  pre, res = orc.prefill(toks, n, slot=0, want_logits=True)
  return pre["logits"].copy().reshape(-1).view(np.uint32), orc


def test_chunked_prefill_equals_whole_and_autoregress_equals_longer_prefill():
  cfg = _cfg()
  w = H.synth.synth_weights(cfg, 3)
  toks, _ = H.synth.synth_prompt(cfg.vocab_size, 48, 7)
  orc = H.make_oracle(cfg, w)
  pre, res = orc.prefill(toks, 40, slot=0, want_logits=True)
  orc2 = H.make_oracle(cfg, w)
  done, p2 = 0, None
  for n in (16, 8, 16):
    chunk = np.zeros((48,), np.int32)
    chunk[:n] = toks[done:done + n]
    p2, r2 = orc2.prefill(chunk, n, slot=0, existing_prefix_tokens=None if done == 0 else toks[:done], want_logits=True)
    done += n
  assert np.array_equal(res, r2) and np.array_equal(pre["logits"].view(np.uint32), p2["logits"].view(np.uint32))
  for layer in range(H.sublayers(cfg)):
    k1, v1 = orc.model.gather_kv(layer, orc.page_state.page_map[0], 40)
    k2, v2 = orc2.model.gather_kv(layer, orc2.page_state.page_map[0], 40)
    assert np.array_equal(k1, k2) and np.array_equal(v1, v2)
  ds = orc.insert(pre, orc.init_decode_state(), 0)
  gen = [int(res[0, 0])]
  for _ in range(5):
    ds, r = orc.generate(ds, want_logits=True)
    gen.append(int(r[0, 0]))
  orc3 = H.make_oracle(cfg, w)
  long = np.zeros((48,), np.int32)
  long[:40] = toks[:40]
  long[40:45] = gen[:5]
  p3, r3 = orc3.prefill(long, 45, slot=0, want_logits=True)
  assert int(r3[0, 0]) == gen[5]
  assert np.array_equal(p3["logits"].reshape(-1).view(np.uint32), ds["logits"][0].reshape(-1).view(np.uint32))


def test_pattern_local_rope_and_scalar():
  w = H.synth.synth_weights(_cfg(), 5)
  toks, _ = H.synth.synth_prompt(1024, 48, 11)
  base, _ = _prefill_logits(_cfg(), w, toks, 44)
  # a window past the sequence end is no window
  big, _ = _prefill_logits(_cfg(sliding_window_size=4096), w, toks, 44)
  off, _ = _prefill_logits(_cfg(sliding_window_size=0), w, toks, 44)
  assert np.array_equal(big, off) and not np.array_equal(base, off)
  # the local layers have their own RoPE base: equal bases == no local base; different bases differ
  same, _ = _prefill_logits(_cfg(local_rope_max_timescale=1000000), w, toks, 44)
  none, _ = _prefill_logits(_cfg(local_rope_max_timescale=-1), w, toks, 44)
  assert np.array_equal(same, none) and not np.array_equal(base, none)
  # layer 5 of every six is global: with a single layer... the model of 5 layers is all local, so its keys beyond the window never matter
  cfg5 = _cfg(base_num_decoder_layers=5)
  w5 = H.synth.synth_weights(cfg5, 6)
  orc = H.make_oracle(cfg5, w5)
  pre, _ = orc.prefill(toks, 44, slot=0, want_logits=True)
  lg = pre["logits"].copy()
  # a 6-layer model has one global layer: the window no longer bounds what a row depends on -- the first token changes the last logits
  cfg6 = _cfg(base_num_decoder_layers=6)
  w6 = H.synth.synth_weights(cfg6, 6)
  t2 = toks.copy()
  t2[0] = (int(t2[0]) + 1) % 1024
  a6, _ = _prefill_logits(cfg6, w6, toks, 44)
  b6, _ = _prefill_logits(cfg6, w6, t2, 44)
  assert not np.array_equal(a6, b6)
  # 2 local layers, window 8: row 43 depends on keys >= 36 of layer 1, which depend on keys >= 29 of layer 0: token 0 is out of reach
  cfg2 = _cfg(base_num_decoder_layers=2, sliding_window_size=8)
  w2 = H.synth.synth_weights(cfg2, 6)
  a2, _ = _prefill_logits(cfg2, w2, toks, 44)
  b2, _ = _prefill_logits(cfg2, w2, t2, 44)
  assert np.array_equal(a2, b2)
  assert lg.size == 1024
  # q * query_pre_attn_scalar is applied: scalar 1 (none) gives other logits than head_dim ** -0.5
  oc = H.oracle_config(_cfg())
  oc.query_pre_attn_scalar = 1.0
  orc1 = O.OracleEngine(O.OracleModel(oc, w))
  p1, _ = orc1.prefill(toks, 44, slot=0, want_logits=True)
  assert not np.array_equal(p1["logits"].reshape(-1).view(np.uint32), base)
