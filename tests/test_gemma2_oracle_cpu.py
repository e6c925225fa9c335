"""The oracle's Gemma-2 block (layers/gemma2.py:69-236, attentions.py:625-631 and :1257-1259,
decoders.py:552-565), checked on the CPU: the scalar restatements against float64 formulas, and the invariants the reference's tests state for
any decoder (chunked == unchunked, prefill + autoregress == longer prefill) plus the ones the Gemma features add (a window at least as long
as the sequence changes nothing; a window of w hides exactly the keys at distance >= w; the cap bounds the logits)."""
import ctypes

import numpy as np
import pytest

import helpers as H
from oracle import oracle as O


def _gemma_cfg(**kw):
  base = dict(max_prefill_predict_length=48, max_target_length=128)
  base.update(kw)
  return H.make_config("gemma2-tiny", **base)


def test_tanh_and_gelu_gate_restatement():
  lib = O.lib()
  x = np.concatenate([np.linspace(-12, 12, 100001), [0.0, 1e-8, -1e-8, 0.0624, 0.0626, 9.99, 10.0, 40.0]]).astype(np.float32)
  y = np.zeros_like(x)
  lib.orc_tanh_vec(x.ctypes.data_as(O._f32p), y.ctypes.data_as(O._f32p), x.size)
  ref = np.tanh(x.astype(np.float64))
  assert np.max(np.abs(y - ref) / np.maximum(np.abs(ref), 1e-30)) < 5e-7
  assert np.all(np.abs(y) <= 1.0) and np.array_equal(np.sign(y), np.sign(x))
  # the gate against jax.nn.gelu(approximate=True) evaluated in float64: a handful of bf16 roundings apart
  rng = np.random.default_rng(0)
  a = H.synth.f32_to_bf16_bits(rng.standard_normal(20000).astype(np.float32) * 3)
  b = H.synth.f32_to_bf16_bits(rng.standard_normal(20000).astype(np.float32))
  out = np.zeros_like(a)
  lib.orc_geglu(a.ctypes.data_as(O._u16p), b.ctypes.data_as(O._u16p), out.ctypes.data_as(O._u16p), a.size)
  af, bf = H.synth.bf16_bits_to_f32(a).astype(np.float64), H.synth.bf16_bits_to_f32(b).astype(np.float64)
  ref = af * 0.5 * (1.0 + np.tanh(np.sqrt(2 / np.pi) * (af + 0.044715 * af**3))) * bf
  got = H.synth.bf16_bits_to_f32(out).astype(np.float64)
  # bf16 arithmetic, every primitive rounded: tanh near -1 keeps 8 bits, so cdf carries an absolute error of about 2^-9 (times |a b|)
  assert np.all(np.abs(got - ref) <= 0.004 * np.abs(af * bf) + 0.02 * np.abs(ref) + 1e-3)


def _engine(cfg, seed=3):
  w = H.synth.synth_weights(cfg, seed)
  return H.make_oracle(cfg, w), w


def test_weight_tree_has_the_reference_names():
  cfg = _gemma_cfg()
  w = H.synth.synth_weights(cfg, 0)
  names = set(w)
  for part in ("pre_self_attention_norm_local/scale", "self_attention_global/out/kernel", "post_self_attention_norm_global/scale",
               "mlp_local/mlp_layer_norm/scale", "pre_ffw_norm_global/scale", "mlp_global/wi_1/kernel", "post_ffw_norm_local/scale"):
    assert f"params/decoder/layers_1/{part}" in names
  assert "params/decoder/logits_dense/kernel" not in names        # logits_via_embedding
  assert H.sublayers(cfg) == 2 * cfg.num_decoder_layers == 4


def test_chunked_prefill_equals_whole_and_autoregress_equals_longer_prefill():
  cfg = _gemma_cfg()
  orc, w = _engine(cfg)
  toks, _ = H.synth.synth_prompt(cfg.vocab_size, 48, 7)
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
  # decode 5 tokens, then prefill prompt + those tokens in one pass: same next token and logits (window 24 < 45 keys: the mask is active)
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


def test_window_semantics_and_caps():
  cfg_w = _gemma_cfg(sliding_window_size=24)
  cfg_big = _gemma_cfg(sliding_window_size=4096)
  cfg_off = _gemma_cfg(sliding_window_size=0)
  w = H.synth.synth_weights(cfg_w, 5)
  toks, _ = H.synth.synth_prompt(cfg_w.vocab_size, 48, 11)
  outs = {}
  for name, cfg in (("w", cfg_w), ("big", cfg_big), ("off", cfg_off)):
    orc = H.make_oracle(cfg, w)
    pre, res = orc.prefill(toks, 44, slot=0, want_logits=True)
    outs[name] = pre["logits"].copy()
  assert np.array_equal(outs["big"].view(np.uint32), outs["off"].view(np.uint32))     # a window past the sequence end is no window
  assert not np.array_equal(outs["w"].view(np.uint32), outs["off"].view(np.uint32))  # 44 keys > 24: the local sub-layers see fewer
  assert np.max(np.abs(outs["w"])) <= 30.0                                           # final_logits_soft_cap
  # a row sees exactly the last `window` keys on a local sub-layer: garbage in the hidden keys' V pages changes nothing
  orc = H.make_oracle(cfg_w, w)
  m = orc.model
  lib = O.lib()
  rng = np.random.default_rng(1)
  P, KV, D = cfg_w.pagedattn_tokens_per_page, cfg_w.num_kv_heads, cfg_w.head_dim
  n_keys = 40
  page_row = np.arange(1, 1 + (n_keys + P - 1) // P, dtype=np.int32)
  m.kpages[:] = H.synth.f32_to_bf16_bits(rng.standard_normal(m.kpages.size).astype(np.float32))
  m.vpages[:] = H.synth.f32_to_bf16_bits(rng.standard_normal(m.vpages.size).astype(np.float32))
  q = H.synth.f32_to_bf16_bits(rng.standard_normal(cfg_w.num_query_heads * D).astype(np.float32))
  sc = np.zeros((n_keys,), np.float32)

  def attend(window):
    out = np.zeros((cfg_w.num_query_heads * D,), np.uint16)
    lib.orc_attend_row_ex(ctypes.byref(m.c), 0, q.ctypes.data_as(O._u16p), page_row.ctypes.data_as(O._i32p), n_keys, out.ctypes.data_as(O._u16p),
                          sc.ctypes.data_as(O._f32p), ctypes.c_float(50.0), window)
    return out

  ref = attend(24)
  pe = P * KV * D
  for j in range(n_keys - 24):           # keys 0 .. 15 are hidden from the row at position 39
    off = int(page_row[j // P]) * pe + (j % P) * KV * D
    m.vpages[off:off + KV * D] = 0x7F00   # huge values: any non-zero weight would show
    m.kpages[off:off + KV * D] = 0x4700
  assert np.array_equal(attend(24), ref)
  assert not np.array_equal(attend(25), ref)
