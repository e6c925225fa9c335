"""The oracle's Mixtral block (layers/mixtral.py:52-170, layers/moe.py:170-204, :326-345, :420-484, :855-942) on
the CPU: routing against numpy (top_k order, ties, softmax over the k logits), the mixture against a float64 formula built from the
oracle's own per-expert MLP, a one-expert mixture == the llama2 block bit for bit, and the decoder invariants of the reference's tests."""
import ctypes

import numpy as np

import helpers as H
from oracle import oracle as O


def _cfg(**kw):
  base = dict(max_prefill_predict_length=48, max_target_length=128)
  base.update(kw)
  return H.make_config("mixtral-tiny", **base)


def test_routing_restatement():
  lib = O.lib()
  rng = np.random.default_rng(0)
  for trial in range(200):
    ne, k = [(8, 2), (8, 2), (4, 1), (16, 4), (8, 8)][trial % 5]
    lg = H.synth.bf16_bits_to_f32(H.synth.f32_to_bf16_bits(rng.standard_normal(ne).astype(np.float32) * 2))
    if trial % 3 == 0:            # ties: jax.lax.top_k keeps the lower index first
      lg[rng.integers(0, ne)] = lg[rng.integers(0, ne)]
    idx = np.zeros((k,), np.int32)
    wt = np.zeros((k,), np.float32)
    lib.orc_route_row(lg.ctypes.data_as(O._f32p), ne, k, idx.ctypes.data_as(O._i32p), wt.ctypes.data_as(O._f32p))
    order = np.argsort(-lg, kind="stable")[:k]
    assert np.array_equal(idx, order)
    ref = np.exp(lg[order].astype(np.float64) - lg[order[0]])
    ref /= ref.sum()
    assert np.all(np.abs(wt - ref) <= 2.0 ** -8 * ref + 1e-7)     # bf16(softmax in f32)
    assert np.array_equal(H.synth.bf16_bits_to_f32(H.synth.f32_to_bf16_bits(wt)), wt)


def test_weight_tree_has_the_reference_names():
  cfg = _cfg()
  w = H.synth.synth_weights(cfg, 0)
  assert w["params/decoder/layers_2/MoeBlock_0/gate/kernel"].shape == (256, 8)
  assert w["params/decoder/layers_2/MoeBlock_0/wi_0"].shape == (8, 256, 512)
  assert w["params/decoder/layers_2/MoeBlock_0/wo"].shape == (8, 512, 256)
  assert "params/decoder/layers_0/post_self_attention_layer_norm/scale" in w
  assert "params/decoder/layers_0/mlp/wi_0/kernel" not in w


def test_one_expert_mixture_is_the_llama2_block():
  """num_experts = num_experts_per_tok = 1: the softmax over one logit is 1, fma(y, 1, 0) = y, so the model must reproduce the llama2
  block with the same kernels bit for bit (ties the mixture's wiring to the dense block)."""
  cfg_m = _cfg(num_experts=1, num_experts_per_tok=1)
  cfg_l = H.make_config("mixtral-tiny", decoder_block="llama2", num_experts=1, num_experts_per_tok=1, max_prefill_predict_length=48, max_target_length=128)
  wm = H.synth.synth_weights(cfg_m, 2)
  wl = {}
  for name, a in wm.items():
    if "MoeBlock_0/gate" in name:
      continue
    name = name.replace("post_self_attention_layer_norm/scale", "mlp/mlp_layer_norm/scale")
    for src, dst in (("MoeBlock_0/wi_0", "mlp/wi_0/kernel"), ("MoeBlock_0/wi_1", "mlp/wi_1/kernel"), ("MoeBlock_0/wo", "mlp/wo/kernel")):
      if name.endswith(src):
        name, a = name.replace(src, dst), a[0]
    wl[name] = a
  toks, _ = H.synth.synth_prompt(cfg_m.vocab_size, 48, 7)
  om, ol = H.make_oracle(cfg_m, wm), H.make_oracle(cfg_l, wl)
  pm, rm = om.prefill(toks, 40, slot=0, want_logits=True)
  pl, rl = ol.prefill(toks, 40, slot=0, want_logits=True)
  assert np.array_equal(rm, rl) and np.array_equal(pm["logits"].view(np.uint32), pl["logits"].view(np.uint32))


def test_mixture_against_float64_combination():
  cfg = _cfg()
  w = H.synth.synth_weights(cfg, 3)
  orc = H.make_oracle(cfg, w)
  m, lib = orc.model, O.lib()
  rng = np.random.default_rng(5)
  R, E, K, NE = 9, cfg.emb_dim, 2, 8
  xn = H.synth.f32_to_bf16_bits(rng.standard_normal((R, E)).astype(np.float32))
  out = np.zeros((R, E), np.uint16)
  routing = np.zeros((R, K), np.int32)
  lib.orc_moe_rows(ctypes.byref(m.c), 1, xn.ctypes.data_as(O._u16p), R, out.ctypes.data_as(O._u16p), routing.ctypes.data_as(O._i32p))
  # gate logits in float64, routing, and the mixture of float64 MLPs
  xf = H.synth.bf16_bits_to_f32(xn).astype(np.float64)
  p = "params/decoder/layers_1/MoeBlock_0/"
  g = H.synth.bf16_bits_to_f32(w[p + "gate/kernel"]).astype(np.float64)
  w0, w1, wo = (H.synth.bf16_bits_to_f32(w[p + n]).astype(np.float64) for n in ("wi_0", "wi_1", "wo"))
  lg = xf @ g
  used = set()
  for r in range(R):
    order = np.argsort(-lg[r], kind="stable")[:K]
    gap = lg[r][order[K - 1]] - np.sort(lg[r])[::-1][K]
    if gap > 0.05 and lg[r][order[0]] - lg[r][order[1]] > 0.05:     # bf16 logits: only clearly separated rows pin the order
      assert np.array_equal(routing[r], order), (r, routing[r], order)
    sel = routing[r]
    used.update(int(e) for e in sel)
    wt = np.exp(lg[r][sel] - lg[r][sel].max())
    wt /= wt.sum()
    ref = np.zeros((E,))
    for j, e in enumerate(sel):
      a, b = xf[r] @ w0[e], xf[r] @ w1[e]
      ref += wt[j] * ((a / (1.0 + np.exp(-a)) * b) @ wo[e])
    got = H.synth.bf16_bits_to_f32(out[r]).astype(np.float64)
    assert np.max(np.abs(got - ref)) <= 0.03 * np.max(np.abs(ref)) + 1e-3
  assert len(used) >= 4          # the rows really go to different experts


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
