"""Synthetic weights / prompts -- host-side numpy, shared by tests and bench.

Weights follow the reference's init distributions, drawn from numpy `Generator(PCG64(seed))`:
  * embedding ~ N(0, 1)                                 (layers/models.py:68)
  * dense kernels ~ truncated normal (+-2 sigma), variance 1/fan_in, sigma corrected by
    1/0.87962566 as jax.nn.initializers.variance_scaling does      (layers/linears.py:106)
  * query kernel additionally divided by sqrt(head_dim)            (layers/attentions.py:1900-1904)
  * norm scales = 1                                                (layers/normalizations.py:40)
all rounded once to bf16 (round-to-nearest-even).  Keys are the unscanned MaxText tree paths.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np

_TRUNC_STD = 0.87962566103423978


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
  u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
  return ((u + (0x7FFF + ((u >> 16) & 1))) >> 16).astype(np.uint16)


def bf16_bits_to_f32(h: np.ndarray) -> np.ndarray:
  return (np.ascontiguousarray(h, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def _trunc_normal(rng: np.random.Generator, n: int) -> np.ndarray:
  x = rng.standard_normal(n, dtype=np.float32)
  bad = np.nonzero(np.abs(x) > 2.0)[0]
  while bad.size:
    x[bad] = rng.standard_normal(bad.size, dtype=np.float32)
    bad = bad[np.abs(x[bad]) > 2.0]
  return x


def _dense(rng, shape, fan_in, extra_div=1.0):
  std = math.sqrt(1.0 / fan_in) / _TRUNC_STD / extra_div
  n = int(np.prod(shape))
  return f32_to_bf16_bits(_trunc_normal(rng, n) * np.float32(std)).reshape(shape)


def layer_paths(i: int):
  p = f"params/decoder/layers_{i}/"
  return {
      "ln1": p + "pre_self_attention_layer_norm/scale",
      "q": p + "self_attention/query/kernel",
      "k": p + "self_attention/key/kernel",
      "v": p + "self_attention/value/kernel",
      "o": p + "self_attention/out/kernel",
      "ln2": p + "mlp/mlp_layer_norm/scale",
      "wi0": p + "mlp/wi_0/kernel",
      "wi1": p + "mlp/wi_1/kernel",
      "wo": p + "mlp/wo/kernel",
  }


def gemma2_layer_paths(i: int, sub: str):
  """layers/gemma2.py:69-236 parameter names of reference layer i; sub = "local" | "global"."""
  p = f"params/decoder/layers_{i}/"
  return {
      "ln1": p + f"pre_self_attention_norm_{sub}/scale",
      "q": p + f"self_attention_{sub}/query/kernel",
      "k": p + f"self_attention_{sub}/key/kernel",
      "v": p + f"self_attention_{sub}/value/kernel",
      "o": p + f"self_attention_{sub}/out/kernel",
      "pan": p + f"post_self_attention_norm_{sub}/scale",
      # the local MLP carries its own pre-norm (mlp_block(use_pre_norm=True), gemma2.py:121-133), the global one is a separate module
      "ln2": p + ("mlp_local/mlp_layer_norm/scale" if sub == "local" else "pre_ffw_norm_global/scale"),
      "wi0": p + f"mlp_{sub}/wi_0/kernel",
      "wi1": p + f"mlp_{sub}/wi_1/kernel",
      "wo": p + f"mlp_{sub}/wo/kernel",
      "pfn": p + f"post_ffw_norm_{sub}/scale",
  }


def synth_weights_gemma2(cfg, seed: int = 0) -> Dict[str, np.ndarray]:
  """Random Gemma-2 tree (same distributions as synth_weights); norm scales are drawn around 1 so that every norm is exercised."""
  rng = np.random.Generator(np.random.PCG64(seed))
  E, Q, KV, D, F, V, L = (cfg.emb_dim, cfg.num_query_heads, cfg.num_kv_heads, cfg.head_dim, cfg.mlp_dim, cfg.vocab_size,
                          cfg.num_decoder_layers)

  def scale():
    return f32_to_bf16_bits(1.0 + 0.25 * rng.standard_normal(E, dtype=np.float32))

  w: Dict[str, np.ndarray] = {}
  w["params/token_embedder/embedding"] = f32_to_bf16_bits(rng.standard_normal(V * E, dtype=np.float32)).reshape(V, E)
  for i in range(L):
    for sub in ("local", "global"):
      p = gemma2_layer_paths(i, sub)
      w[p["ln1"]] = scale()
      w[p["q"]] = _dense(rng, (E, Q, D), E, math.sqrt(D))
      w[p["k"]] = _dense(rng, (E, KV, D), E)
      w[p["v"]] = _dense(rng, (E, KV, D), E)
      w[p["o"]] = _dense(rng, (Q, D, E), Q * D)
      w[p["ln2"]] = scale()
      w[p["wi0"]] = _dense(rng, (E, F), E)
      w[p["wi1"]] = _dense(rng, (E, F), E)
      w[p["wo"]] = _dense(rng, (F, E), F)
      if getattr(cfg, "use_post_attn_norm", False):
        w[p["pan"]] = scale()
      if getattr(cfg, "use_post_ffw_norm", False):
        w[p["pfn"]] = scale()
  w["params/decoder/decoder_norm/scale"] = scale()
  if not getattr(cfg, "logits_via_embedding", False):
    w["params/decoder/logits_dense/kernel"] = _dense(rng, (E, V), E)
  return w


def gemma3_layer_paths(i: int):
  """layers/gemma3.py:83-200 parameter names of layer i (query / key norms: attentions.py:1872-1888)."""
  p = f"params/decoder/layers_{i}/"
  d = layer_paths(i)
  d.update({
      "ln1": p + "pre_self_attention_norm/scale",
      "qn": p + "self_attention/query_norm/scale",
      "kn": p + "self_attention/key_norm/scale",
      "pan": p + "post_self_attention_norm/scale",
      "pfn": p + "post_ffw_norm/scale",
  })
  return d


def synth_weights_gemma3(cfg, seed: int = 0) -> Dict[str, np.ndarray]:
  """Random Gemma-3 tree; every norm scale (also the per-head query / key norms) is drawn around 1."""
  rng = np.random.Generator(np.random.PCG64(seed))
  E, Q, KV, D, F, V, L = (cfg.emb_dim, cfg.num_query_heads, cfg.num_kv_heads, cfg.head_dim, cfg.mlp_dim, cfg.vocab_size,
                          cfg.num_decoder_layers)

  def scale(n=E):
    return f32_to_bf16_bits(1.0 + 0.25 * rng.standard_normal(n, dtype=np.float32))

  w: Dict[str, np.ndarray] = {}
  w["params/token_embedder/embedding"] = f32_to_bf16_bits(rng.standard_normal(V * E, dtype=np.float32)).reshape(V, E)
  for i in range(L):
    p = gemma3_layer_paths(i)
    w[p["ln1"]] = scale()
    w[p["q"]] = _dense(rng, (E, Q, D), E, math.sqrt(D))
    w[p["k"]] = _dense(rng, (E, KV, D), E)
    w[p["v"]] = _dense(rng, (E, KV, D), E)
    w[p["o"]] = _dense(rng, (Q, D, E), Q * D)
    w[p["qn"]] = scale(D)
    w[p["kn"]] = scale(D)
    w[p["ln2"]] = scale()
    w[p["wi0"]] = _dense(rng, (E, F), E)
    w[p["wi1"]] = _dense(rng, (E, F), E)
    w[p["wo"]] = _dense(rng, (F, E), F)
    if getattr(cfg, "use_post_attn_norm", False):
      w[p["pan"]] = scale()
    if getattr(cfg, "use_post_ffw_norm", False):
      w[p["pfn"]] = scale()
  w["params/decoder/decoder_norm/scale"] = scale()
  if not getattr(cfg, "logits_via_embedding", False):
    w["params/decoder/logits_dense/kernel"] = _dense(rng, (E, V), E)
  return w


def synth_weights_mixtral(cfg, seed: int = 0) -> Dict[str, np.ndarray]:
  """Random Mixtral tree (layers/mixtral.py:52-170; MoeBlock_0: gate/kernel [emb, experts], wi_0 / wi_1 [experts, emb, mlp], wo [experts, mlp, emb]).
  The gate kernel is drawn 4 x wider than a dense kernel so that the routing logits are well separated and vary from row to row."""
  rng = np.random.Generator(np.random.PCG64(seed))
  E, Q, KV, D, F, V, L, NE = (cfg.emb_dim, cfg.num_query_heads, cfg.num_kv_heads, cfg.head_dim, cfg.mlp_dim, cfg.vocab_size,
                              cfg.num_decoder_layers, cfg.num_experts)
  one = f32_to_bf16_bits(np.ones((E,), np.float32))
  w: Dict[str, np.ndarray] = {}
  w["params/token_embedder/embedding"] = f32_to_bf16_bits(rng.standard_normal(V * E, dtype=np.float32)).reshape(V, E)
  for i in range(L):
    p = f"params/decoder/layers_{i}/"
    w[p + "pre_self_attention_layer_norm/scale"] = one.copy()
    w[p + "self_attention/query/kernel"] = _dense(rng, (E, Q, D), E, math.sqrt(D))
    w[p + "self_attention/key/kernel"] = _dense(rng, (E, KV, D), E)
    w[p + "self_attention/value/kernel"] = _dense(rng, (E, KV, D), E)
    w[p + "self_attention/out/kernel"] = _dense(rng, (Q, D, E), Q * D)
    w[p + "post_self_attention_layer_norm/scale"] = one.copy()
    w[p + "MoeBlock_0/gate/kernel"] = _dense(rng, (E, NE), E, 0.25)
    w[p + "MoeBlock_0/wi_0"] = _dense(rng, (NE, E, F), E)
    w[p + "MoeBlock_0/wi_1"] = _dense(rng, (NE, E, F), E)
    w[p + "MoeBlock_0/wo"] = _dense(rng, (NE, F, E), F)
  w["params/decoder/decoder_norm/scale"] = one.copy()
  w["params/decoder/logits_dense/kernel"] = _dense(rng, (E, V), E)
  return w


def synth_weights(cfg, seed: int = 0) -> Dict[str, np.ndarray]:
  """cfg needs vocab_size, emb_dim, num_query_heads, num_kv_heads, head_dim, mlp_dim, num_decoder_layers."""
  if getattr(cfg, "decoder_block", "llama2") == "mixtral":
    return synth_weights_mixtral(cfg, seed)
  if getattr(cfg, "decoder_block", "llama2") == "gemma2":
    return synth_weights_gemma2(cfg, seed)
  if getattr(cfg, "decoder_block", "llama2") == "gemma3":
    return synth_weights_gemma3(cfg, seed)
  rng = np.random.Generator(np.random.PCG64(seed))
  E, Q, KV, D, F, V, L = (cfg.emb_dim, cfg.num_query_heads, cfg.num_kv_heads, cfg.head_dim, cfg.mlp_dim, cfg.vocab_size,
                          cfg.num_decoder_layers)
  one = f32_to_bf16_bits(np.ones((E,), np.float32))
  w: Dict[str, np.ndarray] = {}
  w["params/token_embedder/embedding"] = f32_to_bf16_bits(rng.standard_normal(V * E, dtype=np.float32)).reshape(V, E)
  for i in range(L):
    p = layer_paths(i)
    w[p["ln1"]] = one.copy()
    w[p["q"]] = _dense(rng, (E, Q, D), E, math.sqrt(D))
    w[p["k"]] = _dense(rng, (E, KV, D), E)
    w[p["v"]] = _dense(rng, (E, KV, D), E)
    w[p["o"]] = _dense(rng, (Q, D, E), Q * D)
    w[p["ln2"]] = one.copy()
    w[p["wi0"]] = _dense(rng, (E, F), E)
    w[p["wi1"]] = _dense(rng, (E, F), E)
    w[p["wo"]] = _dense(rng, (F, E), F)
  w["params/decoder/decoder_norm/scale"] = one.copy()
  w["params/decoder/logits_dense/kernel"] = _dense(rng, (E, V), E)
  return w


def audio_id_to_row(a: np.ndarray) -> np.ndarray:
  """audio id in [0, 8192) -> embedding row (audio_token_mapping_adjusted.json; extend_tokenizer.py:29-46)."""
  a = np.asarray(a, np.int64)
  return np.where(a < 99, 6 + a, 256001 + (a - 99)).astype(np.int32)


def synth_prompt(vocab_size: int, prefill_length: int, seed: int = 1234, n_text: int = 46):
  """SURVEY.md section 8d prompt: [BT] + n_text text ids + [BA] + audio rows, padded to prefill_length.
  Returns (padded int32[prefill_length], true_length).  Small vocabularies (test models) get uniform ids."""
  rng = np.random.Generator(np.random.PCG64(seed))
  true_length = int(prefill_length - rng.integers(0, 9)) if prefill_length >= 16 else int(prefill_length)
  if vocab_size >= 264192:
    n_audio = true_length - n_text - 2
    text = rng.integers(106, 256000, size=n_text)
    audio = audio_id_to_row(rng.integers(0, 8192, size=n_audio))
    ids = np.concatenate([[264094], text, [264095], audio])
  else:
    ids = rng.integers(1, vocab_size, size=true_length)
  out = np.zeros((prefill_length,), np.int32)
  out[:true_length] = ids[:true_length]
  return out, true_length


# ----------------------------------------------------------------------------- full-size trees drawn on the device
class DeviceTree:
  """{MaxText tree path: bf16 CUDA tensor} for MaxEngine.load_params, each tensor drawn (truncated normal of the kernel's fan-in; norm scales 1)
  when load_params asks for it and dropped afterwards: the 5 GB Gemma-2 2B and 93 GB Mixtral-8x7B trees load in well under a second, where the
  host-side generator above needs minutes.  Same values for the same (spec, seed) on the same torch build; NOT the values of synth_weights."""

  def __init__(self, spec, seed: int = 0):
    self.spec, self.seed = spec, seed

  def items(self):
    import torch  # pylint: disable=import-outside-toplevel
    g = torch.Generator(device="cuda")
    for i, (name, shape, std) in enumerate(self.spec):
      g.manual_seed(self.seed * 100003 + i)
      if std is None:
        yield name, torch.ones(shape, dtype=torch.bfloat16, device="cuda")
      else:
        t = torch.empty(shape, dtype=torch.float32, device="cuda").normal_(0.0, std, generator=g).clamp_(-2 * std, 2 * std)
        yield name, t.to(torch.bfloat16)
        del t


def device_tree_spec(cfg):
  """[(tree path, shape, std or None for a norm scale)] of the configured block (llama2, gemma2, mixtral)."""
  E, Q, KV, D, F, V = cfg.emb_dim, cfg.num_query_heads, cfg.num_kv_heads, cfg.head_dim, cfg.mlp_dim, cfg.vocab_size
  block = getattr(cfg, "decoder_block", "llama2")
  spec = [("params/token_embedder/embedding", (V, E), 1.0)]

  def attn(p):
    return [(p["q"], (E, Q, D), 1 / math.sqrt(E * D)), (p["k"], (E, KV, D), 1 / math.sqrt(E)), (p["v"], (E, KV, D), 1 / math.sqrt(E)),
            (p["o"], (Q, D, E), 1 / math.sqrt(Q * D))]

  for i in range(cfg.num_decoder_layers):
    if block == "gemma2":
      for sub in ("local", "global"):
        p = gemma2_layer_paths(i, sub)
        spec += [(p["ln1"], (E,), None)] + attn(p) + [(p["ln2"], (E,), None), (p["wi0"], (E, F), 1 / math.sqrt(E)), (p["wi1"], (E, F), 1 / math.sqrt(E)),
                                                       (p["wo"], (F, E), 1 / math.sqrt(F))]
        if getattr(cfg, "use_post_attn_norm", False):
          spec.append((p["pan"], (E,), None))
        if getattr(cfg, "use_post_ffw_norm", False):
          spec.append((p["pfn"], (E,), None))
    elif block == "mixtral":
      p = layer_paths(i)
      b = f"params/decoder/layers_{i}/"
      NE = cfg.num_experts
      spec += [(p["ln1"], (E,), None)] + attn(p) + [(b + "post_self_attention_layer_norm/scale", (E,), None), (b + "MoeBlock_0/gate/kernel", (E, NE), 4 / math.sqrt(E)),
                                                     (b + "MoeBlock_0/wi_0", (NE, E, F), 1 / math.sqrt(E)), (b + "MoeBlock_0/wi_1", (NE, E, F), 1 / math.sqrt(E)),
                                                     (b + "MoeBlock_0/wo", (NE, F, E), 1 / math.sqrt(F))]
    else:
      p = layer_paths(i)
      spec += [(p["ln1"], (E,), None)] + attn(p) + [(p["ln2"], (E,), None), (p["wi0"], (E, F), 1 / math.sqrt(E)), (p["wi1"], (E, F), 1 / math.sqrt(E)),
                                                     (p["wo"], (F, E), 1 / math.sqrt(F))]
  spec.append(("params/decoder/decoder_norm/scale", (E,), None))
  if not getattr(cfg, "logits_via_embedding", False):
    spec.append(("params/decoder/logits_dense/kernel", (E, V), 1 / math.sqrt(E)))
  return spec


def device_tree(cfg, seed: int = 0) -> DeviceTree:
  return DeviceTree(device_tree_spec(cfg), seed)
