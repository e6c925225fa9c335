"""f32 PyTorch-CPU ORACLE of the training step (test infrastructure, NOT product code).

Restates MaxText/train.py:255-472 (loss_fn, train_step), the llama2 block stack in train mode (layers/llama2.py:54-165 and what it calls),
max_utils.cross_entropy_with_logits (max_utils.py:563-654), optax.clip_by_global_norm (maxtext_utils.py:825-846), optax.adamw
(optimizers.py:28-38) and the warmup-cosine schedule (maxtext_utils.py:1245-1285) with autograd doing the backward pass.  Arithmetic is f32;
every place where the product stores bf16 (weights at use, activations, softmax probabilities, logits) rounds to bf16 in the FORWARD pass and
lets the gradient through unchanged, so the loss must agree with the HIP path to f32 summation-order noise (bar: 1e-4 relative) and the
gradients to the bf16 rounding of the product's backward signals.

PARITY UNPINNED against the reference itself (JAX is not installed; its tests hold no numeric fixture for the train step); the AdamW update is
cross-checked against torch.optim.AdamW and the schedule against its closed form in tests/test_train_oracle_cpu.py.
Only tests/ may import this module."""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch

EPS = 1e-8


class _RoundBF16(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x):
    return x.to(torch.bfloat16).to(torch.float32)

  @staticmethod
  def backward(ctx, g):
    return g


rb = _RoundBF16.apply


def rope_tables(D, n, min_ts=1.0, max_ts=10000.0):
  half = D // 2
  frac = (2 * np.arange(half, dtype=np.int32)).astype(np.float32) / np.float32(D)
  ts = (np.float64(min_ts) * np.power(np.float64(max_ts) / np.float64(min_ts), frac.astype(np.float64))).astype(np.float32)
  ang = (np.arange(n, dtype=np.float32)[:, None] / ts[None, :]).astype(np.float32)
  to = lambda a: torch.from_numpy(a.astype(np.float32)).to(torch.bfloat16).to(torch.float32)
  return to(np.sin(ang.astype(np.float64))), to(np.cos(ang.astype(np.float64)))


def learning_rate(config, step: int) -> float:
  lr = float(config.learning_rate)
  final = lr * float(config.cosine_learning_rate_final_fraction)
  sched = int(config.learning_rate_schedule_steps) if int(config.learning_rate_schedule_steps) > 0 else int(config.steps)
  warm = int(sched * float(config.warmup_steps_fraction))
  if step < warm:
    return lr * step / warm
  if step < sched:
    a = 0.5 * (math.cos(math.pi * (step - warm) / (sched - warm)) + 1.0)
    return lr * a + final * (1.0 - a)
  return 0.0


class TrainOracle:
  def __init__(self, config, params: Dict[str, np.ndarray]):
    self.cfg = config
    self.p = {k: torch.tensor(np.asarray(v, np.float32).reshape(-1), dtype=torch.float32, requires_grad=True) for k, v in params.items()}
    self.mu = {k: torch.zeros_like(v) for k, v in self.p.items()}
    self.nu = {k: torch.zeros_like(v) for k, v in self.p.items()}
    self.step = 0
    self.sin, self.cos = rope_tables(config.head_dim, config.max_target_length, float(config.rope_min_timescale), float(config.rope_max_timescale))

  # ---- forward (train mode)
  def _norm(self, x, scale):
    r = torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + float(self.cfg.normalization_layer_epsilon))
    return rb(rb(x * r) * rb(scale))

  def _rope(self, x, pos):   # x [B, S, H, D]
    half = x.shape[-1] // 2
    s, c = self.sin[pos].unsqueeze(-2), self.cos[pos].unsqueeze(-2)
    x1, x2 = x[..., :half], x[..., half:]
    return torch.cat([rb(rb(x1 * c) - rb(x2 * s)), rb(rb(x2 * c) + rb(x1 * s))], -1)

  def loss(self, batch):
    c = self.cfg
    E, H, D, F, V, L = c.emb_dim, c.num_query_heads, c.head_dim, c.mlp_dim, c.vocab_size, c.num_decoder_layers
    KVH = int(c.num_kv_heads)
    G = H // KVH                                     # grouped-query attention (attentions.py:1206-1271): G query heads share a kv head
    tok = torch.as_tensor(np.asarray(batch["inputs"]), dtype=torch.long)
    pos = torch.as_tensor(np.asarray(batch["inputs_position"]), dtype=torch.long)
    seg = torch.as_tensor(np.asarray(batch["inputs_segmentation"]), dtype=torch.long)
    B, S = tok.shape
    w = lambda name, shape: rb(self.p[name]).reshape(shape)
    h = rb(w("params/token_embedder/embedding", (V, E))[tok])
    idx = torch.arange(S)
    mask = (seg[:, :, None] == seg[:, None, :]) & (idx[None, None, :] <= idx[None, :, None])       # attentions.py:586-618
    for i in range(L):
      pre = f"params/decoder/layers_{i}/"
      xn = self._norm(h, self.p[pre + "pre_self_attention_layer_norm/scale"])
      q = self._rope(rb(xn @ w(pre + "self_attention/query/kernel", (E, H * D))).reshape(B, S, H, D), pos)
      k = self._rope(rb(xn @ w(pre + "self_attention/key/kernel", (E, KVH * D))).reshape(B, S, KVH, D), pos)
      v = rb(xn @ w(pre + "self_attention/value/kernel", (E, KVH * D))).reshape(B, S, KVH, D)
      if G > 1:                                      # query head h attends to kv head h // G
        k, v = k.repeat_interleave(G, dim=2), v.repeat_interleave(G, dim=2)
      sc = torch.einsum("bihd,bjhd->bhij", q, k).masked_fill(~mask[:, None], float("-inf"))
      p = rb(torch.softmax(sc, dim=-1))
      ao = rb(torch.einsum("bhij,bjhd->bihd", p, v)).reshape(B, S, H * D)
      h = rb(h + rb(ao @ w(pre + "self_attention/out/kernel", (H * D, E))))
      xn = self._norm(h, self.p[pre + "mlp/mlp_layer_norm/scale"])
      a, b = rb(xn @ w(pre + "mlp/wi_0/kernel", (E, F))), rb(xn @ w(pre + "mlp/wi_1/kernel", (E, F)))
      act = rb(rb(a * rb(torch.sigmoid(a))) * b)
      h = rb(h + rb(act @ w(pre + "mlp/wo/kernel", (F, E))))
    xn = self._norm(h, self.p["params/decoder/decoder_norm/scale"])
    logits = rb(xn @ w("params/decoder/logits_dense/kernel", (E, V)))
    tgt = torch.as_tensor(np.asarray(batch["targets"]), dtype=torch.long)
    wgt = (torch.as_tensor(np.asarray(batch["targets_segmentation"])) != 0).float()
    xent = torch.logsumexp(logits, dim=-1) - torch.gather(logits, -1, tgt[..., None])[..., 0]      # max_utils.py:563-590, z_loss = 0
    return (xent * wgt).sum(), wgt.sum()

  # ---- train_step
  def train_step(self, batch):
    c = self.cfg
    for v in self.p.values():
      v.grad = None
    total_loss, total_weights = self.loss(batch)
    total_loss.backward()
    denom = float(total_weights) + EPS
    grads = {k: v.grad / denom for k, v in self.p.items()}
    raw = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads.values()))
    clip = float(c.gradient_clipping_threshold)
    f = clip / raw if (clip > 0 and raw >= clip) else 1.0                                           # optax.clip_by_global_norm
    lr = learning_rate(c, self.step)
    self.step += 1
    b1, b2, eps, eps_root, wd = float(c.adam_b1), float(c.adam_b2), float(c.adam_eps), float(c.adam_eps_root), float(c.adam_weight_decay)
    with torch.no_grad():
      for k, pv in self.p.items():
        g = grads[k] * f
        self.mu[k] = b1 * self.mu[k] + (1 - b1) * g
        self.nu[k] = b2 * self.nu[k] + (1 - b2) * g * g
        mh, nh = self.mu[k] / (1 - b1 ** self.step), self.nu[k] / (1 - b2 ** self.step)
        pv -= lr * (mh / (torch.sqrt(nh + eps_root) + eps) + wd * pv)
    total_loss = total_loss.detach()
    return {"learning/loss": float(total_loss) / denom, "learning/total_weights": float(total_weights), "learning/raw_grad_norm": raw,
            "learning/grad_norm": raw * f, "learning/current_learning_rate": lr, "grads": grads}


# ---- stand-alone restatements pinned by the reference's own test cases (tests/test_reference_cases_cpu.py)
def cross_entropy_with_logits(logits: np.ndarray, targets_one_hot: np.ndarray, z_loss: float = 0.0):
  """max_utils.py:563-621 (the forward rule of the custom VJP, :594-621): (loss, z_loss term) per position, stabilised by the row maximum."""
  logits = np.asarray(logits)
  m = logits.max(axis=-1, keepdims=True)
  shifted = logits - m
  sum_exp = np.exp(shifted).sum(axis=-1, keepdims=True)
  log_softmax = shifted - np.log(sum_exp)
  loss = -(targets_one_hot * log_softmax).sum(axis=-1)
  log_z = (np.log(sum_exp) + m)[..., 0]
  total_z = z_loss * np.square(log_z)
  return loss + total_z, total_z


def cross_entropy_vjp(logits: np.ndarray, targets_one_hot: np.ndarray, z_loss: float, g: np.ndarray) -> np.ndarray:
  """max_utils.py:624-651: d loss / d logits under the custom VJP = g * ((1 + 2 z log_z) softmax - targets)."""
  m = logits.max(axis=-1, keepdims=True)
  e = np.exp(logits - m)
  s = e.sum(axis=-1, keepdims=True)
  log_z = np.log(s) + m
  return g[..., None] * ((1.0 + 2.0 * z_loss * log_z) * e / s - targets_one_hot)


def l2norm_pytree(tree) -> float:
  """max_utils.py:72-74: sqrt of the sum of squares over every leaf of a (nested dict / list) tree."""
  if isinstance(tree, dict):
    tree = list(tree.values())
  if isinstance(tree, (list, tuple)):
    return math.sqrt(sum(l2norm_pytree(t) ** 2 for t in tree))
  return math.sqrt(float(np.sum(np.square(np.asarray(tree, np.float64)))))


def apply_gradient_clipping(raw_grads: dict, clipping_threshold: float, keep_key: str = "_overwrite_with_gradient") -> dict:
  """maxtext_utils.py:825-846 = optax.clip_by_global_norm over everything except the fp8 statistics subtree (`keep_key`), which neither
  counts towards the norm nor is scaled: g * threshold / max(norm, threshold)."""
  body = {k: v for k, v in raw_grads.items() if k != keep_key}
  norm = l2norm_pytree(body)
  f = clipping_threshold / max(norm, clipping_threshold)

  def scale(t):
    if isinstance(t, dict):
      return {k: scale(v) for k, v in t.items()}
    return np.asarray(t, np.float32) * np.float32(f)
  out = scale(body)
  if keep_key in raw_grads:
    out[keep_key] = raw_grads[keep_key]
  return out
