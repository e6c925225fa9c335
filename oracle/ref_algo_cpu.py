"""CPU BASELINE of the reference ALGORITHM (test / measurement infrastructure, NOT product code, NOT the parity oracle).

BASELINE.md section 3 / SURVEY.md section 8(d): the north star's CPU baseline is the reference on `jax.devices('cpu')`.  JAX is not
installed in this pipeline, so the timed stand-in is this PyTorch-CPU restatement of what the reference *does* on that path -- the
data movement and arithmetic volume that make up its time, not the bit patterns (those belong to oracle/ixt_oracle.c):

  * dense two-segment KV cache: a prefill segment [B, P] and an autoregressive segment [B, T - P] per layer
    (MaxText/inference/kvcache.py:490-831); the AR write index is ONE scalar shared by the whole batch (:481-484, :696-701, :778);
  * every decode step attends over BOTH full fixed-size buffers, masked by segment ids, and merges the two partial softmaxes with
    the (max, sum) rule (MaxText/layers/attentions.py:1376-1397, 1399-1466) -- O(T) per step however many tokens exist;
  * prefill computes logits for all P rows, [1, P, V], and slices one (MaxText/maxengine.py:462-487);
  * llama2 block wiring (layers/llama2.py:54-165), RMSNorm in f32 (normalizations.py:57-70), half-split RoPE (embeddings.py:277-315),
    SwiGLU MLP (linears.py:425-476), untied logits (decoders.py:566-587), greedy argmax (inference_utils.py:74-75);
  * bf16 storage, f32 accumulation: weights hold bf16 VALUES; `matmul="f32"` keeps them in f32 arrays (every bf16 x bf16 product is exact
    in f32, accumulation in f32 -- any CPU), `matmul="bf16"` keeps bf16 arrays and uses torch's bf16 GEMM (fast only with AMX / AVX512-BF16).

Only bench.py's cpu_baseline leg and tests/ may import this module.
"""
from __future__ import annotations

import time
from typing import Dict

import numpy as np
import torch


def _bf16(a: np.ndarray) -> torch.Tensor:
  """numpy uint16 bit patterns -> torch bf16 (no copy of values beyond the view)."""
  return torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).view(torch.bfloat16)


class RefAlgoCPU:
  def __init__(self, cfg, weights: Dict[str, np.ndarray], batch: int, prefill_len: int, target_len: int, matmul: str = "f32", threads: int = 0):
    if threads > 0:
      torch.set_num_threads(threads)
    self.threads = torch.get_num_threads()
    self.B, self.P, self.T = batch, prefill_len, target_len
    self.E, self.H, self.KV, self.D = cfg.emb_dim, cfg.num_query_heads, cfg.num_kv_heads, cfg.head_dim
    self.F, self.V, self.L = cfg.mlp_dim, cfg.vocab_size, cfg.num_decoder_layers
    self.eps = float(cfg.normalization_layer_epsilon)
    self.wdt = torch.float32 if matmul == "f32" else torch.bfloat16
    self.matmul = matmul
    cv = lambda name, shape: _bf16(weights[name]).reshape(shape).to(self.wdt)
    self.emb = _bf16(weights["params/token_embedder/embedding"]).reshape(self.V, self.E)
    self.final_norm = _bf16(weights["params/decoder/decoder_norm/scale"]).float()
    self.logits_w = cv("params/decoder/logits_dense/kernel", (self.E, self.V))
    self.layers = []
    for i in range(self.L):
      p = f"params/decoder/layers_{i}/"
      self.layers.append({
          "ln1": _bf16(weights[p + "pre_self_attention_layer_norm/scale"]).float(),
          "wq": cv(p + "self_attention/query/kernel", (self.E, self.H * self.D)),
          "wk": cv(p + "self_attention/key/kernel", (self.E, self.KV * self.D)),
          "wv": cv(p + "self_attention/value/kernel", (self.E, self.KV * self.D)),
          "wo": cv(p + "self_attention/out/kernel", (self.H * self.D, self.E)),
          "ln2": _bf16(weights[p + "mlp/mlp_layer_norm/scale"]).float(),
          "wi0": cv(p + "mlp/wi_0/kernel", (self.E, self.F)), "wi1": cv(p + "mlp/wi_1/kernel", (self.E, self.F)),
          "wout": cv(p + "mlp/wo/kernel", (self.F, self.E)),
      })
    half = self.D // 2
    lo, hi = float(getattr(cfg, "rope_min_timescale", 1)), float(getattr(cfg, "rope_max_timescale", 10000))
    ts = (lo * ((hi / lo) ** (torch.arange(half, dtype=torch.float32) * 2 / self.D)))       # embeddings.py:254-268
    self.llama31 = str(getattr(cfg, "rope_type", "default")).lower().startswith("llama3.1")
    if self.llama31 and bool(getattr(cfg, "rope_use_scale", True)):      # embeddings.py:417-443
      freq = 1.0 / ts
      wavelen = 2 * torch.pi / freq
      smooth = (8192.0 / wavelen - 1.0) / (4.0 - 1.0)
      freq = torch.where(wavelen < 8192.0 / 4.0, freq, torch.where(wavelen > 8192.0, freq / 8.0, (1 - smooth) * freq / 8.0 + smooth * freq))
      ts = 1.0 / freq
    ang = torch.arange(target_len, dtype=torch.float32)[:, None] / ts[None, :]
    if self.llama31:
      self.sin, self.cos = torch.sin(ang), torch.cos(ang)                # f32 tables (embeddings.py:474-479)
    else:
      self.sin, self.cos = torch.sin(ang).to(torch.bfloat16), torch.cos(ang).to(torch.bfloat16)
    A = target_len - prefill_len
    z = lambda n: torch.zeros((self.L, batch, n, self.KV, self.D), dtype=torch.bfloat16)
    self.pk, self.pv, self.ak, self.av = z(prefill_len), z(prefill_len), z(A), z(A)      # cached_prefill_key/value, cached_ar_key/value
    self.pseg = torch.zeros((batch, prefill_len), dtype=torch.int32)                     # cache_prefill_segment_id
    self.aseg = torch.zeros((batch, A), dtype=torch.int32)                               # cache_ar_segment_id
    self.ar_index = 0                                                                    # cache_ar_index: one scalar for the batch
    self.next_pos = torch.zeros((batch,), dtype=torch.int64)
    self.tokens = torch.zeros((batch,), dtype=torch.int64)

  # ---- pieces
  def _mm(self, x, w):     # bf16 in, f32 accumulate, bf16 out (linears.py:93, :216)
    return (x.to(self.wdt) @ w).to(torch.bfloat16)

  def _norm(self, x, scale):
    xf = x.float()
    y = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + self.eps)).to(torch.bfloat16)
    return (y.float() * scale).to(torch.bfloat16)

  def _rope(self, x, pos):   # x [..., n, heads, D], pos [..., n]
    if self.llama31:         # embeddings.py:446-488, written the reference's way: shifted copies, a parity select, a sign vector
      s = self.sin[pos].unsqueeze(-2).repeat_interleave(2, dim=-1)
      c = self.cos[pos].unsqueeze(-2).repeat_interleave(2, dim=-1)
      left = torch.cat([x[..., 1:], x[..., :1]], -1)
      right = torch.cat([x[..., -1:], x[..., :-1]], -1)
      odd = (torch.arange(self.D) % 2).bool()
      shifted = torch.where(odd, right, left)
      sign = torch.tensor([-1.0, 1.0]).repeat(self.D // 2)
      return (x.float() * c + shifted.float() * s * sign).to(torch.bfloat16)
    half = self.D // 2
    s, c = self.sin[pos].unsqueeze(-2), self.cos[pos].unsqueeze(-2)
    x1, x2 = x[..., :half], x[..., half:]
    return torch.cat([(x1 * c).to(torch.bfloat16) - (x2 * s).to(torch.bfloat16), (x2 * c).to(torch.bfloat16) + (x1 * s).to(torch.bfloat16)], -1)

  def _mlp(self, lw, x):
    a, b = self._mm(x, lw["wi0"]), self._mm(x, lw["wi1"])
    return self._mm((torch.nn.functional.silu(a.float()).to(torch.bfloat16) * b), lw["wout"])

  def _attend(self, q, k, v, mask):
    """q [B, t, H, D], k/v [B, s, KV, D], mask [B, t, s] bool -> un-normalised (O f32, m, l) (attentions.py:1206-1271)."""
    g = self.H // self.KV
    B, t, s = q.shape[0], q.shape[1], k.shape[1]
    cd = self.wdt                                      # bf16 products on CPUs with AMX / AVX512-BF16 (2.5x the f32 path here), else f32
    qg = q.to(cd).reshape(B, t, self.KV, g, self.D)
    sc = torch.einsum("btkgd,bskd->bkgts", qg, k.to(cd)).float()
    sc = sc.masked_fill(~mask[:, None, None, :, :], -0.7 * 3.4028235e38)
    m = sc.amax(-1, keepdim=True)
    e = torch.exp(sc - m)
    l = e.sum(-1, keepdim=True)
    o = torch.einsum("bkgts,bskd->btkgd", e.to(cd), v.to(cd)).float()
    return o, m.permute(0, 3, 1, 2, 4), l.permute(0, 3, 1, 2, 4)

  # ---- engine surface
  def prefill(self, tokens: np.ndarray, true_length: int, slot: int) -> int:
    """maxengine.py:400-530 with the dense cache: all P (padded) rows are computed, logits for all of them, one row sliced."""
    P = self.P
    tok = torch.from_numpy(np.asarray(tokens[:P], np.int64))
    pos = torch.arange(P)
    seg = (pos < true_length).to(torch.int32)
    h = self.emb[tok].unsqueeze(0)                                                  # [1, P, E]
    causal = (pos[None, :] <= pos[:, None]) & (seg[None, :] == seg[:, None])       # attentions.py:506-643 (segment equality and causality)
    for li, lw in enumerate(self.layers):
      x = self._norm(h, lw["ln1"])
      q = self._rope(self._mm(x, lw["wq"]).reshape(1, P, self.H, self.D), pos[None])
      k = self._rope(self._mm(x, lw["wk"]).reshape(1, P, self.KV, self.D), pos[None])
      v = self._mm(x, lw["wv"]).reshape(1, P, self.KV, self.D)
      self.pk[li, slot], self.pv[li, slot] = k[0], v[0]                            # kvcache.py:584-624 (insert copies it; here in place)
      o, _, l = self._attend(q, k, v, causal[None])
      att = (o / l).reshape(1, P, self.H * self.D).to(torch.bfloat16)
      h = h + self._mm(att, lw["wo"])
      h = h + self._mlp(lw, self._norm(h, lw["ln2"]))
    logits = (self._norm(h, self.final_norm).to(self.wdt) @ self.logits_w).float()   # [1, P, V]: every row (maxengine.py:462-476)
    first = int(torch.argmax(logits[0, true_length - 1]))
    self.last_logits = logits[0, true_length - 1:true_length].clone()               # [1, V] (tests/test_cross_oracle_cpu.py)
    self.pseg[slot] = seg
    self.aseg[slot] = 0
    self.next_pos[slot], self.tokens[slot] = true_length, first
    return first

  def generate(self) -> np.ndarray:
    """maxengine.py:868-936 + attentions.py:1399-1466: one token for all B slots; both cache segments are read in full."""
    B, A = self.B, self.T - self.P
    pos = self.next_pos.clone()
    h = self.emb[self.tokens].unsqueeze(1)                                          # [B, 1, E]
    idx = self.ar_index
    self.aseg[:, idx] = 1                                                           # kvcache.py:774-777
    pmask, amask = (self.pseg == 1)[:, None, :], (self.aseg == 1)[:, None, :]      # attentions.py:587-588
    for li, lw in enumerate(self.layers):
      x = self._norm(h, lw["ln1"])
      q = self._rope(self._mm(x, lw["wq"]).reshape(B, 1, self.H, self.D), pos[:, None])
      k = self._rope(self._mm(x, lw["wk"]).reshape(B, 1, self.KV, self.D), pos[:, None])
      v = self._mm(x, lw["wv"]).reshape(B, 1, self.KV, self.D)
      self.ak[li, :, idx], self.av[li, :, idx] = k[:, 0], v[:, 0]                  # the shared ring index (kvcache.py:696-701)
      o1, m1, l1 = self._attend(q, self.pk[li], self.pv[li], pmask)
      o2, m2, l2 = self._attend(q, self.ak[li], self.av[li], amask)
      m = torch.maximum(m1, m2)                                                     # normalize_attention (attentions.py:1376-1397)
      f1, f2 = torch.exp(m1 - m), torch.exp(m2 - m)
      att = ((o1 * f1 + o2 * f2) / (l1 * f1 + l2 * f2)).reshape(B, 1, self.H * self.D).to(torch.bfloat16)
      h = h + self._mm(att, lw["wo"])
      h = h + self._mlp(lw, self._norm(h, lw["ln2"]))
    logits = (self._norm(h, self.final_norm).to(self.wdt) @ self.logits_w).float()   # [B, 1, V]
    new = torch.argmax(logits[:, 0], dim=-1)
    self.last_logits = logits[:, 0]                                                 # [B, V]
    self.tokens, self.next_pos = new, self.next_pos + 1
    self.ar_index = (idx + 1) % A                                                   # kvcache.py:778
    return new.numpy()


def measure(cfg, weights, prompts, batch: int, prefill_len: int, decode_steps: int, slice_steps: int = 2, matmul: str = "auto", threads: int = 0,
            log=lambda m: None):
  """Bounded sample of the bench workload on the CPU: `batch` utterances, one timed prefill, decode slices of `slice_steps` steps
  at three ring positions (the dense cache reads both full buffers at every step, so the step time does not depend on the context;
  the three slices show exactly that), integrated over the whole pass.  Returns a dict for bench.py's cpu_baseline."""
  T = prefill_len + decode_steps
  if matmul == "auto":
    flags = open("/proc/cpuinfo", encoding="utf-8").read() if True else ""
    matmul = "bf16" if ("amx_bf16" in flags or "avx512_bf16" in flags) else "f32"
  eng = RefAlgoCPU(cfg, weights, batch, prefill_len, T, matmul=matmul, threads=threads)
  log(f"reference-algorithm engine built ({matmul} matmul)")
  t0 = time.perf_counter()
  firsts = [eng.prefill(prompts[0][0], prompts[0][1], 0)]
  t_pre = time.perf_counter() - t0
  log(f"cpu prefill {t_pre:.2f} s")
  for s in range(1, batch):   # the other slots reuse the timed prefill's cache rows (same cost per utterance; not timed again)
    eng.pk[:, s], eng.pv[:, s], eng.pseg[s] = eng.pk[:, 0], eng.pv[:, 0], eng.pseg[0]
    eng.next_pos[s], eng.tokens[s] = eng.next_pos[0], eng.tokens[0]
  slices = []
  A = T - prefill_len
  for ring_pos in (0, A // 2, max(0, A - slice_steps - 1)):
    eng.ar_index = ring_pos
    eng.generate()   # warm
    t1 = time.perf_counter()
    for _ in range(slice_steps):
      eng.generate()
    slices.append((time.perf_counter() - t1) / slice_steps)
    log(f"cpu decode slice at ring position {ring_pos}: {slices[-1]:.3f} s/step")
  t_step = float(np.mean(slices))
  total = batch * t_pre + decode_steps * t_step
  return {"value": batch * (decode_steps + 1) / total, "unit": "tokens/s", "cores": eng.threads, "kind": "reference-algorithm", "matmul": matmul,
          "prefill_s_per_utterance": t_pre, "decode_s_per_step": t_step, "decode_slices_s_per_step": slices, "batch": batch,
          "first_token": firsts[0]}
