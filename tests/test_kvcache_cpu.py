"""inference.kvcache.KVCache (the reference's dense two-segment cache, kept by name as a byte container): its state semantics on CPU
tensors -- prefill store, chunked append at next_pos, the ring write at one shared index -- against the facts the reference's own cache
test reads back (tests/inference/kvcache_test.py:25-148: shapes, and the stored values at the written positions)."""
import importlib

import numpy as np
import torch

KV = importlib.import_module("maxtext-indextts2_amd.inference.kvcache")
CT = importlib.import_module("maxtext-indextts2_amd.common_types")


def _mk(**kw):
  return KV.KVCache(max_prefill_length=8, max_target_length=12, batch=3, key_seq_len=8, value_seq_len=8, key_heads=2, value_heads=2, key_head_size=4,
                    value_head_size=4, dtype=torch.bfloat16, device="cpu", **kw)


def test_prefill_then_autoregressive_ring():
  c = _mk()
  assert c.cached_prefill_key.shape == (3, 8, 2, 4) and c.cached_ar_key.shape == (3, 4, 2, 4)
  k = torch.randn(3, 8, 2, 4).to(torch.bfloat16)
  v = torch.randn(3, 8, 2, 4).to(torch.bfloat16)
  seg = torch.tensor([[1] * 8, [1] * 5 + [0] * 3, [1] * 2 + [0] * 6], dtype=torch.int32)
  (pk, pv, ps), ar = c(k, v, seg, CT.MODEL_MODE_PREFILL)
  assert ar is None and torch.equal(pk, k) and torch.equal(c.cached_prefill_value, v) and torch.equal(c.cache_prefill_segment_id, seg)
  for step in range(6):          # 4 ring slots: steps 4, 5 overwrite slots 0, 1
    nk = torch.full((3, 1, 2, 4), float(step + 1)).to(torch.bfloat16)
    (pk, pv, ps), (ak, av, aseg, alen) = c(nk, nk * 2, None, CT.MODEL_MODE_AUTOREGRESSIVE)
    assert torch.equal(ak[:, step % 4], nk[:, 0]) and torch.equal(av[:, step % 4], (nk * 2)[:, 0])
    assert c.cache_ar_index == (step + 1) % 4 and int(alen[0]) == step + 1
    assert torch.equal(aseg[:, :min(step + 1, 4)], torch.ones((3, min(step + 1, 4)), dtype=torch.int32))
  assert torch.equal(pk, k)      # the prefill segment is untouched by decode


def test_chunked_prefill_appends_at_next_pos():
  c = _mk(use_chunked_prefill=True)
  k1, k2 = torch.randn(1, 3, 2, 4).to(torch.bfloat16), torch.randn(1, 4, 2, 4).to(torch.bfloat16)
  (a, _, s1), _ = c(k1, k1, torch.ones((1, 3), dtype=torch.int32), CT.MODEL_MODE_PREFILL)
  assert a.shape[1] == 3
  (b, _, s2), _ = c(k2, k2, torch.tensor([[1, 1, 1, 0]], dtype=torch.int32), CT.MODEL_MODE_PREFILL, previous_chunk=np.zeros((1, 3)))
  assert b.shape[1] == 7 and torch.equal(b[:, :3], k1) and torch.equal(b[:, 3:], k2)      # tests/maxengine_test.py:166-300: chunks concatenate exactly
  assert s2.tolist() == [[1, 1, 1, 1, 1, 1, 0]]
