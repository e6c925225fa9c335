"""MaxText/inference/kvcache.py: KVCache (:229-831), the dense two-segment cache, as a torch container.

The hot path of this build keeps K/V in pages (inference/paged_attention.py); this class keeps the reference's dense-cache NAME and STATE
SEMANTICS for callers that hold one: a prefill segment [batch, max_prefill_length] and an autoregressive ring segment
[batch, max_target_length - max_prefill_length] per layer, chunked-prefill append at `next_pos = previous_chunk.shape[1]` (:490-582), the
ring write at ONE index shared by the batch (:481-484, :696-701), the segment-id marks (:774-777) and the index / length updates
(:778-779).  It moves bytes and integers only (no arithmetic); the values stay bit for bit what was stored.  Logical layout of everything
returned: [batch, sequence, heads, head_dim] (the reference stores a permuted layout, :259-261, and returns this one, :720-736)."""
from __future__ import annotations

from typing import Any, Optional

import torch

from ..common_types import DECODING_ACTIVE_SEQUENCE_INDICATOR, MODEL_MODE_AUTOREGRESSIVE, MODEL_MODE_PREFILL


class KVCache:
  def __init__(self, max_prefill_length: int, max_target_length: int, batch: int, key_seq_len: int, value_seq_len: int, key_heads: int, value_heads: int,
               key_head_size: int, value_head_size: int, dtype: Any = torch.bfloat16, kv_quant: Any = None, prefill_cache_axis_order=(1, 2, 0, 3),
               ar_cache_axis_order=(1, 2, 0, 3), key_axis_order=(2, 0, 1, 3), use_chunked_prefill: bool = False, model_mode: str = MODEL_MODE_PREFILL,
               *, rngs: Any = None, device: Any = None, **kwargs):
    del key_seq_len, value_seq_len, prefill_cache_axis_order, ar_cache_axis_order, key_axis_order, rngs, kwargs
    if kv_quant is not None:
      raise NotImplementedError("KV-cache quantisation is outside the built path")     # kvcache.py:512 refuses it for chunked prefill too
    dev = device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu")
    self.max_prefill_length, self.max_target_length, self.batch = max_prefill_length, max_target_length, batch
    self.use_chunked_prefill, self.model_mode = use_chunked_prefill, model_mode
    P, A = max_prefill_length, max_target_length - max_prefill_length
    z = lambda n, h, d: torch.zeros((batch, n, h, d), dtype=dtype, device=dev)
    self.cached_prefill_key, self.cached_prefill_value = z(P, key_heads, key_head_size), z(P, value_heads, value_head_size)
    self.cache_prefill_segment_id = torch.zeros((batch, P), dtype=torch.int32, device=dev)
    self.cached_ar_key, self.cached_ar_value = z(A, key_heads, key_head_size), z(A, value_heads, value_head_size)
    self.cache_ar_segment_id = torch.zeros((batch, A), dtype=torch.int32, device=dev)
    self.cache_ar_index = 0                                                        # one scalar for the whole batch (:481-484)
    self.cached_ar_lengths = torch.zeros((batch,), dtype=torch.int32, device=dev)

  def kv_cache_prefill(self, key, value, decoder_segment_ids):
    """kvcache.py:584-624."""
    b, s = key.shape[0], key.shape[1]
    self.cached_prefill_key[:b, :s], self.cached_prefill_value[:b, :s] = key, value
    if decoder_segment_ids is not None:
      self.cache_prefill_segment_id[:b, :s] = decoder_segment_ids
    return key, value, decoder_segment_ids

  def kv_cache_chunked_prefill(self, key, value, decoder_segment_ids, previous_chunk: Optional[Any] = None):
    """kvcache.py:490-582: append at next_pos, return the first next_pos + chunk rows."""
    next_pos = 0 if previous_chunk is None else int(previous_chunk.shape[1])
    s = key.shape[1]
    self.cached_prefill_key[:1, next_pos:next_pos + s], self.cached_prefill_value[:1, next_pos:next_pos + s] = key, value
    seg = torch.zeros_like(self.cache_prefill_segment_id[:1])
    seg[:, :next_pos] = DECODING_ACTIVE_SEQUENCE_INDICATOR                         # the earlier chunks hold true tokens only (:554-563)
    if decoder_segment_ids is not None:
      seg[:, next_pos:next_pos + s] = decoder_segment_ids
    self.cache_prefill_segment_id[:1] = seg
    n = next_pos + s
    return self.cached_prefill_key[:1, :n], self.cached_prefill_value[:1, :n], self.cache_prefill_segment_id[:1, :n]

  def kv_cache_autoregressive(self, key, value, use_ragged_attention: bool = False):
    """kvcache.py:738-795."""
    del use_ragged_attention
    if value.shape[1] != 1:
      raise ValueError(f"Sequence length should be 1 during autoregression, got sequence={value.shape[1]}")
    i = self.cache_ar_index
    self.cached_ar_key[:, i], self.cached_ar_value[:, i] = key[:, 0], value[:, 0]
    self.cache_ar_segment_id[:, i] = DECODING_ACTIVE_SEQUENCE_INDICATOR
    self.cache_ar_index = (i + 1) % (self.max_target_length - self.max_prefill_length)
    self.cached_ar_lengths += 1
    return ((self.cached_prefill_key, self.cached_prefill_value, self.cache_prefill_segment_id),
            (self.cached_ar_key, self.cached_ar_value, self.cache_ar_segment_id, self.cached_ar_lengths))

  def __call__(self, key, value, decoder_segment_ids=None, model_mode: str = MODEL_MODE_PREFILL, use_ragged_attention: bool = False, previous_chunk: Any = None):
    """kvcache.py:804-831."""
    if model_mode == MODEL_MODE_PREFILL:
      if self.use_chunked_prefill:
        return self.kv_cache_chunked_prefill(key, value, decoder_segment_ids, previous_chunk), None
      return self.kv_cache_prefill(key, value, decoder_segment_ids), None
    if model_mode == MODEL_MODE_AUTOREGRESSIVE:
      return self.kv_cache_autoregressive(key, value, use_ragged_attention)
    raise ValueError(f"Model Mode isn't supported! model_mode={model_mode!r}")
