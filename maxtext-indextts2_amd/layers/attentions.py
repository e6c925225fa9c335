"""MaxText/layers/attentions.py: Attention (:1524-2321; __call__ :2179), AttentionOp (:249-1466), attention_as_linen (:1469-1521)."""
from __future__ import annotations

from typing import Any, Optional

import torch

from .. import runtime
from ..common_types import MODEL_MODE_AUTOREGRESSIVE, MODEL_MODE_PREFILL
from ..inference.paged_attention import PagedAttentionOp
from .embeddings import RotaryEmbedding
from .linears import DenseGeneral


class AttentionOp:
  """attentions.py:249: the dense dot-product family (apply_attention_dot :1206, two-segment merge :1376-1466).  This build attends over
  pages only (attention="paged": PagedAttentionOp); the dense kernels exist in the reference-algorithm CPU baseline, not here."""

  def __init__(self, *args, **kwargs):
    self.args, self.kwargs = args, kwargs

  def __call__(self, *args, **kwargs):
    raise NotImplementedError("dense AttentionOp is outside the built path: configure attention='paged'")


class Attention:
  """attentions.py:1524-2321: q / k / v DenseGeneral (:2240-2242; the query kernel carries the 1/sqrt(head_dim) of :1900-1904), RoPE on q and
  k (:2254-2256), PagedAttentionOp (:2298-2302), out projection over (heads, head_dim) (:2319)."""

  def __init__(self, config: Any, num_query_heads: int, num_kv_heads: int, head_dim: int, max_target_length: int, mesh: Any = None,
               attention_kernel: str = "paged", inputs_q_shape=None, inputs_kv_shape=None, dtype: Any = None, weight_dtype: Any = None,
               max_prefill_predict_length: int = -1, dropout_rate: float = 0.0, name: Optional[str] = None, layer_idx: int = 0, **kwargs):
    del mesh, dtype, weight_dtype, dropout_rate, kwargs
    if attention_kernel not in ("paged", "autoselected"):
      raise NotImplementedError(f"attention_kernel={attention_kernel}: only 'paged' is built")
    self.config, self.name, self.layer_idx = config, name, layer_idx
    self.num_query_heads, self.num_kv_heads, self.head_dim = num_query_heads, num_kv_heads, head_dim
    emb = int(inputs_q_shape[-1]) if inputs_q_shape is not None else int(config.emb_dim)
    self.query = DenseGeneral(emb, (num_query_heads, head_dim), config=config, name="query")
    self.key = DenseGeneral(emb, (num_kv_heads, head_dim), config=config, name="key")
    self.value = DenseGeneral(emb, (num_kv_heads, head_dim), config=config, name="value")
    self.out = DenseGeneral((num_query_heads, head_dim), emb, axis=(-2, -1), config=config, name="out")
    self.rotary_embedding = RotaryEmbedding(config.rope_min_timescale, config.rope_max_timescale, head_dim, config=config, name="rotary_embedding")
    self.paged_attention_op = PagedAttentionOp(
        mesh=None, num_pages=config.pagedattn_num_pages, tokens_per_page=config.pagedattn_tokens_per_page,
        max_pages_per_slot=config.pagedattn_max_pages_per_group, max_pages_per_prefill=-1, pages_per_compute_block=config.pagedattn_pages_per_compute_block,
        num_kv_heads=num_kv_heads, kv_head_dim_size=head_dim, config=config, layer=layer_idx)
    self.attention_op = AttentionOp()

  def __call__(self, inputs_q: torch.Tensor, inputs_kv: torch.Tensor, inputs_positions: Optional[torch.Tensor] = None, decoder_segment_ids: Any = None, *,
               model_mode: str = MODEL_MODE_PREFILL, deterministic: bool = False, previous_chunk: Any = None, slot: Optional[int] = None,
               page_state: Any = None, bidirectional_mask: Any = None):
    del deterministic, bidirectional_mask
    query = self.query(inputs_q)
    key = self.key(inputs_kv)
    value = self.value(inputs_kv)
    query = self.rotary_embedding(query, inputs_positions)
    key = self.rotary_embedding(key, inputs_positions)
    out, _, _ = self.paged_attention_op(query, key, value, decoder_segment_ids, model_mode, previous_chunk, slot=slot, page_state=page_state,
                                        positions=inputs_positions)
    return self.out(out)


def attention_as_linen(*, config: Any, num_query_heads: int, num_kv_heads: int, head_dim: int, max_target_length: int, mesh: Any = None,
                       attention_kernel: str = "paged", inputs_q_shape=None, inputs_kv_shape=None, name: Optional[str] = None, **kwargs) -> Attention:
  """attentions.py:1469-1521."""
  return Attention(config, num_query_heads, num_kv_heads, head_dim, max_target_length, mesh, attention_kernel, inputs_q_shape, inputs_kv_shape, name=name, **kwargs)
