"""MaxText/inference/paged_attention.py: PagedAttentionOp (:99-471) over the page pools of libixt.so.

Prefill writes the chunk's K/V into the slot's pages (the INTENDED semantics of :403-444 -- the reference overwrites the whole pool there,
a defect this build does not copy) and attends with the causal offset of ref_ragged_paged_attention
(paged_attention_kernel_v2.py:71-109); decode scatters the new token at the page manager's position (:446-471) and attends over the
slot's sequence.  Returns (output, None, None) as the reference does for paged attention (attentions.py:2302: already normalised)."""
from __future__ import annotations

from typing import Any, Optional

import numpy as np
import torch

from .. import runtime
from ..common_types import MODEL_MODE_AUTOREGRESSIVE, MODEL_MODE_PREFILL


class PagedAttentionOp:
  def __init__(self, mesh: Any, num_pages: int, tokens_per_page: int, max_pages_per_slot: int, max_pages_per_prefill: int, pages_per_compute_block: int,
               num_kv_heads: int, kv_head_dim_size: int, dtype: Any = None, attn_logits_soft_cap: Optional[float] = None, query_axis_names=None,
               kv_pages_axis_names=None, *, rngs: Any = None, config: Any = None, layer: int = 0):
    del mesh, max_pages_per_prefill, pages_per_compute_block, dtype, query_axis_names, kv_pages_axis_names, rngs
    if attn_logits_soft_cap:
      raise NotImplementedError("attn_logits_soft_cap is outside the built path")
    self.ctx = runtime.context(config)
    c = self.ctx.config
    if (num_pages, tokens_per_page, max_pages_per_slot, num_kv_heads, kv_head_dim_size) != (
        c.pagedattn_num_pages, c.pagedattn_tokens_per_page, c.pagedattn_max_pages_per_group, c.num_kv_heads, c.head_dim):
      raise ValueError("PagedAttentionOp: geometry differs from the context's page pools")
    self.num_pages, self.tokens_per_page, self.max_pages_per_slot = num_pages, tokens_per_page, max_pages_per_slot
    self.num_kv_heads, self.kv_head_dim_size, self.layer = num_kv_heads, kv_head_dim_size, layer
    self.kv_pages_shape = (num_kv_heads, num_pages, tokens_per_page, kv_head_dim_size)

  def __call__(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, decoder_segment_ids: Any, model_mode: str, previous_chunk: Any = None,
               slot: Optional[int] = None, page_state: Any = None, positions: Optional[torch.Tensor] = None):
    dev = query.device
    B, S, NQ, D = query.shape
    if model_mode == MODEL_MODE_PREFILL:
      if B != 1 or slot is None:
        raise ValueError("paged prefill takes one sequence and its slot (maxengine.py:444)")
      start = 0 if previous_chunk is None else int(previous_chunk.shape[1] if hasattr(previous_chunk, "shape") else len(previous_chunk))
      row_pos = (torch.arange(S, dtype=torch.int32, device=dev) + start) if positions is None else positions.reshape(-1).to(torch.int32)
      row_slot = torch.full((S,), int(slot), dtype=torch.int32, device=dev)
      if decoder_segment_ids is not None:
        row_active = (torch.as_tensor(decoder_segment_ids, device=dev).reshape(-1) != 0).to(torch.int32)
      else:
        row_active = torch.ones((S,), dtype=torch.int32, device=dev)
    elif model_mode == MODEL_MODE_AUTOREGRESSIVE:
      if S != 1:
        raise ValueError("autoregressive paged attention takes one token per slot")
      st = page_state.to_numpy() if hasattr(page_state, "to_numpy") else page_state
      seq = np.asarray(st["sequence_lengths"] if isinstance(st, dict) else st.sequence_lengths)
      act = np.asarray(st["has_active_page"] if isinstance(st, dict) else st.has_active_page)
      row_pos = torch.from_numpy(np.maximum(seq[:B] - 1, 0).astype(np.int32)).to(dev)     # the position update_decode_pages has advanced to (:360-372)
      row_slot = torch.arange(B, dtype=torch.int32, device=dev)
      row_active = torch.from_numpy(act[:B].astype(np.int32)).to(dev)
    else:
      raise NotImplementedError(f"PagedAttentionOp: model_mode {model_mode}")   # paged_attention.py:401
    R = B * S
    q = query.reshape(R, NQ, D).contiguous()
    k = key.reshape(R, self.num_kv_heads, D).contiguous()
    v = value.reshape(R, self.num_kv_heads, D).contiguous()
    row_slot, row_pos, row_active = row_slot.contiguous(), row_pos.contiguous(), row_active.contiguous()
    self.ctx.kv_write(self.layer, k, v, row_slot, row_pos, row_active)
    out = self.ctx.paged_attention(self.layer, q, row_slot, row_pos, row_active)
    return out.reshape(B, S, NQ, D), None, None
