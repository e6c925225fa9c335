"""MaxText/layers/embeddings.py: Embed (:89-181), attend_on_embedding (:183-199), RotaryEmbedding (:235-315), LLaMARotaryEmbedding (:354-488)."""
from __future__ import annotations

from typing import Any, Optional

import torch

from .. import runtime
from ..common_types import MODEL_MODE_TRAIN


class Embed:
  """embeddings.py:89-181: integers [0, n) -> rows of the [num_embeddings, num_features] table, in the activation dtype (bf16)."""

  def __init__(self, num_embeddings: int, num_features: int, config: Any, cast_input_dtype: Any = None, dtype: Any = None, attend_dtype: Any = None,
               embedding_init: Any = None, *, rngs: Any = None, name: Optional[str] = None):
    del cast_input_dtype, dtype, attend_dtype, embedding_init, rngs
    self.ctx = runtime.context(config)
    self.num_embeddings, self.num_features, self.config, self.name = num_embeddings, num_features, config, name
    self.embedding = torch.zeros((num_embeddings, num_features), dtype=torch.bfloat16, device="cuda")

  def __call__(self, inputs: torch.Tensor, model_mode: str = MODEL_MODE_TRAIN) -> torch.Tensor:
    del model_mode
    if inputs.dtype not in (torch.int32, torch.int64):
      raise ValueError("Input type must be an integer or unsigned integer.")
    ids = inputs.to(torch.int32).reshape(-1).contiguous()
    return self.ctx.embed(self.embedding, ids).reshape(tuple(inputs.shape) + (self.num_features,))

  def attend(self, query: torch.Tensor) -> torch.Tensor:
    return attend_on_embedding(query, self.embedding, None, self.config)


def embed_as_linen(*, num_embeddings: int, num_features: int, config: Any, name: Optional[str] = None, **kwargs) -> Embed:
  """embeddings.py:45-86."""
  del kwargs
  return Embed(num_embeddings, num_features, config, name=name)


def attend_on_embedding(query: torch.Tensor, embedding_table: torch.Tensor, attend_dtype: Any, config: Any) -> torch.Tensor:
  """embeddings.py:183-199: query . table^T with f32 results (the tied-logits path, decoders.py:552-562, applies its 1/sqrt(emb)
  to the query before this call).  The table is transposed once per call (a copy, no arithmetic)."""
  del attend_dtype
  ctx = runtime.context(config)
  lead, E = query.shape[:-1], query.shape[-1]
  out = ctx.dot(query.reshape(-1, E).to(torch.bfloat16).contiguous(), embedding_table.t().contiguous())
  return out.reshape(tuple(lead) + (embedding_table.shape[0],))


class RotaryEmbedding:
  """embeddings.py:235-315: half-split rotation with sin / cos rounded to the input dtype before the multiply (:306-307).  The tables
  live in the context (built from rope_min_timescale / rope_max_timescale / head_dim of its config)."""

  def __init__(self, min_timescale: int, max_timescale: int, embedding_dims: int = 0, cast_as_fprop_dtype: bool = True, fprop_dtype: Any = None,
               rngs: Any = None, config: Any = None, name: Optional[str] = None):
    del cast_as_fprop_dtype, fprop_dtype, rngs
    if embedding_dims % 2:
      raise ValueError("Embedding dim for rotary position embedding must be a multiple of 2.")
    self.ctx = runtime.context(config)
    c = self.ctx.config
    if (float(min_timescale), float(max_timescale), int(embedding_dims)) != (float(c.rope_min_timescale), float(c.rope_max_timescale), int(c.head_dim)):
      raise ValueError("RotaryEmbedding: timescales / dims differ from the context's rope tables")
    self.min_timescale, self.max_timescale, self.embedding_dims, self.name = min_timescale, max_timescale, embedding_dims, name

  def __call__(self, inputs: torch.Tensor, position: Optional[torch.Tensor] = None) -> torch.Tensor:
    assert position is not None
    if inputs.dim() != 4:
      raise ValueError("Input is assumed to be a rank 4 tensor of shape[batch, sequence, heads, dims].")
    if self.embedding_dims != inputs.shape[3]:
      raise ValueError("The embedding dims of the rotary position embeddingmust match the hidden dimension of the inputs.")
    B, S, N, H = inputs.shape
    x = inputs.reshape(B * S, N, H).contiguous().clone()
    self.ctx.rope(x, position.to(torch.int32).reshape(-1).contiguous(), N)
    return x.reshape(B, S, N, H)


def rotary_embedding_as_linen(*, min_timescale: int, max_timescale: int, embedding_dims: int = 0, name: Optional[str] = None, config: Any = None, **kwargs):
  """embeddings.py:202-232."""
  del kwargs
  return RotaryEmbedding(min_timescale, max_timescale, embedding_dims, config=config, name=name)


class LLaMARotaryEmbedding(RotaryEmbedding):
  """embeddings.py:354-488: the rotation pairs are the neighbouring features (2 i, 2 i + 1), sin / cos are f32, the result is cast to bf16 once;
  `use_scale` = the llama-3.1 frequency scaling (:417-443).  The tables live in the context, which must have been built with
  rope_type "llama3.1" and the same use_scale (ixt_config.rope_llama31 / rope_use_scale)."""

  def __init__(self, min_timescale: int, max_timescale: int, embedding_dims: int = 0, cast_as_fprop_dtype: bool = True, fprop_dtype: Any = None,
               use_scale: bool = True, rngs: Any = None, config: Any = None, name: Optional[str] = None):
    super().__init__(min_timescale, max_timescale, embedding_dims, cast_as_fprop_dtype, fprop_dtype, rngs=rngs, config=config, name=name)
    c = self.ctx.config
    if not str(c.rope_type).lower().startswith("llama3.1") or bool(c.rope_use_scale) != bool(use_scale):
      raise ValueError("LLaMARotaryEmbedding: the context was not built with rope_type llama3.1 and this use_scale")
    self.use_scale = use_scale


def llama_rotary_embedding_as_linen(*, min_timescale: int, max_timescale: int, embedding_dims: int = 0, use_scale: bool = True, name: Optional[str] = None,
                                    config: Any = None, **kwargs):
  """embeddings.py:318-351."""
  del kwargs
  return LLaMARotaryEmbedding(min_timescale, max_timescale, embedding_dims, use_scale=use_scale, config=config, name=name)
