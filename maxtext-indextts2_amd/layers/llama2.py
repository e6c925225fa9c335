"""MaxText/layers/llama2.py: LlamaDecoderLayer (:54-165)."""
from __future__ import annotations

from typing import Any, Optional

import torch

from .. import runtime
from .attentions import attention_as_linen
from .linears import mlp_block
from .normalizations import rms_norm


class LlamaDecoderLayer:
  """llama2.py:54-165: x + Attention(RMSNorm(x)), then h + MlpBlock(h) with the MLP's own pre-norm (linears.py:430-431)."""

  def __init__(self, config: Any, mesh: Any = None, quant: Any = None, model_mode: Optional[str] = None, name: Optional[str] = None, layer_idx: int = 0):
    del mesh, model_mode
    if quant is not None:
      raise NotImplementedError("quantised layers are outside the built path")
    cfg = config
    self.config, self.name = cfg, name
    self.ctx = runtime.context(cfg)
    self.pre_self_attention_layer_norm = rms_norm(num_features=cfg.emb_dim, epsilon=cfg.normalization_layer_epsilon, name="pre_self_attention_layer_norm", config=cfg)
    self.self_attention = attention_as_linen(config=cfg, num_query_heads=cfg.num_query_heads, num_kv_heads=cfg.num_kv_heads, head_dim=cfg.head_dim,
                                             max_target_length=cfg.max_target_length, attention_kernel=cfg.attention, inputs_q_shape=(1, 1, cfg.emb_dim),
                                             inputs_kv_shape=(1, 1, cfg.emb_dim), name="self_attention", layer_idx=layer_idx)
    self.mlp = mlp_block(in_features=cfg.emb_dim, intermediate_dim=cfg.mlp_dim, activations=cfg.mlp_activations, config=cfg, name="mlp", use_pre_norm=True)

  def __call__(self, inputs: torch.Tensor, decoder_segment_ids: Any, decoder_positions: torch.Tensor, deterministic: bool, model_mode: str,
               slot: Optional[int] = None, page_state: Any = None, previous_chunk: Any = None):
    lnx = self.pre_self_attention_layer_norm(inputs)
    attention_lnx = self.self_attention(lnx, lnx, decoder_positions, decoder_segment_ids=decoder_segment_ids, deterministic=deterministic,
                                        model_mode=model_mode, slot=slot, page_state=page_state, previous_chunk=previous_chunk)
    intermediate_inputs = self.ctx.residual(inputs.contiguous(), attention_lnx.contiguous())
    mlp_lnx = self.mlp(intermediate_inputs, deterministic=deterministic)
    return self.ctx.residual(mlp_lnx.contiguous(), intermediate_inputs)
