"""MaxText/layers/mixtral.py: MixtralDecoderLayer (:52-170) = the llama2 wiring with `post_self_attention_layer_norm` and a routed mixture of
experts (`MoeBlock_0`, layers/moe.py) in the MLP's place."""
from __future__ import annotations

from typing import Any, Optional

import torch

from .. import runtime
from .attentions import attention_as_linen
from .moe import get_routed_moe
from .normalizations import rms_norm


class MixtralDecoderLayer:
  """mixtral.py:52-170: h = x + Attention(RMSNorm(x)); out = h + RoutedMoE(RMSNorm(h)).  The experts are split over the ranks of the default
  process group when there is one (moe_ep.expert_parallel_apply)."""

  def __init__(self, config: Any, mesh: Any = None, quant: Any = None, model_mode: Optional[str] = None, name: Optional[str] = None, layer_idx: int = 0,
               always_exchange: bool = False):
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
    self.post_self_attention_layer_norm = rms_norm(num_features=cfg.emb_dim, epsilon=cfg.normalization_layer_epsilon, name="post_self_attention_layer_norm", config=cfg)
    self.MoeBlock_0 = get_routed_moe(cfg, cfg.num_experts, cfg.num_experts_per_tok, engine=None, layer=layer_idx, name="MoeBlock_0", always_exchange=always_exchange)

  def __call__(self, inputs: torch.Tensor, decoder_segment_ids: Any, decoder_positions: torch.Tensor, deterministic: bool, model_mode: str,
               slot: Optional[int] = None, page_state: Any = None, previous_chunk: Any = None):
    lnx = self.pre_self_attention_layer_norm(inputs)
    attention_lnx = self.self_attention(lnx, lnx, decoder_positions, decoder_segment_ids=decoder_segment_ids, deterministic=deterministic,
                                        model_mode=model_mode, slot=slot, page_state=page_state, previous_chunk=previous_chunk)
    intermediate_inputs = self.ctx.residual(inputs.contiguous(), attention_lnx.contiguous())
    hidden_states = self.post_self_attention_layer_norm(intermediate_inputs)
    mlp_lnx, _ = self.MoeBlock_0(hidden_states)
    return self.ctx.residual(mlp_lnx.contiguous(), intermediate_inputs)
