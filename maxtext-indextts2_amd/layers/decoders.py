"""MaxText/layers/decoders.py: Decoder (:242-800): embedding (:493-534), the unscanned layer loop (:762-789; paged attention needs it,
SURVEY.md section 3.1), final norm + logits (:537-589)."""
from __future__ import annotations

from typing import Any, Optional

import torch

from .. import runtime
from ..common_types import MODEL_MODE_PREFILL
from .linears import dense_general
from .llama2 import LlamaDecoderLayer
from .mixtral import MixtralDecoderLayer
from .normalizations import rms_norm


class Decoder:
  def __init__(self, config: Any, shared_embedding: Any, mesh: Any = None, quant: Any = None, model_mode: str = MODEL_MODE_PREFILL):
    del mesh, model_mode
    cfg = config
    if cfg.decoder_block not in ("llama2", "default", "mixtral") or cfg.logits_via_embedding:
      raise NotImplementedError("Decoder: llama2 and mixtral blocks with untied logits are built as modules (gemma2 / gemma3: engine only)")
    self.config, self.shared_embedding = cfg, shared_embedding
    self.ctx = runtime.context(cfg)
    block = MixtralDecoderLayer if cfg.decoder_block == "mixtral" else LlamaDecoderLayer      # decoders.py:330-360
    self.layers = [block(cfg, quant=quant, name=f"layers_{i}", layer_idx=i) for i in range(cfg.num_decoder_layers)]
    self.decoder_norm = rms_norm(num_features=cfg.emb_dim, epsilon=cfg.normalization_layer_epsilon, name="decoder_norm", config=cfg)
    self.logits_dense = dense_general(inputs_shape=(1, 1, cfg.emb_dim), out_features_shape=cfg.vocab_size, name="logits_dense", config=cfg)

  def _apply_embedding(self, decoder_input_tokens, model_mode):
    return self.shared_embedding(decoder_input_tokens.to(torch.int32), model_mode=model_mode)     # decoders.py:493-534 (bf16 rows)

  def _apply_output_head(self, y):
    y = self.decoder_norm(y)
    return self.logits_dense(y).float()     # bf16(t) then cast_logits_to_fp32 (decoders.py:570, :586-587)

  def __call__(self, decoder_input_tokens: torch.Tensor, decoder_positions: torch.Tensor, decoder_segment_ids: Any = None, deterministic: bool = False,
               model_mode: str = MODEL_MODE_PREFILL, previous_chunk: Any = None, slot: Optional[int] = None, page_state: Any = None, **kwargs):
    del kwargs
    y = self._apply_embedding(decoder_input_tokens, model_mode)
    for layer in self.layers:
      y = layer(y, decoder_segment_ids, decoder_positions, deterministic, model_mode, slot=slot, page_state=page_state, previous_chunk=previous_chunk)
    return self._apply_output_head(y)
