"""MaxText/layers/models.py: Transformer (:41-170) = shared Embed (:68) + Decoder."""
from __future__ import annotations

from typing import Any, Dict, Optional

import torch

from .. import runtime
from ..common_types import MODEL_MODE_PREFILL, MODEL_MODE_TRAIN
from .decoders import Decoder
from .embeddings import embed_as_linen


class Transformer:
  """models.py:41-170.  `load_params` takes the unscanned MaxText tree ({"params/decoder/layers_3/mlp/wi_0/kernel": array, ...})."""

  def __init__(self, config: Any, mesh: Any = None, quant: Any = None, model_mode: str = MODEL_MODE_TRAIN):
    self.config, self.mesh, self.quant, self.model_mode = config, mesh, quant, model_mode
    self.ctx = runtime.context(config)
    self.token_embedder = embed_as_linen(num_embeddings=config.vocab_size, num_features=config.emb_dim, config=config, name="token_embedder")
    self.shared_embedding = self.token_embedder
    self.decoder = Decoder(config, shared_embedding=self.shared_embedding, mesh=mesh, quant=quant, model_mode=model_mode)

  def load_params(self, params: Dict[str, Any]) -> None:
    dev = torch.device("cuda", torch.cuda.current_device())
    cv = lambda a, shape: runtime.as_bf16(a, dev).reshape(shape)
    c = self.config
    self.token_embedder.embedding = cv(params["params/token_embedder/embedding"], (c.vocab_size, c.emb_dim))
    self.decoder.decoder_norm.scale = cv(params["params/decoder/decoder_norm/scale"], (c.emb_dim,))
    self.decoder.logits_dense.kernel = cv(params["params/decoder/logits_dense/kernel"], (c.emb_dim, c.vocab_size))
    for i, layer in enumerate(self.decoder.layers):
      p = f"params/decoder/layers_{i}/"
      a = layer.self_attention
      layer.pre_self_attention_layer_norm.scale = cv(params[p + "pre_self_attention_layer_norm/scale"], (c.emb_dim,))
      a.query.kernel = cv(params[p + "self_attention/query/kernel"], (c.emb_dim, c.num_query_heads, c.head_dim))
      a.key.kernel = cv(params[p + "self_attention/key/kernel"], (c.emb_dim, c.num_kv_heads, c.head_dim))
      a.value.kernel = cv(params[p + "self_attention/value/kernel"], (c.emb_dim, c.num_kv_heads, c.head_dim))
      a.out.kernel = cv(params[p + "self_attention/out/kernel"], (c.num_query_heads, c.head_dim, c.emb_dim))
      if c.decoder_block == "mixtral":
        layer.post_self_attention_layer_norm.scale = cv(params[p + "post_self_attention_layer_norm/scale"], (c.emb_dim,))
        layer.MoeBlock_0.load_params(params, p + "MoeBlock_0/")
        continue
      norm_key = p + "mlp/mlp_layer_norm/scale"
      if norm_key not in params:
        norm_key = p + "post_self_attention_layer_norm/scale"      # the converters' spelling (llama_or_mistral_ckpt.py:449-450)
      layer.mlp.mlp_layer_norm.scale = cv(params[norm_key], (c.emb_dim,))
      layer.mlp.wi_0.kernel = cv(params[p + "mlp/wi_0/kernel"], (c.emb_dim, c.mlp_dim))
      layer.mlp.wi_1.kernel = cv(params[p + "mlp/wi_1/kernel"], (c.emb_dim, c.mlp_dim))
      layer.mlp.wo.kernel = cv(params[p + "mlp/wo/kernel"], (c.mlp_dim, c.emb_dim))

  def __call__(self, decoder_input_tokens: torch.Tensor, decoder_positions: torch.Tensor, decoder_segment_ids: Any = None, encoder_images: Any = None,
               enable_dropout: bool = True, model_mode: str = MODEL_MODE_TRAIN, previous_chunk: Any = None, true_length: Optional[int] = None,
               slot: Optional[int] = None, page_state: Any = None, **kwargs):
    """models.py:87-170 -> f32 logits [B, S, V]."""
    del enable_dropout, true_length, kwargs
    if encoder_images is not None:
      raise NotImplementedError("multimodal inputs are outside the built path")
    if decoder_segment_ids is not None and model_mode == "autoregressive":
      raise ValueError("During autoregressive decoding we assume the tokens are in the active sequence which is always first.")   # models.py:112-116
    return self.decoder(decoder_input_tokens, decoder_positions, decoder_segment_ids=decoder_segment_ids, deterministic=True, model_mode=model_mode,
                        previous_chunk=previous_chunk, slot=slot, page_state=page_state)
