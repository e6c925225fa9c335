"""MaxText/layers/moe.py: RoutedMoE (:205-1487) for the Mixtral block -- gate (:170-204), top_k + softmax (:326-345), the selected experts'
MLPs (:855-874) and the weighted sum (:455-484) -- over the op-level C entry points, with the experts of a layer split over the ranks of
the default process group when there is one (expert parallelism, :785-938; the exchange is moe_ep.expert_parallel_apply)."""
from __future__ import annotations

from typing import Any, Optional, Tuple

import torch

from .. import moe_ep, runtime


class RoutedMoE:
  """The router and experts are those of `layer` in the parameter tree loaded into the context (`MoeBlock_0/...`): an engine's (engine=...),
  or the module surface's shared context, into which `load_params` puts this block's kernels.  __call__ keeps the reference's signature:
  inputs [batch, length, emb] (bf16) -> (output, load-balance loss = None outside training)."""

  def __init__(self, config: Any, num_experts: int, num_experts_per_tok: int, *, engine: Any = None, layer: int = 0, name: Optional[str] = "MoeBlock_0",
               always_exchange: bool = False, **unused):
    del unused
    if config.decoder_block != "mixtral":
      raise NotImplementedError("RoutedMoE is built for decoder_block mixtral")
    if (num_experts, num_experts_per_tok) != (config.num_experts, config.num_experts_per_tok):
      raise ValueError("RoutedMoE: num_experts / num_experts_per_tok differ from the context's configuration")
    self.ctx = runtime.OpContext(engine=engine) if engine is not None else runtime.context(config)
    self.config, self.layer, self.name = config, int(layer), name
    self.num_experts, self.num_experts_per_tok = int(num_experts), int(num_experts_per_tok)
    self.always_exchange = always_exchange

  def load_params(self, params: Any, prefix: str) -> None:
    """`prefix` = "params/decoder/layers_<i>/MoeBlock_0/": gate/kernel [emb, experts], wi_0 / wi_1 [experts, emb, mlp], wo [experts, mlp, emb]."""
    for leaf in ("gate/kernel", "wi_0", "wi_1", "wo"):
      self.ctx.load_weight(prefix + leaf, params[prefix + leaf])

  def __call__(self, inputs: torch.Tensor) -> Tuple[torch.Tensor, None]:
    shape = inputs.shape
    x = inputs.reshape(-1, shape[-1]).contiguous()
    experts, weights = self.ctx.moe_route(self.layer, x)
    # dropless fixed-capacity exchange around ONE routed launch sequence for this rank's experts: no host synchronisation, no Python loop
    y = moe_ep.expert_parallel_apply_grouped(x, experts.to(torch.int64), self.num_experts, lambda rows, ids: self.ctx.moe_experts(self.layer, rows.contiguous(), ids),
                                             always_exchange=self.always_exchange)
    return self.ctx.moe_combine(y.contiguous(), weights).reshape(shape), None


def get_routed_moe(config: Any, num_experts: int, num_experts_per_tok: int, *, engine: Any = None, layer: int = 0, name: Optional[str] = None, **kw) -> RoutedMoE:
  """moe.py:1602-1633."""
  return RoutedMoE(config, num_experts, num_experts_per_tok, engine=engine, layer=layer, name=name, **kw)
