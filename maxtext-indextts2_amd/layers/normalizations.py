"""MaxText/layers/normalizations.py: RMSNorm (:31-70), rms_norm factory (:73-95)."""
from __future__ import annotations

from typing import Any, Optional

import torch

from .. import runtime


class RMSNorm:
  """normalizations.py:31-70: y = bf16(x * rsqrt(mean(x^2) + eps)) (statistics in f32), out = bf16(y * scale)."""

  def __init__(self, num_features: int, epsilon: float = 1e-6, dtype: Any = None, weight_dtype: Any = None, kernel_axes=(), scale_init=None,
               parameter_memory_host_offload: bool = False, *, rngs: Any = None, config: Any = None, name: Optional[str] = None):
    del dtype, weight_dtype, kernel_axes, scale_init, parameter_memory_host_offload, rngs
    self.ctx = runtime.context(config)
    cfg_eps = float(self.ctx.config.normalization_layer_epsilon)
    if abs(float(epsilon) - cfg_eps) > 1e-12:
      raise ValueError(f"RMSNorm epsilon {epsilon} differs from the context's normalization_layer_epsilon {cfg_eps}")
    self.num_features, self.epsilon, self.name = num_features, epsilon, name
    self.scale = torch.ones((num_features,), dtype=torch.bfloat16, device="cuda")   # nn.initializers.ones (:40)

  def __call__(self, x: torch.Tensor) -> torch.Tensor:
    shp = x.shape
    return self.ctx.rmsnorm(x.reshape(-1, shp[-1]).contiguous(), self.scale).reshape(shp)


def rms_norm(num_features: int, epsilon: float = 1e-6, dtype: Any = None, weight_dtype: Any = None, kernel_axes=(), scale_init=None,
             name: Optional[str] = None, parameter_memory_host_offload: bool = False, config: Any = None) -> RMSNorm:
  """normalizations.py:73-95."""
  return RMSNorm(num_features, epsilon, dtype, weight_dtype, kernel_axes, scale_init, parameter_memory_host_offload, config=config, name=name)
