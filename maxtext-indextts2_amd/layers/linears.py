"""MaxText/layers/linears.py: DenseGeneral (:96-232), dense_general (:235-294), MlpBlock (:297-476), mlp_block (:478-513)."""
from __future__ import annotations

import math
from typing import Any, Iterable, Optional, Sequence, Union

import torch

from .. import runtime
from .normalizations import RMSNorm


def _tuple(x) -> tuple:
  return tuple(x) if isinstance(x, (tuple, list)) else (int(x),)


class DenseGeneral:
  """linears.py:96-232: contraction of the trailing `axis` dimensions of the input with the leading dimensions of `kernel`
  ([*in_features_shape, *out_features_shape], bf16 at use :216); f32 accumulation in the canonical order, bf16 result (:93)."""

  def __init__(self, in_features_shape: Union[Iterable[int], int], out_features_shape: Union[Iterable[int], int], axis: Union[Iterable[int], int] = -1,
               weight_dtype: Any = None, dtype: Any = None, kernel_init: Any = None, kernel_axes=(), quant: Any = None, use_bias: bool = False,
               matmul_precision: str = "default", parameter_memory_host_offload: bool = False, *, rngs: Any = None, config: Any = None,
               name: Optional[str] = None):
    del weight_dtype, dtype, kernel_init, kernel_axes, matmul_precision, parameter_memory_host_offload, rngs
    if quant is not None or use_bias:
      raise NotImplementedError("DenseGeneral: quantised kernels and biases are outside the built path")
    self.ctx = runtime.context(config)
    self.in_features_shape, self.out_features_shape, self.axis, self.name = _tuple(in_features_shape), _tuple(out_features_shape), _tuple(axis), name
    n = len(self.axis)
    if tuple(sorted(a % 64 for a in self.axis)) != tuple(sorted((-(i + 1)) % 64 for i in range(n))):
      raise NotImplementedError("DenseGeneral: only contractions over the trailing input axes are built")
    self.kernel = torch.zeros(self.in_features_shape + self.out_features_shape, dtype=torch.bfloat16, device="cuda")

  def __call__(self, inputs: torch.Tensor, _initializing: bool = False) -> torch.Tensor:
    n = len(self.axis)
    lead = inputs.shape[:inputs.dim() - n]
    K, N = math.prod(self.in_features_shape), math.prod(self.out_features_shape)
    if tuple(inputs.shape[inputs.dim() - n:]) != self.in_features_shape:
      raise ValueError(f"DenseGeneral {self.name}: input {tuple(inputs.shape)} does not end in {self.in_features_shape}")
    out = self.ctx.dot(inputs.reshape(-1, K).contiguous(), self.kernel.reshape(K, N))
    return out.to(torch.bfloat16).reshape(tuple(lead) + self.out_features_shape)


def dense_general(*, inputs_shape=None, in_features_shape=None, out_features_shape, axis=-1, name: Optional[str] = None, config: Any = None, **kwargs):
  """linears.py:235-294 (either inputs_shape + axis or in_features_shape names the contracted dimensions)."""
  if in_features_shape is None:
    ax = _tuple(axis)
    in_features_shape = tuple(inputs_shape[a] for a in ax)
  kwargs = {k: v for k, v in kwargs.items() if k in ("quant", "use_bias")}
  return DenseGeneral(in_features_shape, out_features_shape, axis, config=config, name=name, **kwargs)


class MlpBlock:
  """linears.py:297-476 with the llama2 settings (activations ("silu", "linear"), no bias): optional pre-norm `mlp_layer_norm` (:347-354),
  wi_0 / wi_1 / wo; out = wo(bf16(bf16(silu(wi_0 x)) * wi_1 x)) (:449-476)."""

  def __init__(self, config: Any, in_features: int, intermediate_dim: int = 2048, activations: Sequence[Any] = ("relu",), kernel_init: Any = None,
               intermediate_dropout_rate: float = 0.1, dtype: Any = None, weight_dtype: Any = None, use_bias: bool = False, use_pre_norm: bool = False,
               quant: Any = None, model_mode: Optional[str] = None, *, rngs: Any = None, name: Optional[str] = None):
    del kernel_init, intermediate_dropout_rate, dtype, weight_dtype, model_mode, rngs
    if tuple(activations) != ("silu", "linear") or use_bias or quant is not None:
      raise NotImplementedError("MlpBlock: only the gated silu MLP without bias / quantisation is built")
    self.config, self.name = config, name
    self.mlp_layer_norm = RMSNorm(in_features, float(config.normalization_layer_epsilon), config=config, name="mlp_layer_norm") if use_pre_norm else None
    self.wi_0 = DenseGeneral(in_features, intermediate_dim, config=config, name="wi_0")
    self.wi_1 = DenseGeneral(in_features, intermediate_dim, config=config, name="wi_1")
    self.wo = DenseGeneral(intermediate_dim, in_features, config=config, name="wo")
    self.ctx = runtime.context(config)

  def __call__(self, inputs: torch.Tensor, decode: bool = False, deterministic: bool = False) -> torch.Tensor:
    del decode, deterministic
    x = self.mlp_layer_norm(inputs) if self.mlp_layer_norm is not None else inputs
    a, b = self.wi_0(x), self.wi_1(x)
    return self.wo(self.ctx.swiglu(a.contiguous(), b.contiguous()))


def mlp_block(*, config: Any, in_features: int, intermediate_dim: int = 2048, activations=("relu",), intermediate_dropout_rate: float = 0.1, dtype=None,
              weight_dtype=None, use_bias: bool = False, use_pre_norm: bool = False, quant=None, model_mode=None, name: Optional[str] = None, **_):
  """linears.py:478-513."""
  return MlpBlock(config, in_features, intermediate_dim, activations, None, intermediate_dropout_rate, dtype, weight_dtype, use_bias, use_pre_norm, quant,
                  model_mode, name=name)
