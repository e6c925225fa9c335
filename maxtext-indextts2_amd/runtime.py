"""Op-level runtime of the `layers.*` / `inference.*` module surface: one ixt_ctx (scratch, RoPE tables, page pools) per configuration,
shared by every module built from it.  Each module call is one or a few `ixt_op_*` calls of the C ABI (include/ixt.h) on the caller's
torch CUDA tensors; PyTorch carries the memory and the stream, nothing else -- there is no CPU path."""
from __future__ import annotations

import ctypes
from typing import Any

import torch

from . import _lib


def _vp(t) -> ctypes.c_void_p:
  return ctypes.c_void_p(t.data_ptr())


def _stream() -> ctypes.c_void_p:
  return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class OpContext:
  """Owns (or borrows from a MaxEngine) the ixt_ctx the op-level entry points run in."""

  def __init__(self, config: Any = None, engine: Any = None):
    if engine is not None:
      self.lib, self.ctx, self.config, self._owner = engine.lib, engine._ctx, engine.config, engine
      self.page_manager = engine.page_manager
      return
    if not torch.cuda.is_available():
      raise _lib.IxtError("the layers.* modules need a ROCm GPU: there is no CPU path")
    from .maxengine import MaxEngine  # pylint: disable=import-outside-toplevel
    self._owner = MaxEngine(config)      # an engine without weights: ctx, pools, tables
    self.lib, self.ctx, self.config = self._owner.lib, self._owner._ctx, config
    self.page_manager = self._owner.page_manager

  # ---- thin wrappers; tensors are contiguous CUDA tensors, bf16 unless stated
  def dot(self, x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """x bf16 [R, K] . w bf16 [K, N] -> f32 [R, N] (canonical order, unrounded)."""
    R, K = x.shape
    N = w.shape[1]
    out = torch.empty((R, N), dtype=torch.float32, device=x.device)
    _lib.check(self.lib.ixt_op_dot(self.ctx, _vp(x), _vp(w), _vp(out), R, K, N, _stream()))
    return out

  def rmsnorm(self, x: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    R, E = x.shape
    out = torch.empty_like(x)
    _lib.check(self.lib.ixt_op_rmsnorm(self.ctx, _vp(x), _vp(scale), _vp(out), R, E, _stream()))
    return out

  def rope(self, x: torch.Tensor, pos: torch.Tensor, nheads: int) -> torch.Tensor:
    """in place on x bf16 [R, nheads, head_dim]; pos int32 [R]."""
    _lib.check(self.lib.ixt_op_rope(self.ctx, _vp(x), _vp(pos), x.shape[0], nheads, _stream()))
    return x

  def load_weight(self, name: str, array: Any) -> None:
    """One leaf of the MaxText parameter tree into the context (ixt_load_weight): the kernels a module cannot carry as a plain tensor because
    the library keeps them in its own layout (the experts of a Mixtral block)."""
    dev = torch.device("cuda", torch.cuda.current_device())
    t = as_bf16(array, dev)
    shape = (ctypes.c_int64 * t.dim())(*t.shape)
    _lib.check(self.lib.ixt_load_weight(self.ctx, name.encode(), _vp(t), _lib.IXT_BF16, shape, t.dim(), _stream()))
    torch.cuda.current_stream().synchronize()

  def moe_route(self, layer: int, x: torch.Tensor):
    """x bf16 [R, emb] -> (experts int32 [R, k] in top_k order, weights f32 [R, k])."""
    R, K = x.shape[0], int(self.config.num_experts_per_tok)
    experts = torch.empty((R, K), dtype=torch.int32, device=x.device)
    weights = torch.empty((R, K), dtype=torch.float32, device=x.device)
    _lib.check(self.lib.ixt_op_moe_route(self.ctx, layer, _vp(x), R, _vp(experts), _vp(weights), _stream()))
    return experts, weights

  def moe_expert(self, layer: int, expert: int, x: torch.Tensor) -> torch.Tensor:
    out = torch.empty_like(x)
    _lib.check(self.lib.ixt_op_moe_expert(self.ctx, layer, int(expert), _vp(x), x.shape[0], _vp(out), _stream()))
    return out

  def moe_experts(self, layer: int, x: torch.Tensor, expert_ids: torch.Tensor) -> torch.Tensor:
    """Routed evaluation of this context's experts: x bf16 [n, emb], expert_ids int32 [n] (global ids; negative or another rank's: padding ->
    a zero row).  One launch sequence for all experts, no host synchronisation."""
    out = torch.empty_like(x)
    ids = expert_ids.to(torch.int32).contiguous()
    _lib.check(self.lib.ixt_op_moe_experts(self.ctx, layer, _vp(x), _vp(ids), x.shape[0], _vp(out), _stream()))
    return out

  def moe_combine(self, y: torch.Tensor, weights: torch.Tensor) -> torch.Tensor:
    out = torch.empty((y.shape[0], y.shape[2]), dtype=y.dtype, device=y.device)
    _lib.check(self.lib.ixt_op_moe_combine(self.ctx, _vp(y), _vp(weights), y.shape[0], _vp(out), _stream()))
    return out

  def swiglu(self, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    out = torch.empty_like(a)
    _lib.check(self.lib.ixt_op_swiglu(self.ctx, _vp(a), _vp(b), _vp(out), a.numel(), _stream()))
    return out

  def residual(self, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    out = torch.empty_like(a)
    _lib.check(self.lib.ixt_op_residual(self.ctx, _vp(a), _vp(b), _vp(out), a.numel(), _stream()))
    return out

  def embed(self, table: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    R = ids.numel()
    V, E = table.shape
    out = torch.empty((R, E), dtype=torch.bfloat16, device=table.device)
    _lib.check(self.lib.ixt_op_embed(self.ctx, _vp(table), _vp(ids), _vp(out), R, E, V, _stream()))
    return out

  def kv_write(self, layer: int, k, v, row_slot, row_pos, row_active):
    _lib.check(self.lib.ixt_op_kv_write(self.ctx, layer, _vp(k), _vp(v), _vp(row_slot), _vp(row_pos), _vp(row_active), k.shape[0], _stream()))

  def paged_attention(self, layer: int, q, row_slot, row_pos, row_active) -> torch.Tensor:
    out = torch.empty_like(q)
    _lib.check(self.lib.ixt_op_paged_attention(self.ctx, layer, _vp(q), _vp(row_slot), _vp(row_pos), _vp(row_active), q.shape[0], _vp(out), _stream()))
    return out


_default: OpContext | None = None


def set_context(ctx: OpContext) -> OpContext:
  global _default
  _default = ctx
  return ctx


def context(config: Any = None) -> OpContext:
  """The module surface's shared context; created from `config` on first use."""
  global _default
  if _default is None:
    if config is None:
      raise _lib.IxtError("layers.*: no op context yet -- build the first module with a config, or call runtime.set_context(OpContext(...))")
    _default = OpContext(config)
  return _default


def as_bf16(x, device) -> torch.Tensor:
  """torch bf16 / f32 tensor or numpy (uint16 = bf16 bit patterns, float32) -> contiguous CUDA bf16 tensor."""
  import numpy as np  # pylint: disable=import-outside-toplevel
  if isinstance(x, torch.Tensor):
    return x.to(device=device, dtype=torch.bfloat16).contiguous()
  a = np.asarray(x)
  if a.dtype == np.uint16:
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).to(device).view(torch.bfloat16)
  return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device).to(torch.bfloat16)
