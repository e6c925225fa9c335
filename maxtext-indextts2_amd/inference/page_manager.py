"""PageManager / PageState with the reference's names (/root/reference/MaxText/inference/page_manager.py),
backed by the device-resident state inside libixt.so.

The reference's manager is functional (state in, state out, each call a jit).  Here the state lives in HBM so
that the per-token `update_decode_pages` can sit inside the captured decode graph; the `page_state` argument of
each method is accepted for signature compatibility and the returned object is a live view of the device state.
Validation and error text follow page_manager.py:450-473, :527-532, :592-594."""
from __future__ import annotations

import ctypes
from typing import Any, Optional

import numpy as np

from .. import _lib


class PageState:
  """Live view of the seven PageState fields (page_manager.py:48-91); each access copies device -> host."""

  FIELDS = ("page_status", "page_map", "num_pages_used", "sequence_lengths", "active_page", "has_active_page", "active_page_position")

  def __init__(self, manager: "PageManager"):
    self._m = manager

  def to_numpy(self) -> dict:
    return self._m._export()

  def __getattr__(self, name):
    if name in PageState.FIELDS:
      return self._m._export()[name]
    raise AttributeError(name)

  def replace(self, **kwargs) -> "PageState":
    """Only page_status can be replaced (test hook: page_manager_test.py:148-150)."""
    for k, v in kwargs.items():
      if k != "page_status":
        raise NotImplementedError(f"PageState.replace({k}=...) is not supported on the device-resident state")
      self._m._import_status(np.asarray(v, np.int32))
    return self


class PageManager:
  """page_manager.py:415-621."""

  def __init__(self, config: Any, engine: Any = None):
    self.num_pages = int(config.pagedattn_num_pages)
    self.tokens_per_page = int(config.pagedattn_tokens_per_page)
    self.max_target_length = int(config.max_target_length)
    self.max_page_groups = int(config.per_device_batch_size) if engine is not None else int(config.global_batch_size_to_load)
    self.max_pages_per_group = int(config.pagedattn_max_pages_per_group)
    self._validate_init_params()
    if engine is None:
      raise _lib.IxtError("PageManager needs the engine that owns the device page state (no CPU implementation in the product)")
    self._engine = engine

  def _validate_init_params(self) -> None:
    if self.max_pages_per_group <= 0:
      raise ValueError("`pagedattn_max_pages_per_group` must be positive.")
    min_required = (self.max_target_length + self.tokens_per_page - 1) // self.tokens_per_page
    if self.max_pages_per_group < min_required:
      raise ValueError(f"`pagedattn_max_pages_per_group` ({self.max_pages_per_group}) is insufficient for `max_target_length` "
                       f"({self.max_target_length}). Needs {min_required}.")
    if self.num_pages <= 1:
      raise ValueError("`pagedattn_num_pages` must be greater than 1.")
    if self.tokens_per_page <= 0:
      raise ValueError("`pagedattn_tokens_per_page` must be positive.")
    if self.max_page_groups <= 0:
      raise ValueError("`pagedattn_max_page_groups` must be positive.")

  # -- plumbing
  def _ctx(self):
    return self._engine._ctx

  def _stream(self):
    import torch  # pylint: disable=import-outside-toplevel
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

  def _export(self) -> dict:
    G, M, N = self.max_page_groups, self.max_pages_per_group, self.num_pages
    out = {
        "page_status": np.zeros((N,), np.int32), "page_map": np.zeros((G, M), np.int32), "num_pages_used": np.zeros((G,), np.int32),
        "sequence_lengths": np.zeros((G,), np.int32), "active_page": np.zeros((G,), np.int32),
        "has_active_page": np.zeros((G,), np.uint8), "active_page_position": np.zeros((G,), np.int32),
    }
    p = lambda a, t: a.ctypes.data_as(t)
    _lib.check(self._engine.lib.ixt_pages_export(
        self._ctx(), p(out["page_status"], _lib.c_i32p), p(out["page_map"], _lib.c_i32p), p(out["num_pages_used"], _lib.c_i32p),
        p(out["sequence_lengths"], _lib.c_i32p), p(out["active_page"], _lib.c_i32p), p(out["has_active_page"], _lib.c_u8p),
        p(out["active_page_position"], _lib.c_i32p), self._stream()))
    out["has_active_page"] = out["has_active_page"].astype(np.bool_)
    return out

  def _import_status(self, status: np.ndarray):
    if status.shape != (self.num_pages,):
      raise ValueError("page_status shape mismatch")
    status = np.ascontiguousarray(status, np.int32)
    _lib.check(self._engine.lib.ixt_pages_import_status(self._ctx(), status.ctypes.data_as(_lib.c_i32p), self._stream()))

  def device_state(self) -> PageState:
    return PageState(self)

  # -- reference surface
  def update_prefill_pages(self, page_state: Optional[PageState], page_group_id: int, true_length: int, _staging: bool = False) -> PageState:
    """page_manager.py:493-536.  `_staging` (engine-internal): the group is one of the staging groups behind the decode slots."""
    if page_group_id < 0 or (page_group_id >= self.max_page_groups and not _staging):
      raise ValueError(f"PageManager: page_group_id ({page_group_id}) out of range [0, {self.max_page_groups})")
    if true_length <= 0 or true_length > self.max_target_length:
      raise ValueError(f"PageManager: true_length ({true_length}) out of range (0, {self.max_target_length}]")
    _lib.check(self._engine.lib.ixt_pages_reserve(self._ctx(), page_group_id, true_length, self._stream()))
    return PageState(self)

  def extend_prefill_pages(self, page_state: Optional[PageState], page_group_id: int, total_length: int) -> PageState:
    """Build extension for chunked prefill over pages."""
    if page_group_id < 0 or page_group_id >= self.max_page_groups:
      raise ValueError(f"PageManager: page_group_id ({page_group_id}) out of range [0, {self.max_page_groups})")
    if total_length <= 0 or total_length > self.max_target_length:
      raise ValueError(f"PageManager: true_length ({total_length}) out of range (0, {self.max_target_length}]")
    _lib.check(self._engine.lib.ixt_pages_extend(self._ctx(), page_group_id, total_length, self._stream()))
    return PageState(self)

  def update_decode_pages(self, page_state: Optional[PageState]) -> PageState:
    _lib.check(self._engine.lib.ixt_pages_advance(self._ctx(), self._stream()))
    return PageState(self)

  def release_pages(self, page_state: Optional[PageState], page_group_id: int) -> PageState:
    if page_group_id < 0 or page_group_id >= self.max_page_groups:
      raise ValueError(f"PageManager: page_group_id ({page_group_id}) out of range [0, {self.max_page_groups})")
    _lib.check(self._engine.lib.ixt_pages_release(self._ctx(), page_group_id, self._stream()))
    return PageState(self)

  def get_initial_page_state(self) -> PageState:
    _lib.check(self._engine.lib.ixt_pages_reset(self._ctx(), self._stream()))
    return PageState(self)
