"""The training step of the reference (MaxText/train.py:255-472) on MI355X: `Trainer.train_step(batch)` = loss_fn + value_and_grad on this
GPU's rows (HIP, include/ixt_train.h), the data-parallel gradient exchange (RCCL reduce-scatter + all-gather over xGMI through
torch.distributed, SURVEY.md section 5.8 -- the only collective of the whole build), global-norm clip, AdamW, warmup-cosine schedule.

State layout: ONE flat f32 device buffer each for parameters, gradients, AdamW mu and nu (caller = this module owns them as torch tensors,
the library reads and writes them in place); `param_views()` gives the MaxText tree ({"params/decoder/layers_3/mlp/wi_0/kernel": tensor}).
Batches follow maxtext_utils.get_shaped_batch (maxtext_utils.py:85-99): six int32 [batch, max_target_length] arrays
(maxtext-indextts2_amd/audio_training.py builds them).  There is no CPU path."""
from __future__ import annotations

import ctypes
import math
from typing import Any, Callable, Dict, Optional

import numpy as np
import torch

from . import _lib
from . import dp as _dp

EPS = 1e-8   # MaxText/globals.py:20


class IxtTrainConfig(ctypes.Structure):
  _fields_ = [(n, ctypes.c_int32) for n in ("vocab_size", "emb_dim", "num_query_heads", "num_kv_heads", "head_dim", "mlp_dim", "num_decoder_layers")] + [
      ("normalization_layer_epsilon", ctypes.c_float), ("rope_min_timescale", ctypes.c_float), ("rope_max_timescale", ctypes.c_float),
      ("batch", ctypes.c_int32), ("seq_len", ctypes.c_int32), ("logits_chunk", ctypes.c_int32)]


def create_learning_rate_schedule(config) -> Callable[[int], float]:
  """maxtext_utils.py:1245-1285: linear warm-up 0 -> lr over schedule_steps * warmup_steps_fraction, cosine to lr * final_fraction until
  schedule_steps, then 0 (optax.join_schedules of optax.linear_schedule, the cosine piece and optax.constant_schedule)."""
  lr = float(config.learning_rate)
  final = lr * float(config.cosine_learning_rate_final_fraction)
  sched = int(config.learning_rate_schedule_steps) if int(config.learning_rate_schedule_steps) > 0 else int(config.steps)
  warm = int(sched * float(config.warmup_steps_fraction))
  cos_steps = sched - warm

  def schedule(step: int) -> float:
    if step < warm:
      return lr * step / warm
    if step < warm + cos_steps:
      a = 0.5 * (math.cos(math.pi * (step - warm) / cos_steps) + 1.0)
      return lr * a + final * (1.0 - a)
    return 0.0
  return schedule


def _signatures(lib):
  vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
  lib.ixt_train_last_error.restype = ctypes.c_char_p
  lib.ixt_train_create.argtypes, lib.ixt_train_create.restype = [ctypes.POINTER(IxtTrainConfig), ctypes.POINTER(vp)], ci
  lib.ixt_train_destroy.argtypes = [vp]
  lib.ixt_train_total_numel.argtypes, lib.ixt_train_total_numel.restype = [vp], ctypes.c_int64
  lib.ixt_train_param_count.argtypes, lib.ixt_train_param_count.restype = [vp], ci
  lib.ixt_train_param_info.argtypes = [vp, ci, ctypes.c_char_p, ci, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
  lib.ixt_train_bind.argtypes = [vp, vp, vp, vp, vp]
  lib.ixt_train_refresh_bf16.argtypes = [vp, vp]
  lib.ixt_train_forward_backward.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
  lib.ixt_train_apply.argtypes = [vp, cf, cf, cf, cf, cf, cf, cf, cf, ci, vp, vp]
  for f in ("ixt_train_param_info", "ixt_train_bind", "ixt_train_refresh_bf16", "ixt_train_forward_backward", "ixt_train_apply", "ixt_train_destroy"):
    getattr(lib, f).restype = ci


def _check(lib, rc):
  if rc == _lib.IXT_OK:
    return
  msg = lib.ixt_train_last_error().decode("utf-8", "replace")
  if rc in (_lib.IXT_ERR_INVALID, _lib.IXT_ERR_RANGE):
    raise ValueError(msg)
  if rc == _lib.IXT_ERR_UNSUPPORTED:
    raise NotImplementedError(msg)
  raise _lib.IxtError(f"ixt_train status {rc}: {msg}")


def _sp():
  return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class Trainer:
  def __init__(self, config: Any, batch: Optional[int] = None, seq_len: Optional[int] = None, logits_chunk: int = 0, exchange: str = "auto"):
    if not torch.cuda.is_available():
      raise _lib.IxtError("Trainer needs a ROCm GPU: there is no CPU path")
    if str(config.rope_type).lower() != "default":
      raise NotImplementedError(f"train step: rope_type {config.rope_type} is not built (the default RotaryEmbedding is)")
    self.config, self.lib = config, _lib.load()
    _signatures(self.lib)
    self.device = torch.device("cuda", torch.cuda.current_device())
    self.B = int(batch if batch is not None else config.micro_batch_size_to_train_on)
    self.S = int(seq_len if seq_len is not None else config.max_target_length)
    c = IxtTrainConfig()
    for k in ("vocab_size", "emb_dim", "num_query_heads", "num_kv_heads", "head_dim", "mlp_dim", "num_decoder_layers"):
      setattr(c, k, int(getattr(config, k)))
    c.normalization_layer_epsilon = float(config.normalization_layer_epsilon)
    c.rope_min_timescale, c.rope_max_timescale = float(config.rope_min_timescale), float(config.rope_max_timescale)
    c.batch, c.seq_len, c.logits_chunk = self.B, self.S, int(logits_chunk)
    self._h = ctypes.c_void_p()
    _check(self.lib, self.lib.ixt_train_create(ctypes.byref(c), ctypes.byref(self._h)))
    n = int(self.lib.ixt_train_total_numel(self._h))
    z = lambda: torch.zeros((n,), dtype=torch.float32, device=self.device)
    self.params, self.grads, self.mu, self.nu = z(), z(), z(), z()
    _check(self.lib, self.lib.ixt_train_bind(self._h, *[ctypes.c_void_p(t.data_ptr()) for t in (self.params, self.grads, self.mu, self.nu)]))
    self.layout: Dict[str, tuple] = {}
    name = ctypes.create_string_buffer(256)
    off, num = ctypes.c_int64(), ctypes.c_int64()
    for i in range(self.lib.ixt_train_param_count(self._h)):
      _check(self.lib, self.lib.ixt_train_param_info(self._h, i, name, 256, ctypes.byref(off), ctypes.byref(num)))
      self.layout[name.value.decode()] = (int(off.value), int(num.value))
    self.step = 0                                    # optimizer steps taken (optax count)
    self.schedule = create_learning_rate_schedule(config)
    self.exchange = exchange
    self._lw = torch.zeros((2,), dtype=torch.float32, device=self.device)
    self._norms = torch.zeros((2,), dtype=torch.float32, device=self.device)

  def __del__(self):
    try:
      if getattr(self, "_h", None):
        self.lib.ixt_train_destroy(self._h)
        self._h = None
    except Exception:  # pylint: disable=broad-except
      pass

  # ---- state
  def param_views(self, buf: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    buf = self.params if buf is None else buf
    return {k: buf[o:o + n] for k, (o, n) in self.layout.items()}

  def load_params(self, params: Dict[str, Any]) -> None:
    """MaxText tree -> the flat f32 state (weight_dtype float32, base.yml:130).  uint16 numpy arrays are bf16 bit patterns."""
    views = self.param_views()
    for k, v in views.items():
      src = params[k] if k in params else params[k.replace("mlp/mlp_layer_norm", "post_self_attention_layer_norm")]
      if isinstance(src, torch.Tensor):
        t = src.to(self.device, torch.float32)
      else:
        a = np.asarray(src)
        t = (torch.from_numpy(a.view(np.int16)).view(torch.bfloat16).float() if a.dtype == np.uint16 else torch.from_numpy(a.astype(np.float32))).to(self.device)
      v.copy_(t.reshape(-1))
    _check(self.lib, self.lib.ixt_train_refresh_bf16(self._h, _sp()))

  # ---- one step
  def forward_backward(self, batch: Dict[str, Any]) -> torch.Tensor:
    """-> device tensor [total_loss, total_weights] of THIS rank's rows; gradients of total_loss in self.grads."""
    arr = {}
    for k in ("inputs", "inputs_position", "inputs_segmentation", "targets", "targets_segmentation"):
      x = batch[k]
      x = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x, dtype=np.int32))
      x = x.to(self.device, torch.int32).contiguous()
      if tuple(x.shape) != (self.B, self.S):
        raise ValueError(f"batch[{k}] has shape {tuple(x.shape)}, expected {(self.B, self.S)}")
      arr[k] = x
    _check(self.lib, self.lib.ixt_train_forward_backward(self._h, *[ctypes.c_void_p(arr[k].data_ptr()) for k in
                                                                  ("inputs", "inputs_position", "inputs_segmentation", "targets", "targets_segmentation")],
                                                         ctypes.c_void_p(self._lw.data_ptr()), _sp()))
    return self._lw

  def train_step(self, batch: Dict[str, Any]) -> Dict[str, float]:
    """train.py:352-472 for gradient_accumulation_steps = 1: returns the scalar metrics of :453-460."""
    cfg = self.config
    lw = self.forward_backward(batch)
    _dp.exchange_gradients(self.grads, lw, algorithm=self.exchange)            # sums over the data-parallel ranks (no-op on one rank)
    total_loss, total_weights = (float(v) for v in lw.tolist())
    lr = self.schedule(self.step)                                              # optax evaluates the schedule at the count BEFORE the update
    _check(self.lib, self.lib.ixt_train_apply(self._h, total_weights, float(cfg.gradient_clipping_threshold), lr, float(cfg.adam_b1), float(cfg.adam_b2),
                                              float(cfg.adam_eps), float(cfg.adam_eps_root), float(cfg.adam_weight_decay), self.step + 1,
                                              ctypes.c_void_p(self._norms.data_ptr()), _sp()))
    self.step += 1
    raw, clipped = (float(v) for v in self._norms.tolist())
    return {"learning/loss": total_loss / (total_weights + EPS), "learning/total_weights": total_weights, "learning/raw_grad_norm": raw,
            "learning/grad_norm": clipped, "learning/current_learning_rate": lr}


def train_loop(config: Any, data_iterator, steps: Optional[int] = None, trainer: Optional[Trainer] = None, params: Optional[Dict[str, Any]] = None):
  """train.py:600-720 reduced to its arithmetic: `steps` train steps over the iterator's batches; returns (trainer, [metrics])."""
  from . import synth  # pylint: disable=import-outside-toplevel
  trainer = trainer or Trainer(config)
  if params is not None or trainer.step == 0:
    trainer.load_params(params if params is not None else synth.synth_weights(config, int(config.ixt_synthetic_seed)))
  out = []
  for _ in range(int(steps if steps is not None else config.steps)):
    out.append(trainer.train_step(next(data_iterator)))
  return trainer, out
