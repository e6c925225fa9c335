"""MaxEngine over libixt.so -- same method names, argument names and result structures as the reference
engine (/root/reference/MaxText/maxengine.py:100-1487) for the prefill / insert / generate path, with
`attention="paged"` semantics.  PyTorch is used for device memory and streams only; all arithmetic is in
the HIP library.  Thread rule mirrors the reference: one engine per GPU, calls externally serialised.

Differences that a caller can observe (all documented in DESIGN.md section 6):
  * prefill writes K/V straight into the slot's pages, so `Prefix["cache"]` is a handle, and `insert`
    is a state copy (the reference's page-copy loop, maxengine.py:1106-1125, is unnecessary);
  * `decode_state` tensors are updated in place and returned (the reference donates them, :868, :1045);
  * padding rows (index >= true_length) are never computed;
  * chunked prefill (`existing_prefix`) extends the slot's pages instead of release+reserve.
"""
from __future__ import annotations

import ctypes
import dataclasses
import os
from typing import Any, Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib
from . import engine_api
from . import synth as _synth
from .inference.page_manager import PageManager, PageState

Params = Any
Prefix = Dict[str, Any]
DecodeState = Dict[str, Any]


@dataclasses.dataclass
class ExistingPrefix:
  """maxengine.py:66-76."""
  cache: Any
  common_prefix_tokens: Any


class PagedCacheHandle:
  """Stands where the reference's cache pytree stands: K/V live in the engine's page pools.  `slot` is the page group that
  holds them (a decode slot, or a staging group of a packed prefill); `groups` / `lengths` describe a packed prefill."""

  def __init__(self, engine: "MaxEngine", slot: Optional[int], length: int, groups=None, lengths=None):
    self.engine, self.slot, self.length = engine, slot, length
    self.groups, self.lengths = groups, lengths

  def gather(self, layer: int):
    """Dense (k, v) bf16 tensors [length, num_kv_heads, head_dim] of this slot."""
    return self.engine.gather_kv(layer, self.slot, self.length)


def _stream_ptr() -> int:
  return torch.cuda.current_stream().cuda_stream


def get_query_pre_attn_scalar(config) -> float:
  """layers/gemma3.py:50-58.  The reference knows three model names; any other gemma3-* name (the small test geometries) takes the
  4b / 12b rule."""
  if config.model_name == "gemma3-27b":
    return (config.base_emb_dim // config.base_num_query_heads) ** -0.5
  if str(config.model_name).startswith("gemma3"):
    return config.head_dim ** -0.5
  raise ValueError(f"Unsupported model name: {config.model_name}")


class MaxEngine(engine_api.Engine):
  """See module docstring; reference: maxengine.py:100."""

  def __init__(self, config: Any, devices: Any | None = None):
    if not torch.cuda.is_available():
      raise _lib.IxtError("MaxEngine needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU path")
    self.config = config
    self.lib = _lib.load()
    if config.attention not in ("paged", "autoselected"):
      raise NotImplementedError(f"attention={config.attention}: this build implements the paged path only")
    if config.quantization not in ("", None, "fp8") or config.quantize_kvcache:
      raise NotImplementedError("quantization: only '' (bf16) and 'fp8' (IXT-FP8-v1: e4m3 projection kernels on the fp8 matrix cores) are built; KV cache quantisation is not")
    if config.decode_sampling_strategy != "greedy":
      raise NotImplementedError("only greedy sampling is built (parity is defined on greedy decode)")
    self.device = torch.device("cuda", torch.cuda.current_device())
    B = int(config.per_device_batch_size)
    if B <= 0:
      raise ValueError("per_device_batch_size must be >= 1")
    c = _lib.IxtConfig()
    c.vocab_size, c.emb_dim = config.vocab_size, config.emb_dim
    c.num_query_heads, c.num_kv_heads, c.head_dim = config.num_query_heads, config.num_kv_heads, config.head_dim
    c.mlp_dim, c.num_decoder_layers = config.mlp_dim, config.num_decoder_layers
    c.normalization_layer_epsilon = config.normalization_layer_epsilon
    c.rope_min_timescale, c.rope_max_timescale = float(config.rope_min_timescale), float(config.rope_max_timescale)
    c.max_prefill_predict_length, c.max_target_length = config.max_prefill_predict_length, config.max_target_length
    c.max_slots = B
    c.pagedattn_num_pages, c.pagedattn_tokens_per_page = config.pagedattn_num_pages, config.pagedattn_tokens_per_page
    c.pagedattn_max_pages_per_group = config.pagedattn_max_pages_per_group
    c.use_hip_graph = 1 if config.ixt_use_hip_graph else 0
    c.max_packed_rows = int(config.ixt_max_packed_rows)
    c.staging_groups = int(config.ixt_staging_groups)
    c.fp8_weights = 1 if config.quantization == "fp8" else 0
    rope_type = str(config.rope_type).lower()      # attentions.py:2052-2071
    if rope_type.startswith("llama3.1") or str(config.model_name).startswith("llama3.1"):
      c.rope_llama31, c.rope_use_scale = 1, 1 if config.rope_use_scale else 0
    elif rope_type != "default":
      raise NotImplementedError(f"rope_type {config.rope_type}: 'default' and 'llama3.1' are built")
    if config.decoder_block in ("gemma2", "gemma3"):   # layers/gemma2.py:69-236, layers/gemma3.py:35-200
      if tuple(config.mlp_activations) != ("gelu", "linear"):
        raise ValueError(f"decoder_block {config.decoder_block} is built with mlp_activations ['gelu', 'linear']")
      c.decoder_block = _lib.BLOCK_GEMMA2 if config.decoder_block == "gemma2" else _lib.BLOCK_GEMMA3
      if config.decoder_block == "gemma3":
        c.local_rope_max_timescale = float(config.local_rope_max_timescale)
        c.query_pre_attn_scalar = float(get_query_pre_attn_scalar(config))
      c.attn_logits_soft_cap = float(config.attn_logits_soft_cap or 0.0)
      c.final_logits_soft_cap = float(config.final_logits_soft_cap or 0.0)
      c.sliding_window_size = int(config.sliding_window_size or 0)
      c.use_post_attn_norm = 1 if config.use_post_attn_norm else 0
      c.use_post_ffw_norm = 1 if config.use_post_ffw_norm else 0
      c.logits_via_embedding = 1 if config.logits_via_embedding else 0
      c.normalize_embedding_logits = 1 if config.normalize_embedding_logits else 0
    elif tuple(config.mlp_activations) != ("silu", "linear"):
      raise ValueError(f"decoder_block {config.decoder_block} is built with mlp_activations ['silu', 'linear']")
    elif config.decoder_block == "mixtral":            # layers/mixtral.py:52-170, layers/moe.py
      c.decoder_block = _lib.BLOCK_MIXTRAL
      c.num_experts, c.num_experts_per_tok = int(config.num_experts), int(config.num_experts_per_tok)
      c.ep_rank, c.ep_size = int(config.ixt_expert_parallel_rank), int(config.ixt_expert_parallel_size)
    self._staging = int(config.ixt_staging_groups)
    self._packed_rows = max(int(config.ixt_max_packed_rows), int(config.max_prefill_predict_length))
    self._ccfg = c
    self._ctx = ctypes.c_void_p()
    _lib.check(self.lib.ixt_create(ctypes.byref(c), ctypes.byref(self._ctx)))
    self._B = B
    self.page_manager = PageManager(config, engine=self)
    self.page_state = self.page_manager.get_initial_page_state()
    self.rng = None
    self._params_loaded = False
    self._result = torch.zeros((B, 3), dtype=torch.int32, device=self.device)
    self._first = torch.zeros((1,), dtype=torch.int32, device=self.device)
    self._page_failures = 0
    self._status_host = torch.zeros((4,), dtype=torch.int32).pin_memory()   # deferred reservation check of the packed prefill
    self._status_event = None
    self._status_what = ""

  def _raise_if_new_failure(self, failures: int, group: int, what: str) -> None:
    if failures != self._page_failures:
      self._page_failures = failures
      raise ValueError(f"{what}: the page pool (pagedattn_num_pages={self.config.pagedattn_num_pages}) could not serve the reservation "
                       f"of page group {group}; the group holds no pages")

  def _snapshot_pages(self, what: str) -> None:
    """The throughput form of _check_pages: the status words are copied to pinned memory in stream order and looked at by the next engine
    call (`insert_partial`, which hands the pages on, at the latest) -- the host does not wait here, so the pass this call enqueues
    starts right behind the previous one."""
    if not self.config.ixt_check_pages:
      return
    self._settle_pages()
    _lib.check(self.lib.ixt_pages_status_async(self._ctx, ctypes.c_void_p(self._status_host.data_ptr()), ctypes.c_void_p(_stream_ptr())))
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream())
    self._status_event, self._status_what = ev, what

  def _settle_pages(self) -> None:
    ev = self._status_event
    if ev is None:
      return
    self._status_event = None
    ev.synchronize()
    self._raise_if_new_failure(int(self._status_host[0]), int(self._status_host[1]), self._status_what)

  def _check_pages(self, what: str) -> None:
    """The reference's page manager leaves a group empty, silently, when the pool cannot serve a reservation
    (page_manager.py:305-311); rows of such a group are not computed here (no write goes through a stale page_map entry) and,
    unless ixt_check_pages is off, the call that asked for the pages raises."""
    if not self.config.ixt_check_pages:
      return
    self._settle_pages()
    out = (ctypes.c_int32 * 4)()
    _lib.check(self.lib.ixt_pages_status(self._ctx, out, 0, ctypes.c_void_p(_stream_ptr())))
    self._raise_if_new_failure(out[0], out[1], what)
    if self.exchange_status() != 0:     # the stream is synchronised here anyway: the cheapest place to notice a decode step that went wrong
      raise _lib.IxtError(f"{what}: a wait inside the split decode attention gave up in an earlier generate step (ixt_exchange_status); "
                          "the tokens of that step are not to be trusted")

  def exchange_status(self, reset: bool = True) -> int:
    """1 if a bounded wait of the split decode attention (few rows: the blocks of one head exchange their softmax statistics inside a launch)
    ever gave up since the last reset -- the numbers of that step are wrong --, else 0.  Synchronises the stream."""
    out = (ctypes.c_int32 * 1)()
    _lib.check(self.lib.ixt_exchange_status(self._ctx, out, 1 if reset else 0, ctypes.c_void_p(_stream_ptr())))
    return int(out[0])

  def close(self) -> None:
    """Give the engine's HBM (weights copy, page pools, workspaces) back now; the engine cannot be used afterwards.  Handles returned by
    earlier calls keep a reference to the engine object, so waiting for the garbage collector can hold 100+ GB for a long time."""
    if getattr(self, "_ctx", None) is not None and self._ctx:
      torch.cuda.synchronize()
      self.lib.ixt_destroy(self._ctx)
      self._ctx = None

  def __del__(self):
    try:
      if getattr(self, "_ctx", None) is not None and self._ctx:
        self.lib.ixt_destroy(self._ctx)
        self._ctx = None
    except Exception:  # pylint: disable=broad-except
      pass

  # ------------------------------------------------------------------ params
  def load_params(self, *args, params: Optional[Dict[str, Any]] = None, rng: Any = None, **kwargs) -> Params:
    """maxengine.py:218-271.  `params`: {MaxText tree path: array}; arrays may be torch (bf16/f32), numpy float32 or
    numpy uint16 (bf16 bit patterns).  With params=None and load_parameters_path == "" the weights are random-init
    (seed `ixt_synthetic_seed`), as the reference's setup_decode_state does (maxtext_utils.py:917-922)."""
    self._param_source = ("tree", params)       # what set_engine_vars_from_base_engine replays into a second engine
    if params is None:
      path = self.config.load_parameters_path
      if path:
        params = _load_checkpoint(path)
        self._param_source = ("path", path)
      else:
        params = _synth.synth_weights(self.config, int(self.config.ixt_synthetic_seed))
        self._param_source = ("synthetic", int(self.config.ixt_synthetic_seed))
    st = _stream_ptr()
    for name, arr in params.items():
      t, dtype = _to_device_tensor(arr, self.device)
      shape = (ctypes.c_int64 * t.dim())(*t.shape)
      _lib.check(self.lib.ixt_load_weight(self._ctx, name.encode(), ctypes.c_void_p(t.data_ptr()), dtype, shape, t.dim(), ctypes.c_void_p(st)))
      t.record_stream(torch.cuda.current_stream())   # the copy kernel may still be reading `t` when this name is dropped
      del t
    torch.cuda.current_stream().synchronize()
    if not self.lib.ixt_weights_complete(self._ctx):
      raise ValueError("load_params: parameter tree is incomplete for the configured model")
    self._params_loaded = True
    return {"params": "resident in HBM (libixt)", "engine": id(self)}

  # ------------------------------------------------------------------ sampling columns
  def set_sampling_ranges(self, ranges) -> None:
    """Restrict greedy sampling to the union of up to 4 half-open column ranges [(lo, hi), ...]; [] or None lifts it.
    With `AudioTokenMapping.audio_row_ranges()` every generated id is the row of an audio code.  Logits stay unmasked."""
    ranges = list(ranges or [])
    arr = (ctypes.c_int32 * max(1, 2 * len(ranges)))(*[int(v) for r in ranges for v in r])
    _lib.check(self.lib.ixt_set_sampling_ranges(self._ctx, arr, len(ranges)))

  def set_audio_only_sampling(self, mapping=None) -> None:
    """Sample audio codes only: ranges from `mapping` (an AudioTokenMapping) or from config.audio_token_mapping_path (base.yml:496)."""
    if mapping is None:
      from .vocab_expansion import AudioTokenMapping  # pylint: disable=import-outside-toplevel
      path = self.config.audio_token_mapping_path
      mapping = AudioTokenMapping.from_npz(path) if path.endswith(".npz") else AudioTokenMapping.from_json(path)
    self.set_sampling_ranges(mapping.audio_row_ranges())

  def _token_logp(self, logits: torch.Tensor, tokens: torch.Tensor) -> torch.Tensor:
    """inference_utils.log_prob_of_chosen_token (inference_utils.py:56-63) of `tokens` [n, 1] under `logits` [n, ..., V]: f32 [n, 1]."""
    n = tokens.shape[0]
    out = torch.empty((n, 1), dtype=torch.float32, device=self.device)
    tok = tokens.reshape(n).to(torch.int32).contiguous()
    _lib.check(self.lib.ixt_op_logprob(self._ctx, ctypes.c_void_p(logits.data_ptr()), n, ctypes.c_void_p(tok.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                       ctypes.c_void_p(_stream_ptr())))
    return out

  # ------------------------------------------------------------------ prefill
  def prefill(self, *, params: Params, existing_prefix: Optional[ExistingPrefix] = None, padded_tokens: Any, images: Any = None,
              true_length: int, sampler: Any = None, rng: Any = None, request_id: Any = None, slot: Optional[int] = None,
              return_prompt_logp: bool = False) -> Tuple[Prefix, engine_api.ResultTokens]:
    """maxengine.py:533-574 (+ _prefill_jit :400-530)."""
    del params, sampler, rng, request_id
    if images is not None:
      raise NotImplementedError("multimodal prefill is outside the built path")
    if slot is None:
      slot = 0
    true_length = int(true_length)
    toks = _as_device_i32(padded_tokens, self.device).reshape(-1)
    start = 0
    if existing_prefix is not None:
      if not self.use_chunked_prefill:
        raise ValueError("Using chunked prefill is needed for existing_prefix.")
      start = int(_numel(existing_prefix.common_prefix_tokens))
    if true_length <= 0 or true_length > toks.numel():
      raise ValueError(f"true_length ({true_length}) out of range (0, {toks.numel()}]")
    if start == 0:
      self.page_state = self.page_manager.update_prefill_pages(page_state=self.page_state, page_group_id=slot, true_length=true_length)
    else:
      self.page_state = self.page_manager.extend_prefill_pages(page_state=self.page_state, page_group_id=slot, total_length=start + true_length)
    self._check_pages("prefill")
    V = self.config.vocab_size
    logits = torch.empty((1, 1, V), dtype=torch.float32, device=self.device)
    first = torch.empty((1, 1), dtype=torch.int32, device=self.device)
    _lib.check(self.lib.ixt_prefill_chunk(self._ctx, slot, ctypes.c_void_p(toks.data_ptr()), true_length, start,
                                          ctypes.c_void_p(first.data_ptr()), ctypes.c_void_p(logits.data_ptr()), ctypes.c_void_p(_stream_ptr())))
    prompt_logp = None
    if return_prompt_logp:     # maxengine.py:477-478: log_softmax(logits[t])[tokens[t]] per position; positions past true_length (padding) hold 0
      prompt_logp = torch.zeros((1, toks.numel()), dtype=torch.float32, device=self.device)
      _lib.check(self.lib.ixt_prefill_prompt_logp(self._ctx, ctypes.c_void_p(toks.data_ptr()), true_length, ctypes.c_void_p(prompt_logp.data_ptr()),
                                                  ctypes.c_void_p(_stream_ptr())))
    generated = torch.zeros((1, 1), dtype=torch.int32, device=self.device)
    data = torch.cat([first, torch.ones_like(first), generated], dim=1)
    token_logp = self._token_logp(logits, first) if self.config.return_log_prob else None     # maxengine.py:502-505
    result = engine_api.ResultTokens(data=data, tokens_idx=(0, 1), valid_idx=(1, 2), length_idx=(2, 3), log_prob=token_logp, samples_per_slot=1)
    next_pos = torch.full((1, 1), start + true_length, dtype=torch.int32, device=self.device)
    prefix = {
        "logits": logits,
        "cache": PagedCacheHandle(self, slot, start + true_length),
        "next_pos": next_pos,
        "generated_tokens": generated,
        "tokens": first,
        "prompt_logp": prompt_logp,
    }
    return prefix, result

  def prefill_multisampling(self, *, params: Params, padded_tokens: Any, true_length: int, sampler: Any = None, rng: Any = None,
                            num_samples: int = 1, slot: Optional[int] = None) -> Tuple[Prefix, engine_api.ResultTokens]:
    """maxengine.py:595-717: one prefill, `num_samples` first tokens drawn from the same last-row logits.  Under the built sampling
    strategy (greedy, inference_utils.py:74-75) every draw is the argmax, so the samples are equal; the shapes are the reference's:
    tokens / generated_tokens [num_samples, 1], ResultTokens.data [num_samples, 3] with samples_per_slot = num_samples, one cache."""
    num_samples = int(num_samples)
    if num_samples < 1:
      raise ValueError(f"num_samples ({num_samples}) must be at least 1")
    prefix, res1 = self.prefill(params=params, padded_tokens=padded_tokens, true_length=true_length, sampler=sampler, rng=rng, slot=slot)
    first = prefix["tokens"].expand(num_samples, 1).contiguous()
    generated = torch.zeros((num_samples, 1), dtype=torch.int32, device=self.device)
    data = torch.cat([first, torch.ones_like(first), generated], dim=1)
    logp = res1.log_prob.expand(num_samples, 1).contiguous() if res1.log_prob is not None else None      # maxengine.py:684-689
    result = engine_api.ResultTokens(data=data, tokens_idx=(0, 1), valid_idx=(1, 2), length_idx=(2, 3), log_prob=logp, samples_per_slot=num_samples)
    prefix = dict(prefix, tokens=first, generated_tokens=generated)
    return prefix, result

  def prefill_concat(self, *, params: Params, existing_prefix: Optional[ExistingPrefix] = None, padded_tokens: Any, decoder_positions: Any,
                     decoder_segment_ids: Any, start_pos: Any, true_lengths: Any, num_prompts: int, sampler: Any = None, rng: Any = None,
                     return_prompt_logp: bool = False):
    """maxengine.py:719-836: several prompts packed in one token sequence share ONE pass over the weights.  Prompt i occupies
    padded_tokens[start_pos[i] : start_pos[i] + true_lengths[i]] (positions decoder_positions, segment id i + 1).  Over pages the
    K/V of prompt i go to staging page group i (config `ixt_staging_groups`); `insert_partial` hands those pages to the slots.
    Returns (cache handle, prefill_results {logits [n,1,1,V], next_pos [n,1,1], generated_tokens [n,1,1], tokens [n,1,1]},
    [ResultTokens] per prompt)."""
    del params, sampler, rng, decoder_segment_ids
    if existing_prefix:
      raise ValueError("We don't know what to do with existing_prefix")
    n = int(num_prompts)
    if n <= 0 or n > self._staging:
      raise ValueError(f"prefill_concat: num_prompts ({n}) out of range (0, ixt_staging_groups={self._staging}]")
    starts = [int(v) for v in _host_ints(start_pos)[:n]]
    lens = [int(v) for v in _host_ints(true_lengths)[:n]]
    toks = _as_device_i32(padded_tokens, self.device).reshape(-1)
    pos_all = _as_device_i32(decoder_positions, self.device).reshape(-1)
    total = sum(lens)
    max_len = max(lens) if lens else 0     # positions of a prompt run 0 .. true_length - 1 (maxengine.py:746-760)
    if any(l <= 0 for l in lens) or any(st < 0 or st + l > toks.numel() for st, l in zip(starts, lens)):
      raise ValueError("prefill_concat: start_pos / true_lengths outside padded_tokens")
    if total > self._packed_rows:
      raise ValueError(f"prefill_concat: {total} rows exceed ixt_max_packed_rows / max_prefill_predict_length ({self._packed_rows})")
    B = self._B
    for i, l in enumerate(lens):
      self.page_state = self.page_manager.update_prefill_pages(page_state=self.page_state, page_group_id=B + i, true_length=l, _staging=True)
    self._snapshot_pages("prefill_concat")
    idx = torch.cat([torch.arange(st, st + l, device=self.device) for st, l in zip(starts, lens)])
    row_tok = toks[idx].contiguous()
    row_pos = pos_all[idx].contiguous()
    row_grp = torch.cat([torch.full((l,), B + i, dtype=torch.int32, device=self.device) for i, l in enumerate(lens)])
    ends = torch.tensor(np.cumsum(lens) - 1, dtype=torch.int32, device=self.device)
    V = self.config.vocab_size
    logits = torch.empty((n, 1, 1, V), dtype=torch.float32, device=self.device)
    first = torch.empty((n,), dtype=torch.int32, device=self.device)
    _lib.check(self.lib.ixt_prefill_rows(self._ctx, total, ctypes.c_void_p(row_tok.data_ptr()), ctypes.c_void_p(row_grp.data_ptr()),
                                         ctypes.c_void_p(row_pos.data_ptr()), max_len, n, ctypes.c_void_p(ends.data_ptr()),
                                         ctypes.c_void_p(first.data_ptr()),
                                         ctypes.c_void_p(logits.data_ptr()), ctypes.c_void_p(_stream_ptr())))
    prompt_logp = None
    if return_prompt_logp:     # maxengine.py:777-778: log_softmax(logits[t])[tokens[t]] at every position of the packed sequence (padding: 0)
      row_logp = torch.empty((total,), dtype=torch.float32, device=self.device)
      _lib.check(self.lib.ixt_prefill_prompt_logp(self._ctx, ctypes.c_void_p(row_tok.data_ptr()), total, ctypes.c_void_p(row_logp.data_ptr()),
                                                  ctypes.c_void_p(_stream_ptr())))
      prompt_logp = torch.zeros((1, toks.numel()), dtype=torch.float32, device=self.device)
      prompt_logp[0, idx] = row_logp
    next_pos = torch.tensor(lens, dtype=torch.int32, device=self.device).reshape(n, 1, 1)
    generated = torch.zeros((n, 1, 1), dtype=torch.int32, device=self.device)
    tokens = first.reshape(n, 1, 1)
    results = []
    logps = self._token_logp(logits, first.reshape(n, 1)) if self.config.return_log_prob else None     # maxengine.py:801-804
    for i in range(n):
      data = torch.cat([tokens[i], torch.ones_like(tokens[i]), generated[i]], dim=1)
      results.append(engine_api.ResultTokens(data=data, tokens_idx=(0, 1), valid_idx=(1, 2), length_idx=(2, 3),
                                             log_prob=logps[i:i + 1] if logps is not None else None, samples_per_slot=1))
    prefill_results = {"logits": logits, "next_pos": next_pos, "generated_tokens": generated, "tokens": tokens, "prompt_logp": prompt_logp}
    cache = PagedCacheHandle(self, None, 0, groups=[B + i for i in range(n)], lengths=lens)
    return cache, prefill_results, results

  # ------------------------------------------------------------------ generate
  def generate(self, params: Params, decode_state: DecodeState, sampler: Any = None, rng: Any = None) -> Tuple[DecodeState, engine_api.ResultTokens]:
    """maxengine.py:838-936: page advance + one token for all slots + state update, in one library call."""
    del params, sampler, rng
    self._settle_pages()
    lg = decode_state.get("logits")
    if self.config.return_log_prob and not isinstance(lg, torch.Tensor):     # the log-probabilities are taken from the step's rounded logits
      lg = torch.empty((self._B, 1, self.config.vocab_size), dtype=torch.float32, device=self.device)
      decode_state["logits"] = lg
    lg_ptr = ctypes.c_void_p(lg.data_ptr()) if isinstance(lg, torch.Tensor) else None
    _lib.check(self.lib.ixt_generate(self._ctx, ctypes.c_void_p(decode_state["tokens"].data_ptr()),
                                     ctypes.c_void_p(decode_state["next_pos"].data_ptr()),
                                     ctypes.c_void_p(decode_state["generated_tokens"].data_ptr()),
                                     ctypes.c_void_p(self._result.data_ptr()), lg_ptr, ctypes.c_void_p(_stream_ptr())))
    data = self._result.clone()
    token_logp = self._token_logp(lg, data[:, 0:1]) if self.config.return_log_prob else None   # maxengine.py:907-910 (idle slots: of their stale logits)
    result = engine_api.ResultTokens(data=data, tokens_idx=(0, 1), valid_idx=(1, 2), length_idx=(2, 3), log_prob=token_logp, samples_per_slot=1)
    return decode_state, result

  # ------------------------------------------------------------------ insert
  def insert(self, prefix: Prefix, decode_state: DecodeState, slot: int, request_id: Any = None) -> DecodeState:
    """maxengine.py:1166-1192.  Into the slot the prefix was prefilled into: no K/V moves (prefill wrote the slot's pages in place).
    Into another slot: reserve pages there and copy the prefix's pages (the reference's page copy loop, :1106-1125)."""
    del request_id
    if slot < 0 or slot >= self._B:
      raise ValueError(f"slot {slot} out of range [0, {self._B})")
    handle = prefix["cache"]
    st = ctypes.c_void_p(_stream_ptr())
    if isinstance(handle, PagedCacheHandle) and handle.slot is not None and handle.length > 0:
      if handle.slot != slot:
        _lib.check(self.lib.ixt_pages_copy(self._ctx, handle.slot, slot, handle.length, st))
        self._check_pages("insert")
      _lib.check(self.lib.ixt_pages_resync(self._ctx, slot, handle.length, st))
    decode_state["next_pos"][slot] = prefix["next_pos"][0]
    decode_state["generated_tokens"][slot] = prefix["generated_tokens"][0]
    decode_state["tokens"][slot] = prefix["tokens"][0]
    if isinstance(decode_state.get("logits"), torch.Tensor) and prefix.get("logits") is not None:
      decode_state["logits"][slot] = prefix["logits"][0]
    _lib.check(self.lib.ixt_pages_set_active(self._ctx, slot, 1, st))
    return decode_state

  def bulk_insert(self, prefix: Prefix, decode_state: DecodeState, slots: list[int]) -> DecodeState:
    """maxengine.py:946-1043: one computed prefix into several slots (each gets its own copy of the prefix's pages)."""
    for s in slots:
      decode_state = self.insert(prefix, decode_state, s)
    return decode_state

  def insert_partial(self, prefix: Any, decode_state: DecodeState, cache: Any, slots: Any, *, start_indices: Any = None, num_prompts: int,
                     seq_len: int = 0) -> DecodeState:
    """maxengine.py:1206-1318: the prompts of a packed prefill into their slots.  `prefix` / `cache` are prefill_concat's
    prefill_results / cache handle; prompt i goes to slots[i].  Over pages this moves page lists (no K/V byte is copied)."""
    del start_indices, seq_len
    if not isinstance(cache, PagedCacheHandle) or cache.groups is None:
      raise ValueError("insert_partial: `cache` must be the handle returned by prefill_concat")
    self._settle_pages()       # a reservation of the packed prefill that the pool could not serve surfaces here, before its pages move
    n = int(num_prompts)
    slot_list = [int(v) for v in _host_ints(slots)[:n]]
    if n > len(cache.groups) or len(slot_list) < n:
      raise ValueError("insert_partial: num_prompts exceeds the packed prefill / slots")
    st = ctypes.c_void_p(_stream_ptr())
    for i, slot in enumerate(slot_list):
      if slot < 0 or slot >= self._B:
        raise ValueError(f"slot {slot} out of range [0, {self._B})")
      _lib.check(self.lib.ixt_pages_move(self._ctx, cache.groups[i], slot, st))
      decode_state["next_pos"][slot] = prefix["next_pos"][i, 0]
      decode_state["generated_tokens"][slot] = prefix["generated_tokens"][i, 0]
      decode_state["tokens"][slot] = prefix["tokens"][i, 0]
      if isinstance(decode_state.get("logits"), torch.Tensor) and prefix.get("logits") is not None:
        decode_state["logits"][slot] = prefix["logits"][i, 0]
    return decode_state

  # ------------------------------------------------------------------ state
  def init_decode_state(self, *args, rng: Any = None, **kwargs) -> DecodeState:
    """maxengine.py:1370-1453.  `logits` ([B,1,V] f32, 1 MB per slot at V=264192) is materialised only when the caller
    asks (`materialize_logits=True`); the field is present either way."""
    del args, rng
    B = self._B
    z = lambda: torch.zeros((B, 1), dtype=torch.int32, device=self.device)
    logits = None
    if kwargs.get("materialize_logits", False):
      logits = torch.zeros((B, 1, self.config.vocab_size), dtype=torch.float32, device=self.device)
    return {"logits": logits, "cache": PagedCacheHandle(self, None, 0), "next_pos": z(), "generated_tokens": z(), "tokens": z()}

  def release_pages(self, slot: int):
    """maxengine.py:1320-1328."""
    self.page_state = self.page_manager.release_pages(page_state=self.page_state, page_group_id=slot)

  def gather_kv(self, layer: int, slot: int, length: int):
    c = self.config
    k = torch.empty((length, c.num_kv_heads, c.head_dim), dtype=torch.bfloat16, device=self.device)
    v = torch.empty_like(k)
    _lib.check(self.lib.ixt_kv_gather(self._ctx, layer, slot, length, ctypes.c_void_p(k.data_ptr()), ctypes.c_void_p(v.data_ptr()),
                                      ctypes.c_void_p(_stream_ptr())))
    return k, v

  def get_prefix_destination_sharding(self) -> Any:
    return None  # single device per engine: nothing to shard (maxengine.py:1330-1337)

  def get_tokenizer(self):
    raise NotImplementedError("tokenizer plumbing (jetstream token_utils) is outside the built path; pass token ids")

  def build_tokenizer(self, metadata):
    raise NotImplementedError("tokenizer plumbing (jetstream token_utils) is outside the built path; pass token ids")

  @property
  def max_concurrent_decodes(self) -> int:
    return self._B

  @property
  def max_prefill_length(self) -> int:
    return int(self.config.max_prefill_predict_length)

  @property
  def use_chunked_prefill(self) -> bool:
    return bool(self.config.use_chunked_prefill)

  @property
  def prefill_chunk_size(self) -> int:
    return int(self.config.prefill_chunk_size)

  @property
  def samples_per_slot(self) -> int:
    return 1

  @property
  def mesh(self):
    return None

  @property
  def colocated_cpus(self) -> None:
    raise NotImplementedError


# ---------------------------------------------------------------------- helpers
def _numel(x) -> int:
  if hasattr(x, "numel"):
    return int(x.numel())
  return int(np.asarray(x).size)


def _host_ints(x):
  if isinstance(x, torch.Tensor):
    return x.detach().reshape(-1).cpu().tolist()
  return np.asarray(x).reshape(-1).tolist()


def _as_device_i32(x, device) -> torch.Tensor:
  if isinstance(x, torch.Tensor):
    return x.to(device=device, dtype=torch.int32).contiguous()
  return torch.from_numpy(np.ascontiguousarray(np.asarray(x), dtype=np.int32)).to(device)


def _to_device_tensor(arr, device):
  """-> (contiguous device tensor, ixt_dtype).  uint16 numpy arrays are bf16 bit patterns."""
  if isinstance(arr, torch.Tensor):
    if arr.dtype == torch.bfloat16:
      return arr.to(device).contiguous(), _lib.IXT_BF16
    if arr.dtype == torch.float32:
      return arr.to(device).contiguous(), _lib.IXT_F32
    raise ValueError(f"unsupported parameter dtype {arr.dtype}")
  a = np.asarray(arr)
  if a.dtype == np.uint16:
    t = torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).to(device)
    return t, _lib.IXT_BF16
  if a.dtype == np.float32:
    return torch.from_numpy(np.ascontiguousarray(a)).to(device), _lib.IXT_F32
  raise ValueError(f"unsupported parameter dtype {a.dtype}")


def _load_checkpoint(path: str) -> Dict[str, Any]:
  """Flat checkpoint keyed by MaxText tree paths: .safetensors or .npz (the reference's Orbax format is not read)."""
  if path.endswith(".safetensors"):
    from safetensors.torch import load_file  # pylint: disable=import-outside-toplevel
    return dict(load_file(path))
  if path.endswith(".npz"):
    with np.load(path) as z:
      return {k: z[k] for k in z.files}
  raise ValueError(f"load_parameters_path {path}: expected .safetensors or .npz")


def set_engine_vars_from_base_engine(engine: MaxEngine, base_engine: MaxEngine, rng: Any = None) -> None:
  """maxengine.py:1490-1506: a second engine takes over what the base engine has already set up after loading the checkpoint.  There it
  is sharding / cache annotations; here the one thing an engine holds besides its own pages is the loaded weights, which each ixt context
  keeps in its own layout -- so the base engine's parameters are loaded into `engine` (a device-to-device copy)."""
  del rng
  source = getattr(base_engine, "_param_source", None)
  if source is None or not getattr(base_engine, "_params_loaded", False):
    raise ValueError("set_engine_vars_from_base_engine: the base engine has no parameters loaded")
  kind, what = source
  if kind == "tree":
    engine.load_params(params=what)
  elif kind == "path":
    engine.load_params(params=_load_checkpoint(what))
  else:
    engine.load_params(params=_synth.synth_weights(engine.config, what))


def create_engine_from_config_flags(maxengine_config_filepath, batch_size, max_prefill_predict_length, max_target_length, args_str) -> MaxEngine:
  """maxengine.py:1509-1545: a new engine with the given batch size and lengths plus the `k=v` pairs of args_str (load_parameters_path
  among them, as the reference asserts).  The reference's defaults that select parallelism or the dense dot-product cache have no
  counterpart here: this engine is one GPU and attention="paged"."""
  from . import pyconfig  # pylint: disable=import-outside-toplevel
  args = {
      "scan_layers": "false",
      "attention": "paged",
      "max_prefill_predict_length": f"{max_prefill_predict_length}",
      "max_target_length": f"{max_target_length}",
      "per_device_batch_size": f"{batch_size}",
  }
  for item in (args_str or "").split(" "):
    if item:
      k, v = item.split("=")
      args[k.strip()] = v.strip()
  if "load_parameters_path" not in args:
    raise AssertionError("load_parameters_path must be defined")
  if maxengine_config_filepath is None:
    maxengine_config_filepath = os.path.join(pyconfig.PKG_DIR, "configs", "base.yml")
  argv = ["maxengine_server.py", maxengine_config_filepath] + [f"{k}={v}" for k, v in args.items()]
  return MaxEngine(pyconfig.initialize(argv))
