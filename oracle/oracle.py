"""CPU ORACLE wrapper (test infrastructure, NOT product code).

Loads oracle/_build/libixt_oracle.so (built from ixt_oracle.c by oracle/Makefile) and restates,
in numpy / pure Python, the integer parts of the path:

  * PageState / page-manager functions   /root/reference/MaxText/inference/page_manager.py:48-412
  * engine call order + result layout    /root/reference/MaxText/maxengine.py:400-574, 838-936, 1166-1192
  * RoPE tables                          /root/reference/MaxText/layers/embeddings.py:270-307
  * audio-id <-> embedding-row mapping   /root/reference/vocab_expansion/extend_tokenizer.py:29-46

PARITY UNPINNED for the floating-point path (no JAX here, no numeric goldens in the reference's
tests).  PINNED for the integer page manager by the reference's own test
cases (tests/test_page_manager_oracle.py re-states tests/inference/page_manager_test.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libixt_oracle.so")
_lib = None


def build(force: bool = False) -> str:
  """Compile the C restatement with gcc (oracle/Makefile)."""
  if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "ixt_oracle.c")):
    subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
  return _SO


_u16p = ctypes.POINTER(ctypes.c_uint16)
_u16pp = ctypes.POINTER(_u16p)
_i32p = ctypes.POINTER(ctypes.c_int32)
_f32p = ctypes.POINTER(ctypes.c_float)


class _CModel(ctypes.Structure):
  _fields_ = [
      ("vocab", ctypes.c_int32), ("emb", ctypes.c_int32), ("nq", ctypes.c_int32), ("nkv", ctypes.c_int32),
      ("hd", ctypes.c_int32), ("mlp", ctypes.c_int32), ("nlayers", ctypes.c_int32),
      ("page", ctypes.c_int32), ("num_pages", ctypes.c_int32), ("max_pages_per_group", ctypes.c_int32),
      ("eps", ctypes.c_float),
      ("embedding", _u16p), ("final_norm", _u16p), ("logits_w", _u16p),
      ("ln1", _u16pp), ("wq", _u16pp), ("wk", _u16pp), ("wv", _u16pp), ("wo", _u16pp),
      ("ln2", _u16pp), ("wi0", _u16pp), ("wi1", _u16pp), ("wout", _u16pp),
      ("rope_sin", _u16p), ("rope_cos", _u16p),
      ("kpages", _u16p), ("vpages", _u16p),
      ("block", ctypes.c_int32), ("attn_cap", ctypes.c_float), ("final_cap", ctypes.c_float), ("window", ctypes.c_int32),
      ("norm_logits", ctypes.c_int32), ("pan", _u16pp), ("pfn", _u16pp),
      ("qn", _u16pp), ("kn", _u16pp), ("q_scalar", ctypes.c_float), ("rope_sin_local", _u16p), ("rope_cos_local", _u16p),
      ("n_exp", ctypes.c_int32), ("top_k", ctypes.c_int32), ("gate", _u16pp),
      ("rope_sin32", _f32p), ("rope_cos32", _f32p),
  ]


def lib():
  global _lib
  if _lib is None:
    build()
    _lib = ctypes.CDLL(_SO)
    _lib.orc_forward.restype = None
    _lib.orc_forward.argtypes = [ctypes.POINTER(_CModel), ctypes.c_int, _i32p, _i32p, _i32p, _i32p, _i32p,
                                 ctypes.c_int, _i32p, _f32p, _i32p, _u16p]
    _lib.orc_rmsnorm.argtypes = [_u16p, _u16p, _u16p, ctypes.c_int, ctypes.c_float]
    _lib.orc_rope.argtypes = [_u16p, ctypes.c_int, ctypes.c_int, _u16p, _u16p]
    _lib.orc_dot_rows.argtypes = [_u16p, ctypes.c_int, ctypes.c_int, _u16p, ctypes.c_int, ctypes.c_int, _f32p, ctypes.c_int]
    _lib.orc_dot_rows_scalar.argtypes = _lib.orc_dot_rows.argtypes
    _lib.orc_mfma_cases.argtypes = [_u16p, _u16p, _f32p, _f32p, ctypes.c_int64]
    _lib.orc_set_sampling_ranges.argtypes = [_i32p, ctypes.c_int]
    _lib.orc_attend_row.argtypes = [ctypes.POINTER(_CModel), ctypes.c_int, _u16p, _i32p, ctypes.c_int, _u16p, _f32p]
    _lib.orc_f32_to_bf16.argtypes = [_f32p, _u16p, ctypes.c_int64]
    _lib.orc_exp_vec.argtypes = [_f32p, _f32p, ctypes.c_int64]
    _lib.orc_swiglu.argtypes = [_u16p, _u16p, _u16p, ctypes.c_int64]
    _lib.orc_geglu.argtypes = [_u16p, _u16p, _u16p, ctypes.c_int64]
    _lib.orc_tanh_vec.argtypes = [_f32p, _f32p, ctypes.c_int64]
    _lib.orc_attend_row_ex.argtypes = [ctypes.POINTER(_CModel), ctypes.c_int, _u16p, _i32p, ctypes.c_int, _u16p, _f32p, ctypes.c_float, ctypes.c_int]
    _lib.orc_log_vec.argtypes = [_f32p, _f32p, ctypes.c_int64]
    _lib.orc_logprob_rows.argtypes = [_f32p, ctypes.c_int, ctypes.c_int, _i32p, _f32p]
    _lib.orc_num_threads.restype = ctypes.c_int
    _lib.orc_set_num_threads.argtypes = [ctypes.c_int]
  return _lib


def _p(a: np.ndarray, t):
  return a.ctypes.data_as(t)


# ----------------------------------------------------------------------------- bf16 helpers
def f32_to_bf16(x: np.ndarray) -> np.ndarray:
  """Round-to-nearest-even f32 -> bf16 bit patterns (uint16)."""
  u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
  r = (u + (0x7FFF + ((u >> 16) & 1))) >> 16
  return r.astype(np.uint16)


def bf16_to_f32(h: np.ndarray) -> np.ndarray:
  return (np.ascontiguousarray(h, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def logprob_rows(logits: np.ndarray, chosen: np.ndarray) -> np.ndarray:
  """inference_utils.log_prob_of_chosen_token (inference_utils.py:56-63) in the canonical order of ixt_oracle.c: f32 [rows]."""
  lg = np.ascontiguousarray(logits, np.float32).reshape(-1, logits.shape[-1])
  ch = np.ascontiguousarray(chosen, np.int32).reshape(-1)
  out = np.zeros((lg.shape[0],), np.float32)
  lib().orc_logprob_rows(_p(lg, _f32p), lg.shape[0], lg.shape[1], _p(ch, _i32p), _p(out, _f32p))
  return out


# ----------------------------------------------------------------------------- RoPE tables
def rope_tables(head_dim: int, max_pos: int, min_timescale: float = 1.0, max_timescale: float = 10000.0):
  """sin/cos as bf16 bit patterns [max_pos][head_dim/2] (embeddings.py:270-307).

  fraction = f32(2i)/f32(D); timescale = f32(min * (max/min) ** fraction) (pow evaluated in f64, rounded
  once to f32); angle = f32(pos) / timescale (f32 divide); sin/cos evaluated in f64 on the f32 angle,
  rounded to f32 then to bf16.
  """
  half = head_dim // 2
  frac = (2 * np.arange(half, dtype=np.int32)).astype(np.float32) / np.float32(head_dim)
  ts = (np.float64(min_timescale) * np.power(np.float64(max_timescale) / np.float64(min_timescale), frac.astype(np.float64))).astype(np.float32)
  pos = np.arange(max_pos, dtype=np.int32).astype(np.float32)
  ang = (pos[:, None] / ts[None, :]).astype(np.float32)
  sn = np.sin(ang.astype(np.float64)).astype(np.float32)
  cs = np.cos(ang.astype(np.float64)).astype(np.float32)
  return f32_to_bf16(sn), f32_to_bf16(cs)


def rope_tables_llama31(head_dim: int, max_pos: int, min_timescale: float = 1.0, max_timescale: float = 10000.0, use_scale: bool = True):
  """LLaMARotaryEmbedding's sin / cos as f32 [max_pos][head_dim/2] (embeddings.py:396-443, :474-479): timescale of the pair (2i, 2i+1) as in
  rope_tables; with use_scale the frequency 1 / timescale is kept for wavelengths below 8192 / 4, divided by 8 above 8192, blended
  smoothly between (scale_factor 8, low / high frequency factors 1 / 4, original context 8192), all in f32; angle = f32(pos) / timescale;
  sin / cos evaluated in f64 on the f32 angle and rounded to f32."""
  half = head_dim // 2
  frac = (2 * np.arange(half, dtype=np.int32)).astype(np.float32) / np.float32(head_dim)
  ts = (np.float64(min_timescale) * np.power(np.float64(max_timescale) / np.float64(min_timescale), frac.astype(np.float64))).astype(np.float32)
  if use_scale:
    one, f32 = np.float32(1.0), np.float32
    freq = one / ts
    wavelen = f32(2.0) * f32(3.14159265358979323846) / freq
    smooth = (f32(8192.0) / wavelen - f32(1.0)) / (f32(4.0) - f32(1.0))
    blended = (one - smooth) * freq / f32(8.0) + smooth * freq
    freq = np.where(wavelen < f32(8192.0) / f32(4.0), freq, np.where(wavelen > f32(8192.0) / f32(1.0), freq / f32(8.0), blended)).astype(np.float32)
    ts = (one / freq).astype(np.float32)
  pos = np.arange(max_pos, dtype=np.int32).astype(np.float32)
  ang = (pos[:, None] / ts[None, :]).astype(np.float32)
  return np.sin(ang.astype(np.float64)).astype(np.float32), np.cos(ang.astype(np.float64)).astype(np.float32)


# ----------------------------------------------------------------------------- page manager
@dataclass
class PageState:
  """page_manager.py:48-91 -- same fields, numpy int32 / bool."""
  page_status: np.ndarray
  page_map: np.ndarray
  num_pages_used: np.ndarray
  sequence_lengths: np.ndarray
  active_page: np.ndarray
  has_active_page: np.ndarray
  active_page_position: np.ndarray

  def copy(self) -> "PageState":
    return PageState(*[getattr(self, f).copy() for f in self.__dataclass_fields__])


def initialize_page_state(num_pages: int, max_page_groups: int, max_pages_per_group: int) -> PageState:
  """page_manager.py:93-126 (page 0 permanently marked used)."""
  st = np.zeros((num_pages,), np.int32)
  st[0] = 1
  return PageState(st, np.zeros((max_page_groups, max_pages_per_group), np.int32), np.zeros((max_page_groups,), np.int32),
                   np.zeros((max_page_groups,), np.int32), np.zeros((max_page_groups,), np.int32),
                   np.zeros((max_page_groups,), np.bool_), np.zeros((max_page_groups,), np.int32))


def find_next_free_page_index(page_status: np.ndarray) -> int:
  """page_manager.py:129-156: lowest index >= 1 with status 0, else -1."""
  free = np.nonzero(page_status[1:] == 0)[0]
  return int(free[0]) + 1 if free.size else -1


def release_pages_for_group(s: PageState, g: int, max_pages_per_group: int) -> PageState:
  """page_manager.py:159-205."""
  s = s.copy()
  nvalid = int(s.num_pages_used[g])
  for i in range(max_pages_per_group):
    idx = int(s.page_map[g, i])
    if i < nvalid and idx > 0:
      s.page_status[idx] = 0
  s.num_pages_used[g] = 0
  s.sequence_lengths[g] = 0
  s.active_page[g] = 0
  s.has_active_page[g] = False
  s.active_page_position[g] = 0
  return s


def reserve_pages_for_group(released: PageState, g: int, true_length: int, tokens_per_page: int, max_pages_per_group: int) -> PageState:
  """page_manager.py:208-311."""
  need = (true_length + tokens_per_page - 1) // tokens_per_page
  next_write = ((true_length - 1) % tokens_per_page + 1) % tokens_per_page
  nfree = int(np.sum(released.page_status == 0))
  if not (nfree >= need and need <= max_pages_per_group):
    return released
  s = released.copy()
  for i in range(need):
    nf = find_next_free_page_index(s.page_status)
    if nf >= 0:
      s.page_status[nf] = 1
      s.page_map[g, i] = nf
      s.num_pages_used[g] += 1
  s.sequence_lengths[g] = true_length
  s.active_page[g] = s.page_map[g, need - 1]
  s.has_active_page[g] = True
  s.active_page_position[g] = next_write
  return s


def update_decode_pages_global(s: PageState, tokens_per_page: int, max_pages_per_group: int) -> PageState:
  """page_manager.py:331-412."""
  s = s.copy()
  act = s.has_active_page
  new_len = s.sequence_lengths + act.astype(np.int32)
  new_pos = np.where(act, (new_len - 1) % tokens_per_page, s.active_page_position).astype(np.int32)
  required = (new_len + tokens_per_page - 1) // tokens_per_page
  needs = act & (required > s.num_pages_used) & (required <= max_pages_per_group)
  s.sequence_lengths = new_len.astype(np.int32)
  s.active_page_position = new_pos
  for g in range(s.sequence_lengths.shape[0]):
    nf = find_next_free_page_index(s.page_status)
    if bool(needs[g]) and nf >= 0:
      s.page_status[nf] = 1
      s.page_map[g, int(s.num_pages_used[g])] = nf
      s.num_pages_used[g] += 1
      s.active_page[g] = nf
  return s


def extend_pages_for_group(s: PageState, g: int, total_length: int, tokens_per_page: int, max_pages_per_group: int) -> PageState:
  """BUILD EXTENSION (no reference counterpart): keep the group's pages and grow the allocation to
  `total_length` tokens -- what a chunked prefill over pages needs for chunk 2..n.  First-fit like
  reserve; on insufficient resources the state is returned unchanged."""
  need = (total_length + tokens_per_page - 1) // tokens_per_page
  have = int(s.num_pages_used[g])
  extra = max(0, need - have)
  nfree = int(np.sum(s.page_status == 0))
  if not (nfree >= extra and need <= max_pages_per_group):
    return s
  s = s.copy()
  for i in range(have, need):
    nf = find_next_free_page_index(s.page_status)
    s.page_status[nf] = 1
    s.page_map[g, i] = nf
    s.num_pages_used[g] += 1
  s.sequence_lengths[g] = total_length
  s.active_page[g] = s.page_map[g, need - 1]
  s.has_active_page[g] = True
  s.active_page_position[g] = ((total_length - 1) % tokens_per_page + 1) % tokens_per_page
  return s


class PageManager:
  """page_manager.py:415-621 (validation + dispatch)."""

  def __init__(self, num_pages, tokens_per_page, max_target_length, max_page_groups, max_pages_per_group):
    self.num_pages, self.tokens_per_page, self.max_target_length = num_pages, tokens_per_page, max_target_length
    self.max_page_groups, self.max_pages_per_group = max_page_groups, max_pages_per_group
    if self.max_pages_per_group <= 0:
      raise ValueError("`pagedattn_max_pages_per_group` must be positive.")
    if self.max_pages_per_group < (max_target_length + tokens_per_page - 1) // tokens_per_page:
      raise ValueError("`pagedattn_max_pages_per_group` is insufficient for `max_target_length`.")
    if num_pages <= 1:
      raise ValueError("`pagedattn_num_pages` must be greater than 1.")
    if tokens_per_page <= 0:
      raise ValueError("`pagedattn_tokens_per_page` must be positive.")
    if max_page_groups <= 0:
      raise ValueError("`pagedattn_max_page_groups` must be positive.")

  def get_initial_page_state(self):
    return initialize_page_state(self.num_pages, self.max_page_groups, self.max_pages_per_group)

  def _check_group(self, g):
    if g < 0 or g >= self.max_page_groups:
      raise ValueError(f"PageManager: page_group_id ({g}) out of range [0, {self.max_page_groups})")

  def update_prefill_pages(self, page_state, page_group_id, true_length):
    self._check_group(page_group_id)
    if true_length <= 0 or true_length > self.max_target_length:
      raise ValueError(f"PageManager: true_length ({true_length}) out of range (0, {self.max_target_length}]")
    rel = release_pages_for_group(page_state, page_group_id, self.max_pages_per_group)
    return reserve_pages_for_group(rel, page_group_id, true_length, self.tokens_per_page, self.max_pages_per_group)

  def extend_prefill_pages(self, page_state, page_group_id, total_length):
    self._check_group(page_group_id)
    if total_length <= 0 or total_length > self.max_target_length:
      raise ValueError(f"PageManager: true_length ({total_length}) out of range (0, {self.max_target_length}]")
    return extend_pages_for_group(page_state, page_group_id, total_length, self.tokens_per_page, self.max_pages_per_group)

  def update_decode_pages(self, page_state):
    return update_decode_pages_global(page_state, self.tokens_per_page, self.max_pages_per_group)

  def release_pages(self, page_state, page_group_id):
    self._check_group(page_group_id)
    return release_pages_for_group(page_state, page_group_id, self.max_pages_per_group)


# ----------------------------------------------------------------------------- audio-id mapping
def audio_id_to_embedding_row(a: int) -> int:
  """extend_tokenizer.py:29-46 as realised in audio_token_mapping_adjusted.json:
  a < 99 -> 6 + a (the <unusedN> rows); a >= 99 -> 256001 + (a - 99)."""
  return 6 + a if a < 99 else 256001 + (a - 99)


# ----------------------------------------------------------------------------- attention masks (boolean, exact)
def reference_prefill_mask(q_len: int, kv_len: int, next_pos: int = 0, q_segment_ids=None, kv_segment_ids=None) -> np.ndarray:
  """attentions.py:586-618 (`generate_attention_mask`, prefill / train mode, global attention): segment equality AND the SARATHI causal
  rule `col <= row + next_pos` (next_pos = length of the previous chunks).  Returns bool [q_len, kv_len], True = attend."""
  row = np.arange(q_len, dtype=np.int32)[:, None]
  col = np.arange(kv_len, dtype=np.int32)[None, :]
  m = col <= row + next_pos
  if q_segment_ids is not None:
    m = m & (np.asarray(q_segment_ids)[:, None] == np.asarray(kv_segment_ids)[None, :])
  return m


def ragged_paged_mask(q_len: int, kv_len: int) -> np.ndarray:
  """paged_attention_kernel_v2.py:100-103 (`ref_ragged_paged_attention`): masked where (kv_len - q_len) + row < col."""
  row = np.arange(q_len, dtype=np.int32)[:, None]
  col = np.arange(kv_len, dtype=np.int32)[None, :]
  return ~((kv_len - q_len) + row < col)


def row_visibility(row_slot, row_pos, n_keys: int) -> np.ndarray:
  """What THIS build's row descriptors express: row r (slot s_r, position p_r) attends to positions [0, p_r] of slot s_r.
  Returns bool [R, nslots, n_keys]."""
  row_slot, row_pos = np.asarray(row_slot), np.asarray(row_pos)
  nslots = int(row_slot.max()) + 1
  vis = np.zeros((len(row_slot), nslots, n_keys), bool)
  for r, (s, p) in enumerate(zip(row_slot, row_pos)):
    vis[r, s, :p + 1] = True
  return vis


# ----------------------------------------------------------------------------- model
@dataclass
class OracleConfig:
  vocab_size: int
  emb_dim: int
  num_query_heads: int
  num_kv_heads: int
  head_dim: int
  mlp_dim: int
  num_decoder_layers: int
  normalization_layer_epsilon: float = 1e-5
  rope_min_timescale: float = 1.0
  rope_max_timescale: float = 10000.0
  max_target_length: int = 2048
  max_prefill_predict_length: int = 64
  pagedattn_num_pages: int = 64
  pagedattn_tokens_per_page: int = 32
  pagedattn_max_pages_per_group: int = -1
  max_slots: int = 1
  # Gemma-2 block (configs/models/gemma2-2b.yml): num_decoder_layers counts the reference's merged
  # [local, global] layers; the model then has 2 x num_decoder_layers (attention, MLP) sub-layers and as many K/V page pools
  decoder_block: str = "llama2"
  attn_logits_soft_cap: float = 0.0
  final_logits_soft_cap: float = 0.0
  sliding_window_size: int = 0
  use_post_attn_norm: bool = False
  use_post_ffw_norm: bool = False
  logits_via_embedding: bool = False
  normalize_embedding_logits: bool = True
  # Gemma-3 block (layers/gemma3.py:35-200; configs/models/gemma3-4b.yml): one (attention, MLP) pair per layer, layer i local unless
  # i % 6 == 5; query / key RMSNorm over head_dim; q * query_pre_attn_scalar after RoPE; the local layers' own RoPE base
  local_rope_max_timescale: float = -1.0
  rope_type: str = "default"          # "llama3.1": LLaMARotaryEmbedding (embeddings.py:354-488)
  rope_use_scale: bool = True
  query_pre_attn_scalar: float = 0.0
  # Mixtral block (layers/mixtral.py:52-170, layers/moe.py; configs/models/mixtral-8x7b.yml): the llama2 wiring with a routed mixture of
  # num_experts MLPs, num_experts_per_tok of them per token
  num_experts: int = 1
  num_experts_per_tok: int = 1

  @property
  def sublayers(self) -> int:
    return self.num_decoder_layers * (2 if self.decoder_block == "gemma2" else 1)

  def __post_init__(self):
    if self.pagedattn_max_pages_per_group <= 0:
      p = self.pagedattn_tokens_per_page
      self.pagedattn_max_pages_per_group = (self.max_target_length + p - 1) // p


class OracleModel:
  """Weights are a dict MaxText-tree-path -> bf16 bit-pattern arrays (uint16), unscanned layout
  (`params/decoder/layers_<i>/...`, decoders.py:762-789)."""

  def __init__(self, cfg: OracleConfig, weights: Dict[str, np.ndarray]):
    self.cfg = cfg
    self.lib = lib()
    L = cfg.sublayers
    gemma = cfg.decoder_block == "gemma2"
    gemma3 = cfg.decoder_block == "gemma3"
    self._keep: List[np.ndarray] = []

    def w(path, shape):
      a = np.ascontiguousarray(weights[path])
      assert a.dtype == np.uint16, path
      assert a.size == int(np.prod(shape)), (path, a.shape, shape)
      self._keep.append(a)
      return a

    def arr(paths_shapes):
      ptrs = (_u16p * L)()
      for i, (pth, shp) in enumerate(paths_shapes):
        ptrs[i] = _p(w(pth, shp), _u16p)
      self._keep.append(ptrs)
      return ctypes.cast(ptrs, _u16pp)

    E, Q, KV, D, F, V = cfg.emb_dim, cfg.num_query_heads, cfg.num_kv_heads, cfg.head_dim, cfg.mlp_dim, cfg.vocab_size
    lp = "params/decoder/layers_%d/"
    m = _CModel()
    m.vocab, m.emb, m.nq, m.nkv, m.hd, m.mlp, m.nlayers = V, E, Q, KV, D, F, L
    m.page, m.num_pages, m.max_pages_per_group = cfg.pagedattn_tokens_per_page, cfg.pagedattn_num_pages, cfg.pagedattn_max_pages_per_group
    m.eps = cfg.normalization_layer_epsilon
    m.embedding = _p(w("params/token_embedder/embedding", (V, E)), _u16p)
    m.final_norm = _p(w("params/decoder/decoder_norm/scale", (E,)), _u16p)
    if (gemma or gemma3) and cfg.logits_via_embedding:   # decoders.py:552-558: the transposed embedding table is the logits kernel
      tied = np.ascontiguousarray(np.asarray(weights["params/token_embedder/embedding"]).reshape(V, E).T)
      self._keep.append(tied)
      m.logits_w = _p(tied, _u16p)
    else:
      m.logits_w = _p(w("params/decoder/logits_dense/kernel", (E, V)), _u16p)
    if gemma:
      # layers/gemma2.py:69-236 parameter names; sub-layer 2 i = "local", 2 i + 1 = "global" of reference layer i
      def names(fmt_local, fmt_global=None):
        out = []
        for i in range(cfg.num_decoder_layers):
          out.append(lp % i + fmt_local.format("local"))
          out.append(lp % i + (fmt_global or fmt_local).format("global"))
        return out
      m.ln1 = arr([(n, (E,)) for n in names("pre_self_attention_norm_{}/scale")])
      m.wq = arr([(n, (E, Q, D)) for n in names("self_attention_{}/query/kernel")])
      m.wk = arr([(n, (E, KV, D)) for n in names("self_attention_{}/key/kernel")])
      m.wv = arr([(n, (E, KV, D)) for n in names("self_attention_{}/value/kernel")])
      m.wo = arr([(n, (Q, D, E)) for n in names("self_attention_{}/out/kernel")])
      m.ln2 = arr([(n, (E,)) for n in names("mlp_{}/mlp_layer_norm/scale", "pre_ffw_norm_{}/scale")])
      m.wi0 = arr([(n, (E, F)) for n in names("mlp_{}/wi_0/kernel")])
      m.wi1 = arr([(n, (E, F)) for n in names("mlp_{}/wi_1/kernel")])
      m.wout = arr([(n, (F, E)) for n in names("mlp_{}/wo/kernel")])
      if cfg.use_post_attn_norm:
        m.pan = arr([(n, (E,)) for n in names("post_self_attention_norm_{}/scale")])
      if cfg.use_post_ffw_norm:
        m.pfn = arr([(n, (E,)) for n in names("post_ffw_norm_{}/scale")])
      m.block = 1
      m.attn_cap = float(cfg.attn_logits_soft_cap or 0.0)
      m.final_cap = float(cfg.final_logits_soft_cap or 0.0) if cfg.logits_via_embedding else 0.0   # decoders.py:552-565: inside the tied branch
      m.window = int(cfg.sliding_window_size or 0)
      m.norm_logits = 1 if (cfg.logits_via_embedding and cfg.normalize_embedding_logits) else 0
    elif cfg.decoder_block == "mixtral":
      NE = cfg.num_experts
      m.ln1 = arr([(lp % i + "pre_self_attention_layer_norm/scale", (E,)) for i in range(L)])
      m.wq = arr([(lp % i + "self_attention/query/kernel", (E, Q, D)) for i in range(L)])
      m.wk = arr([(lp % i + "self_attention/key/kernel", (E, KV, D)) for i in range(L)])
      m.wv = arr([(lp % i + "self_attention/value/kernel", (E, KV, D)) for i in range(L)])
      m.wo = arr([(lp % i + "self_attention/out/kernel", (Q, D, E)) for i in range(L)])
      m.ln2 = arr([(lp % i + "post_self_attention_layer_norm/scale", (E,)) for i in range(L)])   # mixtral.py:123-130
      m.gate = arr([(lp % i + "MoeBlock_0/gate/kernel", (E, NE)) for i in range(L)])
      m.wi0 = arr([(lp % i + "MoeBlock_0/wi_0", (NE, E, F)) for i in range(L)])
      m.wi1 = arr([(lp % i + "MoeBlock_0/wi_1", (NE, E, F)) for i in range(L)])
      m.wout = arr([(lp % i + "MoeBlock_0/wo", (NE, F, E)) for i in range(L)])
      m.block, m.n_exp, m.top_k = 3, NE, cfg.num_experts_per_tok
    elif gemma3:
      # layers/gemma3.py:83-200 parameter names (the MLP pre-norm lives inside mlp_block: use_pre_norm=True)
      m.ln1 = arr([(lp % i + "pre_self_attention_norm/scale", (E,)) for i in range(L)])
      m.wq = arr([(lp % i + "self_attention/query/kernel", (E, Q, D)) for i in range(L)])
      m.wk = arr([(lp % i + "self_attention/key/kernel", (E, KV, D)) for i in range(L)])
      m.wv = arr([(lp % i + "self_attention/value/kernel", (E, KV, D)) for i in range(L)])
      m.wo = arr([(lp % i + "self_attention/out/kernel", (Q, D, E)) for i in range(L)])
      m.qn = arr([(lp % i + "self_attention/query_norm/scale", (D,)) for i in range(L)])
      m.kn = arr([(lp % i + "self_attention/key_norm/scale", (D,)) for i in range(L)])
      m.ln2 = arr([(lp % i + "mlp/mlp_layer_norm/scale", (E,)) for i in range(L)])
      m.wi0 = arr([(lp % i + "mlp/wi_0/kernel", (E, F)) for i in range(L)])
      m.wi1 = arr([(lp % i + "mlp/wi_1/kernel", (E, F)) for i in range(L)])
      m.wout = arr([(lp % i + "mlp/wo/kernel", (F, E)) for i in range(L)])
      if cfg.use_post_attn_norm:
        m.pan = arr([(lp % i + "post_self_attention_norm/scale", (E,)) for i in range(L)])
      if cfg.use_post_ffw_norm:
        m.pfn = arr([(lp % i + "post_ffw_norm/scale", (E,)) for i in range(L)])
      m.block = 2
      m.attn_cap = float(cfg.attn_logits_soft_cap or 0.0)
      m.final_cap = float(cfg.final_logits_soft_cap or 0.0) if cfg.logits_via_embedding else 0.0
      m.window = int(cfg.sliding_window_size or 0)
      m.norm_logits = 1 if (cfg.logits_via_embedding and cfg.normalize_embedding_logits) else 0
      m.q_scalar = float(cfg.query_pre_attn_scalar or 0.0)
      if cfg.local_rope_max_timescale > 0:   # attentions.py:2085-2088
        self.rope_sin_local, self.rope_cos_local = rope_tables(D, cfg.max_target_length, cfg.rope_min_timescale, cfg.local_rope_max_timescale)
        m.rope_sin_local, m.rope_cos_local = _p(self.rope_sin_local, _u16p), _p(self.rope_cos_local, _u16p)
    else:
      m.ln1 = arr([(lp % i + "pre_self_attention_layer_norm/scale", (E,)) for i in range(L)])
      m.wq = arr([(lp % i + "self_attention/query/kernel", (E, Q, D)) for i in range(L)])
      m.wk = arr([(lp % i + "self_attention/key/kernel", (E, KV, D)) for i in range(L)])
      m.wv = arr([(lp % i + "self_attention/value/kernel", (E, KV, D)) for i in range(L)])
      m.wo = arr([(lp % i + "self_attention/out/kernel", (Q, D, E)) for i in range(L)])
      m.ln2 = arr([(lp % i + "mlp/mlp_layer_norm/scale", (E,)) for i in range(L)])
      m.wi0 = arr([(lp % i + "mlp/wi_0/kernel", (E, F)) for i in range(L)])
      m.wi1 = arr([(lp % i + "mlp/wi_1/kernel", (E, F)) for i in range(L)])
      m.wout = arr([(lp % i + "mlp/wo/kernel", (F, E)) for i in range(L)])
    self.rope_sin, self.rope_cos = rope_tables(D, cfg.max_target_length, cfg.rope_min_timescale, cfg.rope_max_timescale)
    m.rope_sin, m.rope_cos = _p(self.rope_sin, _u16p), _p(self.rope_cos, _u16p)
    if str(cfg.rope_type).lower().startswith("llama3.1"):
      self.rope_sin32, self.rope_cos32 = rope_tables_llama31(D, cfg.max_target_length, cfg.rope_min_timescale, cfg.rope_max_timescale, bool(cfg.rope_use_scale))
      m.rope_sin32, m.rope_cos32 = _p(self.rope_sin32, _f32p), _p(self.rope_cos32, _f32p)
    n = L * cfg.pagedattn_num_pages * cfg.pagedattn_tokens_per_page * KV * D
    self.kpages = np.zeros((n,), np.uint16)
    self.vpages = np.zeros((n,), np.uint16)
    m.kpages, m.vpages = _p(self.kpages, _u16p), _p(self.vpages, _u16p)
    self.c = m

  def forward(self, tokens, slot, pos, active, page_map, logits_rows=(), want_logits=False, want_hidden=False):
    """One ragged batch of rows (orc_forward). Returns dict(argmax, logits, hidden)."""
    R = len(tokens)
    tk = np.ascontiguousarray(tokens, np.int32)
    sl = np.ascontiguousarray(slot, np.int32)
    ps = np.ascontiguousarray(pos, np.int32)
    ac = np.ascontiguousarray(active, np.int32)
    pm = np.ascontiguousarray(page_map, np.int32)
    lr = np.ascontiguousarray(list(logits_rows), np.int32)
    nl = lr.shape[0]
    V = self.cfg.vocab_size
    logits = np.zeros((nl, V), np.float32) if (want_logits and nl) else None
    amax = np.zeros((max(nl, 1),), np.int32)
    hidden = np.zeros((R, self.cfg.emb_dim), np.uint16) if want_hidden else None
    self.lib.orc_forward(ctypes.byref(self.c), R, _p(tk, _i32p), _p(sl, _i32p), _p(ps, _i32p), _p(ac, _i32p), _p(pm, _i32p),
                         nl, _p(lr, _i32p) if nl else None, _p(logits, _f32p) if logits is not None else None,
                         _p(amax, _i32p), _p(hidden, _u16p) if hidden is not None else None)
    return {"argmax": amax[:nl].copy(), "logits": logits, "hidden": hidden}

  def gather_kv(self, layer: int, page_row: Sequence[int], length: int):
    """Dense [length][nkv][hd] bf16 views of a slot's K and V (for cache-parity tests)."""
    c = self.cfg
    P, KV, D = c.pagedattn_tokens_per_page, c.num_kv_heads, c.head_dim
    pe = P * KV * D
    base = layer * c.pagedattn_num_pages * pe
    k = np.zeros((length, KV, D), np.uint16)
    v = np.zeros((length, KV, D), np.uint16)
    for j in range(length):
      off = base + int(page_row[j // P]) * pe + (j % P) * KV * D
      k[j] = self.kpages[off:off + KV * D].reshape(KV, D)
      v[j] = self.vpages[off:off + KV * D].reshape(KV, D)
    return k, v


class OracleEngine:
  """Call-order restatement of MaxEngine over the paged oracle model
  (maxengine.py:533-574 prefill, :838-936 generate, :1166-1192 insert, :1320-1328 release_pages,
  :1370-1453 init_decode_state).  Result rows are [token, valid=1, generated_tokens] (:506-518, :915-926)."""

  def __init__(self, model: OracleModel):
    self.model = model
    c = model.cfg
    self.cfg = c
    self.pm = PageManager(c.pagedattn_num_pages, c.pagedattn_tokens_per_page, c.max_target_length, c.max_slots,
                          c.pagedattn_max_pages_per_group)
    self.page_state = self.pm.get_initial_page_state()

  @staticmethod
  def set_sampling_ranges(ranges):
    """Greedy sampling restricted to half-open column ranges [(lo, hi), ...] (library-wide state of the oracle; [] lifts it)."""
    ranges = list(ranges or [])
    arr = np.array([v for r in ranges for v in r] or [0], np.int32)
    lib().orc_set_sampling_ranges(_p(arr, _i32p), len(ranges))

  def prefill(self, padded_tokens, true_length, slot=0, existing_prefix_tokens=None, want_logits=False, want_prompt_logp=False):
    start = 0 if existing_prefix_tokens is None else len(existing_prefix_tokens)
    if start == 0:
      self.page_state = self.pm.update_prefill_pages(self.page_state, slot, true_length)
    else:
      self.page_state = self.pm.extend_prefill_pages(self.page_state, slot, start + true_length)
    n = int(true_length)
    toks = np.asarray(padded_tokens, np.int32)[:n]
    pos = np.arange(start, start + n, dtype=np.int32)
    prompt_logp = None
    if want_prompt_logp:   # maxengine.py:477-478: log_softmax(logits[t])[tokens[t]] for every row of the chunk
      out = self.model.forward(toks, np.full((n,), slot, np.int32), pos, np.ones((n,), np.int32), self.page_state.page_map,
                               logits_rows=list(range(n)), want_logits=True)
      prompt_logp = logprob_rows(out["logits"], toks)
      out = {"argmax": out["argmax"][n - 1:], "logits": out["logits"][n - 1:] if want_logits else None}
    else:
      out = self.model.forward(toks, np.full((n,), slot, np.int32), pos, np.ones((n,), np.int32), self.page_state.page_map,
                               logits_rows=[n - 1], want_logits=want_logits)
    first = int(out["argmax"][0])
    prefix = {"logits": out["logits"], "next_pos": np.full((1, 1), start + n, np.int32), "prompt_logp": prompt_logp,
              "generated_tokens": np.zeros((1, 1), np.int32), "tokens": np.full((1, 1), first, np.int32), "slot": slot}
    result = np.array([[first, 1, 0]], np.int32)
    return prefix, result

  def prefill_many(self, prompts, slots=None, want_logits=False):
    """Several prompts as ONE ragged batch of rows (the packed prefill of maxengine.py:719-836 `prefill_concat`, over pages):
    prompt i = (padded_tokens, true_length) goes to slots[i]; rows of different prompts never see each other (each row
    attends to positions [0, pos] of its own slot).  Returns ([prefix], first_tokens int32[n], logits f32[n][V] or None)."""
    n = len(prompts)
    slots = list(range(n)) if slots is None else list(slots)
    for (toks, tl), s in zip(prompts, slots):
      self.page_state = self.pm.update_prefill_pages(self.page_state, s, int(tl))
    tk = np.concatenate([np.asarray(t, np.int32)[:int(tl)] for t, tl in prompts])
    sl = np.concatenate([np.full((int(tl),), s, np.int32) for (_, tl), s in zip(prompts, slots)])
    ps = np.concatenate([np.arange(int(tl), dtype=np.int32) for _, tl in prompts])
    last = np.cumsum([int(tl) for _, tl in prompts]) - 1
    out = self.model.forward(tk, sl, ps, np.ones_like(tk), self.page_state.page_map, logits_rows=list(last), want_logits=want_logits)
    firsts = out["argmax"].astype(np.int32)
    prefixes = []
    for i, ((_, tl), s) in enumerate(zip(prompts, slots)):
      prefixes.append({"logits": out["logits"][i:i + 1] if want_logits else None, "next_pos": np.full((1, 1), int(tl), np.int32),
                       "generated_tokens": np.zeros((1, 1), np.int32), "tokens": np.full((1, 1), int(firsts[i]), np.int32), "slot": s})
    return prefixes, firsts, out["logits"]

  def init_decode_state(self):
    B = self.cfg.max_slots
    return {"next_pos": np.zeros((B, 1), np.int32), "generated_tokens": np.zeros((B, 1), np.int32),
            "tokens": np.zeros((B, 1), np.int32), "logits": None}

  def insert(self, prefix, decode_state, slot):
    decode_state["next_pos"][slot] = prefix["next_pos"][0]
    decode_state["generated_tokens"][slot] = prefix["generated_tokens"][0]
    decode_state["tokens"][slot] = prefix["tokens"][0]
    self.page_state.has_active_page[slot] = True
    return decode_state

  def generate(self, decode_state, want_logits=False):
    B = self.cfg.max_slots
    self.page_state = self.pm.update_decode_pages(self.page_state)
    active = self.page_state.has_active_page.astype(np.int32)
    toks = decode_state["tokens"][:, 0].copy()
    pos = decode_state["next_pos"][:, 0].copy()
    out = self.model.forward(toks, np.arange(B, dtype=np.int32), pos, active, self.page_state.page_map,
                             logits_rows=list(range(B)), want_logits=want_logits)
    new = out["argmax"].astype(np.int32)
    ds = {"tokens": new.reshape(B, 1).copy(), "next_pos": decode_state["next_pos"] + 1,
          "generated_tokens": decode_state["generated_tokens"] + 1, "logits": out["logits"]}
    result = np.stack([new, np.ones((B,), np.int32), ds["generated_tokens"][:, 0]], axis=1).astype(np.int32)
    return ds, result

  def release_pages(self, slot):
    self.page_state = self.pm.release_pages(self.page_state, slot)
