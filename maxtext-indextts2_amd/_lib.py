"""ctypes binding of libixt.so (include/ixt.h).  There is NO fallback: if the HIP library is missing or a
call fails, the caller gets an exception -- the product path never routes through a CPU implementation."""
from __future__ import annotations

import ctypes
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "libixt.so")

IXT_OK, IXT_ERR_INVALID, IXT_ERR_HIP, IXT_ERR_UNSUPPORTED, IXT_ERR_NOT_LOADED, IXT_ERR_RANGE = 0, -1, -2, -3, -4, -5
IXT_BF16, IXT_F32 = 0, 1

c_i32p = ctypes.POINTER(ctypes.c_int32)
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_u16p = ctypes.POINTER(ctypes.c_uint16)
c_u8p = ctypes.POINTER(ctypes.c_uint8)
c_f32p = ctypes.POINTER(ctypes.c_float)
c_f64p = ctypes.POINTER(ctypes.c_double)
vp = ctypes.c_void_p


class IxtConfig(ctypes.Structure):
  _fields_ = [
      ("vocab_size", ctypes.c_int32), ("emb_dim", ctypes.c_int32), ("num_query_heads", ctypes.c_int32),
      ("num_kv_heads", ctypes.c_int32), ("head_dim", ctypes.c_int32), ("mlp_dim", ctypes.c_int32),
      ("num_decoder_layers", ctypes.c_int32), ("normalization_layer_epsilon", ctypes.c_float),
      ("rope_min_timescale", ctypes.c_float), ("rope_max_timescale", ctypes.c_float),
      ("max_prefill_predict_length", ctypes.c_int32), ("max_target_length", ctypes.c_int32), ("max_slots", ctypes.c_int32),
      ("pagedattn_num_pages", ctypes.c_int32), ("pagedattn_tokens_per_page", ctypes.c_int32),
      ("pagedattn_max_pages_per_group", ctypes.c_int32), ("use_hip_graph", ctypes.c_int32),
      ("max_packed_rows", ctypes.c_int32), ("staging_groups", ctypes.c_int32),
      ("decoder_block", ctypes.c_int32), ("attn_logits_soft_cap", ctypes.c_float), ("final_logits_soft_cap", ctypes.c_float),
      ("sliding_window_size", ctypes.c_int32), ("use_post_attn_norm", ctypes.c_int32), ("use_post_ffw_norm", ctypes.c_int32),
      ("logits_via_embedding", ctypes.c_int32), ("normalize_embedding_logits", ctypes.c_int32),
      ("local_rope_max_timescale", ctypes.c_float), ("query_pre_attn_scalar", ctypes.c_float),
      ("num_experts", ctypes.c_int32), ("num_experts_per_tok", ctypes.c_int32),
      ("ep_rank", ctypes.c_int32), ("ep_size", ctypes.c_int32), ("fp8_weights", ctypes.c_int32),
      ("rope_llama31", ctypes.c_int32), ("rope_use_scale", ctypes.c_int32),
  ]

BLOCK_LLAMA2, BLOCK_GEMMA2, BLOCK_GEMMA3, BLOCK_MIXTRAL = 0, 1, 2, 3


# name -> (restype, argtypes); every symbol include/ixt.h declares
SIGNATURES = {
    "ixt_last_error": (ctypes.c_char_p, []),
    "ixt_create": (ctypes.c_int, [ctypes.POINTER(IxtConfig), ctypes.POINTER(vp)]),
    "ixt_destroy": (ctypes.c_int, [vp]),
    "ixt_load_weight": (ctypes.c_int, [vp, ctypes.c_char_p, vp, ctypes.c_int, c_i64p, ctypes.c_int, vp]),
    "ixt_weights_complete": (ctypes.c_int, [vp]),
    "ixt_pages_reset": (ctypes.c_int, [vp, vp]),
    "ixt_pages_reserve": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, vp]),
    "ixt_pages_extend": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, vp]),
    "ixt_pages_advance": (ctypes.c_int, [vp, vp]),
    "ixt_pages_release": (ctypes.c_int, [vp, ctypes.c_int, vp]),
    "ixt_pages_set_active": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, vp]),
    "ixt_pages_resync": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, vp]),
    "ixt_pages_move": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, vp]),
    "ixt_pages_copy": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]),
    "ixt_pages_status": (ctypes.c_int, [vp, c_i32p, ctypes.c_int, vp]),
    "ixt_pages_status_async": (ctypes.c_int, [vp, vp, vp]),
    "ixt_exchange_status": (ctypes.c_int, [vp, c_i32p, ctypes.c_int, vp]),
    "ixt_prefill_prompt_logp": (ctypes.c_int, [vp, vp, ctypes.c_int, vp, vp]),
    "ixt_build_id": (ctypes.c_char_p, []),
    "ixt_prefill_rows": (ctypes.c_int, [vp, ctypes.c_int, vp, vp, vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp]),
    "ixt_op_logprob": (ctypes.c_int, [vp, vp, ctypes.c_int, vp, vp, vp]),
    "ixt_pages_export": (ctypes.c_int, [vp, c_i32p, c_i32p, c_i32p, c_i32p, c_i32p, c_u8p, c_i32p, vp]),
    "ixt_pages_import_status": (ctypes.c_int, [vp, c_i32p, vp]),
    "ixt_prefill_chunk": (ctypes.c_int, [vp, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, vp, vp, vp]),
    "ixt_decode_step": (ctypes.c_int, [vp, vp, vp, vp, vp, vp]),
    "ixt_generate": (ctypes.c_int, [vp, vp, vp, vp, vp, vp, vp]),
    "ixt_op_dot": (ctypes.c_int, [vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]),
    "ixt_op_rmsnorm": (ctypes.c_int, [vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, vp]),
    "ixt_op_rope": (ctypes.c_int, [vp, vp, vp, ctypes.c_int, ctypes.c_int, vp]),
    "ixt_op_moe_route": (ctypes.c_int, [vp, ctypes.c_int, vp, ctypes.c_int, vp, vp, vp]),
    "ixt_op_moe_expert": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, vp, vp]),
    "ixt_op_moe_combine": (ctypes.c_int, [vp, vp, vp, ctypes.c_int, vp, vp]),
    "ixt_op_moe_experts": (ctypes.c_int, [vp, ctypes.c_int, vp, vp, ctypes.c_int, vp, vp]),
    "ixt_op_swiglu": (ctypes.c_int, [vp, vp, vp, vp, ctypes.c_int64, vp]),
    "ixt_op_geglu": (ctypes.c_int, [vp, vp, vp, vp, ctypes.c_int64, vp]),
    "ixt_op_residual": (ctypes.c_int, [vp, vp, vp, vp, ctypes.c_int64, vp]),
    "ixt_op_embed": (ctypes.c_int, [vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]),
    "ixt_op_kv_write": (ctypes.c_int, [vp, ctypes.c_int, vp, vp, vp, vp, vp, ctypes.c_int, vp]),
    "ixt_op_paged_attention": (ctypes.c_int, [vp, ctypes.c_int, vp, vp, vp, vp, ctypes.c_int, vp, vp]),
    "ixt_op_unembed_argmax": (ctypes.c_int, [vp, vp, ctypes.c_int, vp, vp, vp]),
    "ixt_op_cvt_check": (ctypes.c_int, [vp, c_i64p, vp]),
    "ixt_op_dot_fp8": (ctypes.c_int, [vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]),
    "ixt_op_fp8_roundtrip": (ctypes.c_int, [vp, vp, vp, ctypes.c_int64, vp]),
    "ixt_kv_gather": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp]),
    "ixt_rope_tables": (ctypes.c_int, [vp, c_u16p, c_u16p]),
    "ixt_rope_tables_f32": (ctypes.c_int, [vp, vp, vp]),
    "ixt_timing_enable": (ctypes.c_int, [vp, ctypes.c_int]),
    "ixt_timing_reset": (ctypes.c_int, [vp]),
    "ixt_timing_get": (ctypes.c_int, [vp, ctypes.c_char_p, c_f64p, c_i64p, c_f64p]),
    "ixt_set_option": (ctypes.c_int, [vp, ctypes.c_char_p, ctypes.c_int]),
    "ixt_set_sampling_ranges": (ctypes.c_int, [vp, vp, ctypes.c_int]),
}

_lib = None


class IxtError(RuntimeError):
  pass


def load():
  """Load libixt.so; raises if it has not been built (run __graft_entry__.build())."""
  global _lib
  if _lib is None:
    if not os.path.exists(LIB_PATH):
      raise IxtError(f"{LIB_PATH} is missing: build the HIP extension first (python __graft_entry__.py build); "
                     "there is no CPU fallback")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
      fn = getattr(lib, name)  # AttributeError if the symbol is not exported
      fn.restype = res
      fn.argtypes = args
    # a binary built from other sources than the ones next to it is refused (the built .so is not under version control)
    from . import build as _build  # pylint: disable=import-outside-toplevel
    have, want = lib.ixt_build_id().decode(), _build.source_hash()
    if have != want:
      raise IxtError(f"{LIB_PATH} was built from other sources (library {have}, tree {want}): rebuild it (python __graft_entry__.py build)")
    _lib = lib
  return _lib


def check(rc: int):
  """Map ixt_status to the reference's exception conventions (error conventions)."""
  if rc == IXT_OK:
    return
  msg = load().ixt_last_error().decode("utf-8", "replace")
  if rc in (IXT_ERR_INVALID, IXT_ERR_RANGE):
    raise ValueError(msg)
  if rc == IXT_ERR_UNSUPPORTED:
    raise NotImplementedError(msg)
  raise IxtError(f"ixt status {rc}: {msg}")
