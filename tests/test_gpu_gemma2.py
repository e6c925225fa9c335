"""GPU parity of the Gemma-2 block against the oracle through the engine's call order: tokens, f32 logits
bits (after / sqrt(emb) and the soft cap), K and V of every sub-layer, PageState.  The tiny geometry has every Gemma-2 feature on and a
sliding window (24) shorter than the sequences, so the local sub-layers really mask."""
import numpy as np
import pytest

import helpers as H
from test_gpu_engine import packed_prefill_pair, run_pair

pytestmark = pytest.mark.gpu


def _cfg(**kw):
  base = dict(max_prefill_predict_length=48, max_target_length=128)
  base.update(kw)
  return H.make_config("gemma2-tiny", **base)


def test_geglu_op_matches_oracle():
  import torch
  from oracle import oracle as O
  cfg = _cfg()
  eng, _ = H.make_engine(cfg, H.synth.synth_weights(cfg, 0))
  rng = np.random.default_rng(2)
  a = H.synth.f32_to_bf16_bits(np.concatenate([rng.standard_normal(70000).astype(np.float32) * 4, np.linspace(-12, 12, 4001, dtype=np.float32)]))
  b = H.synth.f32_to_bf16_bits(rng.standard_normal(a.size).astype(np.float32))
  want = np.zeros_like(a)
  O.lib().orc_geglu(a.ctypes.data_as(O._u16p), b.ctypes.data_as(O._u16p), want.ctypes.data_as(O._u16p), a.size)
  ta, tb = H.dev_bf16(a), H.dev_bf16(b)
  out = torch.empty_like(ta)
  import ctypes
  H.lib.check(eng.lib.ixt_op_geglu(eng._ctx, ctypes.c_void_p(ta.data_ptr()), ctypes.c_void_p(tb.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                   a.size, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
  torch.cuda.synchronize()
  assert np.array_equal(H.bits(out), want)


def test_gemma2_single_slot_window_active():
  cfg = _cfg()
  w = H.synth.synth_weights(cfg, 0)
  toks, _ = H.synth.synth_prompt(cfg.vocab_size, 48, 1234)
  run_pair(cfg, w, [(toks, 41)], steps=40)


@pytest.mark.parametrize("use_graph", [True, False])
def test_gemma2_batched_ragged(use_graph):
  cfg = _cfg(per_device_batch_size=6.0, pagedattn_num_pages=160, ixt_use_hip_graph=use_graph)
  w = H.synth.synth_weights(cfg, 1)
  prompts = []
  for s in range(5):
    toks, _ = H.synth.synth_prompt(cfg.vocab_size, 48, 100 + s)
    prompts.append((toks, [48, 1, 25, 24, 33][s]))
  run_pair(cfg, w, prompts, steps=36)


def test_gemma2_chunked_prefill():
  cfg = _cfg(max_prefill_predict_length=16, per_device_batch_size=2.0, use_chunked_prefill=True, prefill_chunk_size=16)
  w = H.synth.synth_weights(cfg, 2)
  toks = H.synth.synth_prompt(cfg.vocab_size, 48, 77)[0]
  toks2 = H.synth.synth_prompt(cfg.vocab_size, 48, 78)[0]
  run_pair(cfg, w, [(toks, 44), (toks2, 30)], steps=12, chunks=[[16, 16, 12], [16, 14]])


@pytest.mark.parametrize("variant", ["no_post_norms", "untied_head", "no_caps_no_window", "no_normalize"])
def test_gemma2_feature_switches(variant):
  kw = {
      "no_post_norms": dict(use_post_attn_norm=False, use_post_ffw_norm=False),
      "untied_head": dict(logits_via_embedding=False),
      "no_caps_no_window": dict(attn_logits_soft_cap=0.0, final_logits_soft_cap=0.0, sliding_window_size=0),
      "no_normalize": dict(normalize_embedding_logits=False),
  }[variant]
  cfg = _cfg(per_device_batch_size=2.0, **kw)
  w = H.synth.synth_weights(cfg, 4)
  prompts = [(H.synth.synth_prompt(cfg.vocab_size, 48, 50 + s)[0], [37, 29][s]) for s in range(2)]
  run_pair(cfg, w, prompts, steps=16)


def test_gemma2_many_rows_ring_gemm_shapes():
  """130 decode slots and a 130-row-plus prefill: the projections take the LDS-ring GEMM with the f32 epilogue (rows >= 128) instead of the
  split-K dots; head_dim 128 with 2 query heads per kv head."""
  cfg = H.make_config("gemma2-tiny", max_prefill_predict_length=160, max_target_length=192, per_device_batch_size=130.0, pagedattn_num_pages=130 * 24 + 8,
                      head_dim=128, base_num_query_heads=4, base_num_kv_heads=2, sliding_window_size=100)
  w = H.synth.synth_weights(cfg, 6)
  prompts = []
  for s in range(130):
    toks, _ = H.synth.synth_prompt(cfg.vocab_size, 160, 900 + s)
    prompts.append((toks, 150 if s == 0 else [9, 17, 33, 2][s % 4]))
  run_pair(cfg, w, prompts, steps=3, check_logits=False)


@pytest.mark.parametrize("nprompts,hi", [(6, 32), (24, 12)])
def test_gemma2_packed_prefill(nprompts, hi):
  """prefill_concat + insert_partial with the Gemma-2 block: several prompts in one ragged pass (98 rows: split-K dots; 150 rows: ring GEMMs),
  window 24 shorter than the longest prompt."""
  packed_prefill_pair(nprompts, hi, model_name="gemma2-tiny")


def test_gemma2_head_dim_256_tiny():
  """head_dim 256 (gemma2-2b's): two RoPE pairs per lane in k_qkv_finalize, 16 features per lane in the PV kernel."""
  cfg = _cfg(head_dim=256, per_device_batch_size=3.0, pagedattn_num_pages=96)
  w = H.synth.synth_weights(cfg, 8)
  prompts = [(H.synth.synth_prompt(cfg.vocab_size, 48, 60 + s)[0], [45, 8, 26][s]) for s in range(3)]
  run_pair(cfg, w, prompts, steps=30)


@pytest.mark.parametrize("tile", [1, 0])
def test_gemma2_head_dim_256_tiled_prefill(tile):
  """Prefill rows through k_attn_tile at head_dim 256 (the key in two halves of d, the chain's partial sum parked in the score row), with the soft
  cap and a sliding window (40) shorter than the 96- and 70-token prompts, against the oracle; tile = 0: the per-row scores / PV / combine kernels."""
  cfg = _cfg(head_dim=256, max_prefill_predict_length=96, max_target_length=160, per_device_batch_size=2.0, pagedattn_num_pages=96, sliding_window_size=40)
  w = H.synth.synth_weights(cfg, 12)
  prompts = [(H.synth.synth_prompt(cfg.vocab_size, 96, 90 + s)[0], [96, 70][s]) for s in range(2)]
  run_pair(cfg, w, prompts, steps=8, options={"gemma_tile": tile})


def test_gemma2_2b_widths():
  """The widths of configs/models/gemma2-2b.yml (emb 2304, 8 x 256 query heads over 4 kv heads, mlp 9216, vocab 256128, tied head, caps 50 / 30)
  with 2 of its 13 merged layers: every projection shape, the 256 128-column unembedding of the transposed table and head_dim 256 at 2 query heads
  per kv head against the oracle."""
  cfg = H.make_config("gemma2-2b", base_num_decoder_layers=2, max_prefill_predict_length=32, max_target_length=64, per_device_batch_size=2.0,
                      pagedattn_num_pages=40, sliding_window_size=20)
  w = H.synth.synth_weights(cfg, 10)
  prompts = [(H.synth.synth_prompt(cfg.vocab_size, 32, 80 + s)[0], [27, 6][s]) for s in range(2)]
  run_pair(cfg, w, prompts, steps=5)


def test_head_dim_256_needs_the_gemma2_block():
  cfg = H.make_config(head_dim=256, max_prefill_predict_length=16, max_target_length=32)
  maxengine = H.importlib.import_module("maxtext-indextts2_amd.maxengine")
  with pytest.raises(Exception, match="head_dim 256"):
    maxengine.MaxEngine(cfg)
