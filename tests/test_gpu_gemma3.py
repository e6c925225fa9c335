"""GPU parity of the Gemma-3 block (layers/gemma3.py:35-200) against the oracle through the engine's call
order: tokens, f32 logits bits, K and V of every layer, PageState.  The tiny geometry has 7 layers (local x 5, global, local), a window (24)
shorter than the sequences, two RoPE bases, the query / key head norms and q * head_dim ** -0.5."""
import numpy as np
import pytest

import helpers as H
from test_gpu_engine import packed_prefill_pair, run_pair

pytestmark = pytest.mark.gpu


def _cfg(**kw):
  base = dict(max_prefill_predict_length=48, max_target_length=128)
  base.update(kw)
  return H.make_config("gemma3-tiny", **base)


def _prompt(vocab, n, seed):
  """n random ids, no padding zeros inside any true length (synth_prompt pads its tail with id 0, which a tied head then echoes for ever)."""
  return np.random.Generator(np.random.PCG64(seed)).integers(1, vocab, size=n).astype(np.int32)


def test_gemma3_single_slot_window_active():
  cfg = _cfg()
  w = H.synth.synth_weights(cfg, 0)
  toks = _prompt(cfg.vocab_size, 48, 1234)
  run_pair(cfg, w, [(toks, 41)], steps=40)


@pytest.mark.parametrize("use_graph", [True, False])
def test_gemma3_batched_ragged(use_graph):
  cfg = _cfg(per_device_batch_size=6.0, pagedattn_num_pages=160, ixt_use_hip_graph=use_graph)
  w = H.synth.synth_weights(cfg, 1)
  prompts = []
  for s in range(5):
    toks = _prompt(cfg.vocab_size, 48, 100 + s)
    prompts.append((toks, [48, 1, 25, 24, 33][s]))
  run_pair(cfg, w, prompts, steps=36)


def test_gemma3_chunked_prefill():
  cfg = _cfg(max_prefill_predict_length=16, per_device_batch_size=2.0, use_chunked_prefill=True, prefill_chunk_size=16)
  w = H.synth.synth_weights(cfg, 2)
  toks = _prompt(cfg.vocab_size, 48, 77)
  toks2 = _prompt(cfg.vocab_size, 48, 78)
  run_pair(cfg, w, [(toks, 44), (toks2, 30)], steps=12, chunks=[[16, 16, 12], [16, 14]])


@pytest.mark.parametrize("variant", ["no_post_norms", "one_rope_base", "no_window", "with_caps"])
def test_gemma3_feature_switches(variant):
  kw = {
      "no_post_norms": dict(use_post_attn_norm=False, use_post_ffw_norm=False),
      "one_rope_base": dict(local_rope_max_timescale=-1),
      "no_window": dict(sliding_window_size=0),
      "with_caps": dict(attn_logits_soft_cap=50.0, final_logits_soft_cap=30.0),
  }[variant]
  cfg = _cfg(per_device_batch_size=2.0, **kw)
  w = H.synth.synth_weights(cfg, 4)
  prompts = [(_prompt(cfg.vocab_size, 48, 50 + s), [37, 29][s]) for s in range(2)]
  run_pair(cfg, w, prompts, steps=16)


def test_gemma3_many_rows_ring_gemm_shapes():
  """130 decode slots and 130-row-plus prefills: the projections take the LDS-ring GEMM with the f32 epilogue (rows >= 128); head_dim 128
  (one RoPE pair per lane in the head norm) with 2 query heads per kv head."""
  cfg = H.make_config("gemma3-tiny", max_prefill_predict_length=160, max_target_length=192, per_device_batch_size=130.0, pagedattn_num_pages=130 * 24 + 8,
                      head_dim=128, base_num_query_heads=4, base_num_kv_heads=2, sliding_window_size=100, base_num_decoder_layers=6)
  w = H.synth.synth_weights(cfg, 6)
  prompts = []
  for s in range(130):
    toks = _prompt(cfg.vocab_size, 160, 900 + s)
    prompts.append((toks, 150 if s == 0 else [9, 17, 33, 2][s % 4]))
  run_pair(cfg, w, prompts, steps=3, check_logits=False)


@pytest.mark.parametrize("nprompts,hi", [(6, 32), (24, 12)])
def test_gemma3_packed_prefill(nprompts, hi):
  packed_prefill_pair(nprompts, hi, model_name="gemma3-tiny")


def test_gemma3_head_dim_256_tiny():
  """head_dim 256 (gemma3-4b's): two RoPE pairs per lane, 32 chunks in the head norm's sum of squares."""
  cfg = _cfg(head_dim=256, per_device_batch_size=3.0, pagedattn_num_pages=96)
  w = H.synth.synth_weights(cfg, 8)
  prompts = [(_prompt(cfg.vocab_size, 48, 60 + s), [45, 8, 26][s]) for s in range(3)]
  run_pair(cfg, w, prompts, steps=30)


def test_gemma3_4b_widths():
  """The widths of configs/models/gemma3-4b.yml (emb 2560, 8 x 256 query heads over 4 kv heads, mlp 10240, vocab 262144, tied head, RoPE bases
  10 000 / 1 000 000) with 6 of its 34 layers -- five local and the global one: every projection shape and the 262 144-column unembedding of
  the transposed table against the oracle."""
  cfg = H.make_config("gemma3-4b", base_num_decoder_layers=6, max_prefill_predict_length=32, max_target_length=64, per_device_batch_size=2.0,
                      pagedattn_num_pages=40, sliding_window_size=20)
  w = H.synth.synth_weights(cfg, 10)
  prompts = [(_prompt(cfg.vocab_size, 32, 80 + s), [27, 6][s]) for s in range(2)]
  run_pair(cfg, w, prompts, steps=5)
