"""GPU parity of the tiled prefill attention (k_attn_tile: consecutive rows of a page group share their K / V loads) -- against the oracle
through MaxEngine with the tile forced on every prefill call, and against the per-row kernel at key ranges the oracle would take minutes
for (the narrow 8-row tile past ~1500 keys).  Bar: token ids, f32 logits and K/V bit-exact."""
import numpy as np
import pytest

import helpers as H
from test_gpu_engine import run_pair, _packed_inputs

pytestmark = pytest.mark.gpu

TILE_ALWAYS = {"attn_tile": 1, "attn_tile_min_rows": 1}


def test_tile_chunked_prefill_with_prefix():
  """Chunks that start inside a page and attend to a prefix written by earlier chunks; 7- and 4-row chunks give partly idle tiles."""
  cfg = H.make_config(use_chunked_prefill=True, prefill_chunk_size=16, max_prefill_predict_length=32, max_target_length=160,
                      pagedattn_num_pages=64)
  w = H.synth.synth_weights(cfg, 3)
  toks = np.random.default_rng(0).integers(1, cfg.vocab_size, size=75).astype(np.int32)
  a = run_pair(cfg, w, [(toks, 75)], steps=8, chunks=[[32, 32, 11]], options=TILE_ALWAYS)
  b = run_pair(cfg, w, [(toks, 75)], steps=8, chunks=[[7, 4, 32, 32]], options=TILE_ALWAYS)
  assert np.array_equal(a, b)


@pytest.mark.parametrize("hd,nq,nkv,page", [(128, 2, 2, 32), (64, 8, 2, 16), (128, 8, 2, 16), (64, 4, 2, 8), (128, 4, 4, 64)])
def test_tile_head_geometries(hd, nq, nkv, page):
  """MHA / 2 / 4 query heads per kv head, head_dim 64 and 128, pages of 8..64 tokens; prompts of 64, 33 and 5 rows (one tile holds rows of
  one prompt only here; the packed test below crosses prompts)."""
  cfg = H.make_config(base_emb_dim=256, base_num_query_heads=nq, base_num_kv_heads=nkv, head_dim=hd, base_mlp_dim=384,
                      base_num_decoder_layers=2, vocab_size=1024, model_name="default", per_device_batch_size=3.0,
                      max_prefill_predict_length=64, max_target_length=96, pagedattn_tokens_per_page=page, pagedattn_num_pages=3 * -(-96 // page) + 2)
  w = H.synth.synth_weights(cfg, 11)
  prompts = [H.synth.synth_prompt(cfg.vocab_size, 64, 60 + s) for s in range(3)]
  prompts[1] = (prompts[1][0], 33)
  prompts[2] = (prompts[2][0], 5)
  run_pair(cfg, w, prompts, steps=6, options=TILE_ALWAYS)


@pytest.mark.parametrize("nprompts,hi", [(9, 40), (30, 7)])
def test_tile_packed_prompts_share_tiles(nprompts, hi):
  """prefill_concat with prompts shorter than a tile: up to 16 page groups inside one 16-row tile, worked off group by group."""
  import torch
  rng = np.random.default_rng(100 + nprompts)
  lens = rng.integers(1, hi + 1, size=nprompts)
  B = nprompts
  cfg = H.make_config(per_device_batch_size=float(B), pagedattn_num_pages=2 * B * 16 + 1, ixt_staging_groups=nprompts, ixt_max_packed_rows=int(lens.sum()),
                      max_prefill_predict_length=64, max_target_length=96)
  w = H.synth.synth_weights(cfg, 9)
  eng, params = H.make_engine(cfg, w)
  for name, value in TILE_ALWAYS.items():
    H.lib.check(eng.lib.ixt_set_option(eng._ctx, name.encode(), value))
  orc = H.make_oracle(cfg, w)
  prompts = [(H.synth.synth_prompt(cfg.vocab_size, 64, 300 + s)[0], int(lens[s])) for s in range(nprompts)]
  toks, pos, seg, starts, ln = _packed_inputs(prompts)
  cache, pres, results = eng.prefill_concat(params=params, padded_tokens=toks, decoder_positions=pos, decoder_segment_ids=seg, start_pos=starts,
                                            true_lengths=ln, num_prompts=nprompts)
  slots = list(range(nprompts))
  opres, ofirst, ologits = orc.prefill_many(prompts, slots=slots, want_logits=True)
  assert np.array_equal(pres["tokens"].reshape(-1).cpu().numpy(), ofirst)
  assert np.array_equal(pres["logits"].reshape(nprompts, -1).cpu().numpy().view(np.uint32), ologits.view(np.uint32))
  del eng
  torch.cuda.synchronize()


def _prefill_long(cfg, w, toks, chunk, options):
  import torch
  maxengine = H.importlib.import_module("maxtext-indextts2_amd.maxengine")
  eng, params = H.make_engine(cfg, w)
  for name, value in options.items():
    H.lib.check(eng.lib.ixt_set_option(eng._ctx, name.encode(), value))
  done, pre, firsts = 0, None, []
  while done < len(toks):
    n = min(chunk, len(toks) - done)
    buf = np.zeros((cfg.max_prefill_predict_length,), np.int32)
    buf[:n] = toks[done:done + n]
    ep = None if done == 0 else maxengine.ExistingPrefix(cache=pre["cache"], common_prefix_tokens=toks[:done])
    pre, res = eng.prefill(params=params, padded_tokens=buf, true_length=n, slot=0, existing_prefix=ep)
    firsts.append(res.data.cpu().numpy().copy())
    done += n
  logits = pre["logits"].cpu().numpy().reshape(-1).view(np.uint32).copy()
  kv = [tuple(H.bits(t).copy() for t in eng.gather_kv(layer, 0, len(toks))) for layer in range(cfg.num_decoder_layers)]
  import ctypes
  out = (ctypes.c_int32 * 4)()
  H.lib.check(eng.lib.ixt_pages_status(eng._ctx, out, 0, None))
  status = list(out)
  del eng
  torch.cuda.synchronize()
  return np.concatenate(firsts), logits, kv, status


@pytest.mark.parametrize("hd,nq,nkv", [(128, 2, 2), (64, 4, 1)])
def test_tile_long_context_equals_row_kernel(hd, nq, nkv):
  """2300 keys: the wide tile (16 query vectors) up to ~1900 keys, the narrow one (8) beyond; every chunk's first token, the last logits
  and the whole K/V equal what the one-block-per-row kernel produces (which the oracle pins at the shorter ranges above)."""
  cfg = H.make_config(base_emb_dim=256, base_num_query_heads=nq, base_num_kv_heads=nkv, head_dim=hd, base_mlp_dim=384, base_num_decoder_layers=2,
                      vocab_size=1024, model_name="default", per_device_batch_size=1.0, use_chunked_prefill=True, prefill_chunk_size=128,
                      max_prefill_predict_length=128, max_target_length=2432, pagedattn_tokens_per_page=32, pagedattn_num_pages=80)
  w = H.synth.synth_weights(cfg, 21)
  toks = np.random.default_rng(5).integers(1, cfg.vocab_size, size=2300).astype(np.int32)
  a = _prefill_long(cfg, w, toks, 128, {"attn_tile": 1, "attn_tile_min_rows": 1})
  b = _prefill_long(cfg, w, toks, 128, {"attn_tile": 0})
  assert np.array_equal(a[0], b[0])
  assert np.array_equal(a[1], b[1])
  for (ka, va), (kb, vb) in zip(a[2], b[2]):
    assert np.array_equal(ka, kb) and np.array_equal(va, vb)
  assert a[3] == [0, 0, 0, 0] and b[3] == [0, 0, 0, 0]


def test_rows_beyond_declared_max_len_are_idle_and_counted():
  """ixt_prefill_rows(max_len): a row at or beyond the caller's bound is not computed against score rows sized for that bound; it is left
  idle and shows in ixt_pages_status()[3]."""
  import ctypes
  import torch
  cfg = H.make_config(per_device_batch_size=2.0, ixt_staging_groups=2, ixt_max_packed_rows=64, max_prefill_predict_length=64, max_target_length=96)
  w = H.synth.synth_weights(cfg, 9)
  eng, params = H.make_engine(cfg, w)
  eng.page_state = eng.page_manager.update_prefill_pages(page_state=eng.page_state, page_group_id=2, true_length=40, _staging=True)
  dev = eng.device
  tok = torch.arange(1, 41, dtype=torch.int32, device=dev)
  grp = torch.full((40,), 2, dtype=torch.int32, device=dev)
  pos = torch.arange(40, dtype=torch.int32, device=dev)
  ends = torch.tensor([39], dtype=torch.int32, device=dev)
  first = torch.empty((1,), dtype=torch.int32, device=dev)
  vp = lambda t: ctypes.c_void_p(t.data_ptr())
  H.lib.check(eng.lib.ixt_prefill_rows(eng._ctx, 40, vp(tok), vp(grp), vp(pos), 32, 1, vp(ends), vp(first), None, None))
  out = (ctypes.c_int32 * 4)()
  H.lib.check(eng.lib.ixt_pages_status(eng._ctx, out, 1, None))
  assert list(out) == [0, 0, 0, 8]
  H.lib.check(eng.lib.ixt_prefill_rows(eng._ctx, 40, vp(tok), vp(grp), vp(pos), 40, 1, vp(ends), vp(first), None, None))
  H.lib.check(eng.lib.ixt_pages_status(eng._ctx, out, 0, None))
  assert list(out) == [0, 0, 0, 0]
  del eng
  torch.cuda.synchronize()
