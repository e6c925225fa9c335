"""GPU parity, engine level: MaxEngine (HIP) against OracleEngine (CPU) through the reference's call order
(decode.py:133-184): prefill -> init_decode_state -> insert -> generate loop.  Bar: token ids, logits (f32 bit
patterns), K/V cache contents and PageState are all bit-exact."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

PAGE_FIELDS = ("page_status", "page_map", "num_pages_used", "sequence_lengths", "active_page", "has_active_page", "active_page_position")


def assert_pages_equal(eng, orc):
  st = eng.page_manager.device_state().to_numpy()
  for f in PAGE_FIELDS:
    assert np.array_equal(st[f], getattr(orc.page_state, f)), f


def run_pair(cfg, weights, prompts, steps, use_graph=True, check_logits=True, chunks=None, options=None, sampling_ranges=None):
  """prompts: list of (padded_tokens, true_length) one per slot (len <= batch).  Returns HIP token matrix."""
  import torch
  eng, params = H.make_engine(cfg, weights)
  for name, value in (options or {}).items():
    H.lib.check(eng.lib.ixt_set_option(eng._ctx, name.encode(), int(value)))
  orc = H.make_oracle(cfg, weights)
  if sampling_ranges is not None:
    eng.set_sampling_ranges(sampling_ranges)
    orc.set_sampling_ranges(sampling_ranges)
  try:
    return _run_pair(cfg, eng, params, orc, prompts, steps, check_logits, chunks)
  finally:
    orc.set_sampling_ranges([])


def _run_pair(cfg, eng, params, orc, prompts, steps, check_logits, chunks):
  import torch
  B = int(cfg.per_device_batch_size)
  prefixes, oprefixes = [], []
  for slot, (toks, tl) in enumerate(prompts):
    if chunks is None:
      pre, res = eng.prefill(params=params, padded_tokens=toks, true_length=tl, slot=slot)
      opre, ores = orc.prefill(toks, tl, slot=slot, want_logits=True)
    else:
      maxengine = H.importlib.import_module("maxtext-indextts2_amd.maxengine")
      done = 0
      pre = opre = None
      for n in chunks[slot]:
        chunk = np.zeros((cfg.max_prefill_predict_length,), np.int32)
        chunk[:n] = toks[done:done + n]
        ep = None if done == 0 else maxengine.ExistingPrefix(cache=pre["cache"], common_prefix_tokens=toks[:done])
        pre, res = eng.prefill(params=params, padded_tokens=chunk, true_length=n, slot=slot, existing_prefix=ep)
        opre, ores = orc.prefill(chunk, n, slot=slot, existing_prefix_tokens=None if done == 0 else toks[:done], want_logits=True)
        done += n
    assert np.array_equal(res.data.cpu().numpy(), ores), (slot, res.data.cpu().numpy(), ores)
    assert np.array_equal(pre["logits"].cpu().numpy().reshape(-1).view(np.uint32), opre["logits"].reshape(-1).view(np.uint32))
    assert int(pre["next_pos"][0, 0]) == int(opre["next_pos"][0, 0])
    prefixes.append(pre)
    oprefixes.append(opre)
  assert_pages_equal(eng, orc)
  # K/V parity after prefill, every layer
  for slot, (toks, tl) in enumerate(prompts):
    n = int(prefixes[slot]["next_pos"][0, 0])
    for layer in range(H.sublayers(cfg)):
      kg, vg = eng.gather_kv(layer, slot, n)
      ko, vo = orc.model.gather_kv(layer, orc.page_state.page_map[slot], n)
      assert np.array_equal(H.bits(kg), ko), (slot, layer)
      assert np.array_equal(H.bits(vg), vo), (slot, layer)
  ds = eng.init_decode_state(materialize_logits=check_logits)
  ods = orc.init_decode_state()
  for slot in range(len(prompts)):
    ds = eng.insert(prefixes[slot], ds, slot)
    ods = orc.insert(oprefixes[slot], ods, slot)
  out = []
  for i in range(steps):
    ds, res = eng.generate(params, ds)
    ods, ores = orc.generate(ods, want_logits=check_logits)
    got = res.data.cpu().numpy()
    act = orc.page_state.has_active_page
    assert np.array_equal(got[act], ores[act]), (i, got, ores)
    if check_logits:
      gl = ds["logits"].cpu().numpy().reshape(B, -1)
      assert np.array_equal(gl[act].view(np.uint32), ods["logits"][act].view(np.uint32)), i
    out.append(got[:, 0].copy())
  assert_pages_equal(eng, orc)
  assert np.array_equal(ds["next_pos"].cpu().numpy(), ods["next_pos"])
  assert np.array_equal(ds["generated_tokens"].cpu().numpy(), ods["generated_tokens"])
  for slot, _ in enumerate(prompts):
    n = int(ods["next_pos"][slot, 0])
    kg, vg = eng.gather_kv(H.sublayers(cfg) - 1, slot, n)
    ko, vo = orc.model.gather_kv(H.sublayers(cfg) - 1, orc.page_state.page_map[slot], n)
    assert np.array_equal(H.bits(kg), ko) and np.array_equal(H.bits(vg), vo)
  assert eng.exchange_status() == 0      # no bounded wait of the split decode attention gave up
  del eng
  torch.cuda.synchronize()
  return np.stack(out, 0)


def test_tiny_single_slot():
  cfg = H.make_config()
  w = H.synth.synth_weights(cfg, 0)
  toks, tl = H.synth.synth_prompt(cfg.vocab_size, cfg.max_prefill_predict_length, 1234)
  run_pair(cfg, w, [(toks, tl)], steps=40)


@pytest.mark.parametrize("B,nprompts", [(2, 2), (4, 3), (8, 8), (12, 9), (20, 13), (40, 33), (72, 70), (136, 131), (1024, 1001)])
def test_tiny_batched_ragged(B, nprompts):
  """Several slots with different prompt lengths (ragged), fewer prompts than slots (inactive slots); 1024 slots = the engine's maximum and
  the bench's default batch (row descriptors, page advance for 1024 slots out of one 10 000-page pool, 1024-row attention grid)."""
  cfg = H.make_config(per_device_batch_size=float(B), pagedattn_num_pages=max(160, B * 10 + 1))
  w = H.synth.synth_weights(cfg, 1)
  prompts = []
  for s in range(nprompts):
    toks, _ = H.synth.synth_prompt(cfg.vocab_size, cfg.max_prefill_predict_length, 100 + s)
    prompts.append((toks, [32, 1, 9, 17, 8, 31, 16, 24][s % 8]))
  run_pair(cfg, w, prompts, steps=40 if B <= 8 else (12 if B <= 40 else 5), check_logits=B <= 20 or 128 < B < 512)


@pytest.mark.parametrize("wave_pairs", [1, 0])
@pytest.mark.parametrize("B,nprompts", [(1, 1), (4, 3), (20, 13)])
def test_tiny_one_launch_decode_attention(B, nprompts, wave_pairs):
  """The decode attention as ONE launch per layer (the default from rows x kv_heads >= 128) forced onto small batches, in both
  forms -- one wave per (row, kv head) without block barriers, and one block per pair: ragged lengths, inactive slots, 2 query
  heads per kv head, page-boundary crossings during the 40 steps."""
  cfg = H.make_config(per_device_batch_size=float(B), pagedattn_num_pages=max(160, B * 10 + 1))
  w = H.synth.synth_weights(cfg, 5)
  prompts = []
  for s in range(nprompts):
    toks, _ = H.synth.synth_prompt(cfg.vocab_size, cfg.max_prefill_predict_length, 300 + s)
    prompts.append((toks, [32, 1, 9, 17, 8, 31, 16, 24][s % 8]))
  run_pair(cfg, w, prompts, steps=40, options={"attn_fused_min_blocks": 1, "attn_wave_pairs": wave_pairs})


def test_audio_only_sampling_ranges():
  """greedy sampling restricted to column ranges (the audio rows of the expanded vocabulary): tokens equal the
  oracle's restricted argmax, all lie inside the ranges, and the returned logits are the unmasked ones (checked by run_pair)."""
  cfg = H.make_config(per_device_batch_size=3.0)
  w = H.synth.synth_weights(cfg, 6)
  prompts = [H.synth.synth_prompt(cfg.vocab_size, 32, 40 + s) for s in range(3)]
  ranges = [(6, 105), (600, 900)]
  got = run_pair(cfg, w, prompts, steps=24, sampling_ranges=ranges)
  assert all(any(lo <= t < hi for lo, hi in ranges) for t in got.reshape(-1))
  free = run_pair(cfg, w, prompts, steps=24)
  assert not np.array_equal(free, got)


def test_tiny_graph_equals_eager():
  cfg_g = H.make_config(per_device_batch_size=2.0, ixt_use_hip_graph=True)
  cfg_e = H.make_config(per_device_batch_size=2.0, ixt_use_hip_graph=False)
  w = H.synth.synth_weights(cfg_g, 2)
  prompts = [H.synth.synth_prompt(cfg_g.vocab_size, 32, 7), H.synth.synth_prompt(cfg_g.vocab_size, 32, 8)]
  a = run_pair(cfg_g, w, prompts, steps=24)
  b = run_pair(cfg_e, w, prompts, steps=24)
  assert np.array_equal(a, b)


def test_tiny_chunked_prefill():
  """Chunked == unchunked exactly (tests/maxengine_test.py:166-300 property), over pages."""
  cfg = H.make_config(use_chunked_prefill=True, prefill_chunk_size=16, max_prefill_predict_length=32, max_target_length=160,
                      pagedattn_num_pages=64)
  w = H.synth.synth_weights(cfg, 3)
  rng = np.random.default_rng(0)
  toks = rng.integers(1, cfg.vocab_size, size=75).astype(np.int32)
  import torch
  eng, params = H.make_engine(cfg, w)
  a = run_pair(cfg, w, [(toks, 75)], steps=20, chunks=[[32, 32, 11]])
  b = run_pair(cfg, w, [(toks, 75)], steps=20, chunks=[[7, 4, 32, 32]])
  assert np.array_equal(a, b)
  del eng


@pytest.mark.parametrize("fused", [0, 1, 2])
def test_head_dim_128_mha_page32(fused):
  """C2's attention geometry (head_dim 128, MHA, 32-token pages) on a small body; key ranges of up to 9 pages = 3 super-blocks,
  through the split kernels and through the one-launch decode attention."""
  cfg = H.make_config(base_emb_dim=256, base_num_query_heads=2, base_num_kv_heads=2, head_dim=128, base_mlp_dim=384,
                      base_num_decoder_layers=2, vocab_size=2048, model_name="default", per_device_batch_size=2.0,
                      max_prefill_predict_length=64, max_target_length=320, pagedattn_tokens_per_page=32, pagedattn_num_pages=24)
  w = H.synth.synth_weights(cfg, 4)
  p0 = H.synth.synth_prompt(cfg.vocab_size, 64, 1)
  p1 = H.synth.synth_prompt(cfg.vocab_size, 64, 2)
  run_pair(cfg, w, [p0, (p1[0], 33)], steps=200, check_logits=False, options={"attn_fused_min_blocks": 1, "attn_wave_pairs": fused - 1} if fused else None)


@pytest.mark.parametrize("split", [0, 102, 103, 108])   # 100 + S: S blocks even where the one-block form would be taken
@pytest.mark.parametrize("geom", ["mha128", "gqa2_64", "gqa4_128"])
def test_split_decode_attention(split, geom):
  """Few decode rows: the keys of a (row, kv head) over `split` blocks of one launch that exchange maxima, page sums and super-block sums through
  HBM (k_attn_split; 0: k_attn_small, one block per pair).  200 steps from 64- and 33-token prompts with 32-token pages: 3 to 9 pages = 1 to 3
  super-blocks, so blocks without keys, blocks with a partial last page and page-boundary crossings all occur; bit-exact against the oracle."""
  heads = {"mha128": (2, 2, 128), "gqa2_64": (4, 2, 64), "gqa4_128": (8, 2, 128)}[geom]
  cfg = H.make_config(base_emb_dim=256, base_num_query_heads=heads[0], base_num_kv_heads=heads[1], head_dim=heads[2], base_mlp_dim=384,
                      base_num_decoder_layers=2, vocab_size=2048, model_name="default", per_device_batch_size=2.0,
                      max_prefill_predict_length=64, max_target_length=320, pagedattn_tokens_per_page=32, pagedattn_num_pages=24)
  w = H.synth.synth_weights(cfg, 4)
  p0 = H.synth.synth_prompt(cfg.vocab_size, 64, 1)
  p1 = H.synth.synth_prompt(cfg.vocab_size, 64, 2)
  run_pair(cfg, w, [p0, (p1[0], 33)], steps=200, check_logits=False, options={"small_attn_split": split})


@pytest.mark.parametrize("hd", [64, 128])
@pytest.mark.parametrize("fused", [0, 1, 2])
def test_four_query_heads_per_kv_head(fused, hd):
  """GQA with 4 query heads per kv head (8 query / 2 kv heads of 64 or 128): split kernels, one-launch block form, one-launch wave form."""
  cfg = H.make_config(base_emb_dim=256, base_num_query_heads=8, base_num_kv_heads=2, head_dim=hd, base_mlp_dim=384,
                      base_num_decoder_layers=2, vocab_size=1024, model_name="default", per_device_batch_size=3.0,
                      max_prefill_predict_length=32, max_target_length=96, pagedattn_tokens_per_page=16, pagedattn_num_pages=32)
  w = H.synth.synth_weights(cfg, 11)
  prompts = [H.synth.synth_prompt(cfg.vocab_size, 32, 60 + s) for s in range(3)]
  prompts[1] = (prompts[1][0], 5)
  run_pair(cfg, w, prompts, steps=40, options={"attn_fused_min_blocks": 1, "attn_wave_pairs": fused - 1} if fused else None)


def test_release_and_reuse_slot():
  cfg = H.make_config(per_device_batch_size=2.0)
  w = H.synth.synth_weights(cfg, 5)
  eng, params = H.make_engine(cfg, w)
  orc = H.make_oracle(cfg, w)
  for rep in range(3):
    for slot in range(2):
      toks, tl = H.synth.synth_prompt(cfg.vocab_size, 32, 10 * rep + slot)
      pre, res = eng.prefill(params=params, padded_tokens=toks, true_length=tl, slot=slot)
      opre, ores = orc.prefill(toks, tl, slot=slot)
      assert np.array_equal(res.data.cpu().numpy(), ores)
      if rep == 0 and slot == 0:
        ds, ods = eng.init_decode_state(), orc.init_decode_state()
      ds = eng.insert(pre, ds, slot)
      ods = orc.insert(opre, ods, slot)
    for _ in range(10):
      ds, res = eng.generate(params, ds)
      ods, ores = orc.generate(ods)
      assert np.array_equal(res.data.cpu().numpy(), ores)
    assert_pages_equal(eng, orc)
    eng.release_pages(rep % 2)
    orc.release_pages(rep % 2)
    assert_pages_equal(eng, orc)


def test_errors_match_reference_conventions():
  cfg = H.make_config()
  w = H.synth.synth_weights(cfg, 0)
  eng, params = H.make_engine(cfg, w)
  toks, tl = H.synth.synth_prompt(cfg.vocab_size, 32, 0)
  maxengine = H.importlib.import_module("maxtext-indextts2_amd.maxengine")
  with pytest.raises(ValueError):  # maxengine.py:435-436
    eng.prefill(params=params, padded_tokens=toks, true_length=tl, slot=0, existing_prefix=maxengine.ExistingPrefix(None, toks[:4]))
  with pytest.raises(ValueError):  # page_manager.py:529-530
    eng.page_manager.update_prefill_pages(None, 5, 4)
  with pytest.raises(ValueError):  # page_manager.py:531-532
    eng.page_manager.update_prefill_pages(None, 0, cfg.max_target_length + 1)
  with pytest.raises(ValueError):
    eng.page_manager.release_pages(None, -1)
  with pytest.raises(NotImplementedError):
    _ = eng.colocated_cpus
  eng2 = maxengine.MaxEngine(cfg)
  with pytest.raises(Exception):  # weights never loaded -> loud failure, no fallback
    eng2.prefill(params=None, padded_tokens=toks, true_length=tl, slot=0)


# ---------------------------------------------------------------------------------------------------------------------------------
# packed prefill (prefill_concat / insert_partial), insert into other slots, and the guards of the decode rows
# ---------------------------------------------------------------------------------------------------------------------------------
STATE_FIELDS = ("num_pages_used", "sequence_lengths", "has_active_page", "active_page_position")


def _packed_inputs(prompts):
  lens = np.array([tl for _, tl in prompts], np.int32)
  toks = np.concatenate([np.asarray(t, np.int32)[:tl] for t, tl in prompts])
  pos = np.concatenate([np.arange(tl) for tl in lens]).astype(np.int32)
  seg = np.concatenate([np.full(tl, i + 1) for i, tl in enumerate(lens)]).astype(np.int32)
  starts = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int32)
  return toks, pos, seg, starts, lens


@pytest.mark.parametrize("nprompts,hi", [(6, 32), (10, 32), (24, 12)])
def test_tiny_packed_prefill(nprompts, hi):
  """prefill_concat (maxengine.py:719-836) + insert_partial (:1206-1318) over pages: several prompts in ONE pass over the weights, their
  pages handed to permuted slots.  98 rows (split-K dots), 211 rows and 150 rows of 24 prompts (LDS-ring GEMMs with every fused epilogue).
  First tokens, f32 logits, the K/V of every layer, the per-slot page-state fields and the following decode steps equal the oracle's."""
  packed_prefill_pair(nprompts, hi)


def packed_prefill_pair(nprompts, hi, model_name="ixt-tiny", **cfg_kw):
  import torch
  B = nprompts + 2
  rng = np.random.default_rng(nprompts)
  lens = rng.integers(1, hi + 1, size=nprompts)
  lens[0] = hi
  cfg = H.make_config(model_name, per_device_batch_size=float(B), pagedattn_num_pages=2 * B * 16 + 1, ixt_staging_groups=nprompts,
                      ixt_max_packed_rows=int(lens.sum()), **cfg_kw)
  w = H.synth.synth_weights(cfg, 9)
  eng, params = H.make_engine(cfg, w)
  orc = H.make_oracle(cfg, w)
  prompts = [(H.synth.synth_prompt(cfg.vocab_size, 32, 700 + s)[0], int(lens[s])) for s in range(nprompts)]
  slots = list(rng.permutation(B)[:nprompts])
  toks, pos, seg, starts, ln = _packed_inputs(prompts)
  cache, pres, results = eng.prefill_concat(params=params, padded_tokens=toks, decoder_positions=pos, decoder_segment_ids=seg, start_pos=starts,
                                            true_lengths=ln, num_prompts=nprompts)
  opres, ofirst, ologits = orc.prefill_many(prompts, slots=slots, want_logits=True)
  assert np.array_equal(pres["tokens"].reshape(-1).cpu().numpy(), ofirst)
  assert np.array_equal(np.stack([r.data.cpu().numpy()[0] for r in results]), np.stack([ofirst, np.ones_like(ofirst), np.zeros_like(ofirst)], 1))
  assert np.array_equal(pres["logits"].reshape(nprompts, -1).cpu().numpy().view(np.uint32), ologits.view(np.uint32))
  ds = eng.insert_partial(pres, eng.init_decode_state(materialize_logits=True), cache, np.array(slots), start_indices=starts, num_prompts=nprompts, seq_len=32)
  ods = orc.init_decode_state()
  for i, s in enumerate(slots):
    ods = orc.insert(opres[i], ods, s)
  st = eng.page_manager.device_state().to_numpy()
  for f in STATE_FIELDS:
    assert np.array_equal(st[f], getattr(orc.page_state, f)), f
  assert int(st["page_status"].sum()) == int(orc.page_state.page_status.sum())     # the staging groups kept nothing
  for i, s in enumerate(slots):
    for layer in range(H.sublayers(cfg)):
      kg, vg = eng.gather_kv(layer, s, int(lens[i]))
      ko, vo = orc.model.gather_kv(layer, orc.page_state.page_map[s], int(lens[i]))
      assert np.array_equal(H.bits(kg), ko) and np.array_equal(H.bits(vg), vo), (s, layer)
  for step in range(6):
    ds, res = eng.generate(params, ds)
    ods, ores = orc.generate(ods, want_logits=True)
    act = orc.page_state.has_active_page
    assert np.array_equal(res.data.cpu().numpy()[act], ores[act]), step
    assert np.array_equal(ds["logits"].cpu().numpy().reshape(B, -1)[act].view(np.uint32), ods["logits"][act].view(np.uint32)), step
  del eng
  torch.cuda.synchronize()


def test_insert_into_other_slots_copies_pages():
  """insert / bulk_insert of one prefix into slots other than the one it was prefilled into (maxengine.py:946-1043, 1106-1125): every
  target slot gets its own copy of the pages and then decodes exactly like the source."""
  cfg = H.make_config(per_device_batch_size=4.0, pagedattn_num_pages=64)
  w = H.synth.synth_weights(cfg, 12)
  eng, params = H.make_engine(cfg, w)
  orc = H.make_oracle(cfg, w)
  toks, tl = H.synth.synth_prompt(cfg.vocab_size, 32, 77)
  pre, res = eng.prefill(params=params, padded_tokens=toks, true_length=tl, slot=1)
  ds = eng.bulk_insert(pre, eng.init_decode_state(), [1, 3, 0])
  ods = orc.init_decode_state()
  for s in (1, 3, 0):
    opre, ores = orc.prefill(toks, tl, slot=s)
    ods = orc.insert(opre, ods, s)
  assert np.array_equal(res.data.cpu().numpy(), ores)
  for step in range(20):
    ds, r = eng.generate(params, ds)
    ods, orr = orc.generate(ods)
    got = r.data.cpu().numpy()
    assert np.array_equal(got[[0, 1, 3]], orr[[0, 1, 3]]), step
    assert np.array_equal(got[0], got[1]) and np.array_equal(got[0], got[3])
  for s in (0, 1, 3):
    n = int(ods["next_pos"][s, 0])
    kg, vg = eng.gather_kv(1, s, n)
    ko, vo = orc.model.gather_kv(1, orc.page_state.page_map[s], n)
    assert np.array_equal(H.bits(kg), ko) and np.array_equal(H.bits(vg), vo)


def test_generate_between_prefill_and_insert():
  """A generate() between prefill(slot) and insert(slot) -- the usual interleaving when one engine serves prefill and decode -- must not
  touch the prefilled slot: its decode row idles (stale next_pos of the previous occupant), its K/V stay intact, and after insert it
  decodes exactly like an engine that never saw the stray steps."""
  cfg = H.make_config(per_device_batch_size=2.0, max_target_length=64, pagedattn_num_pages=32)
  w = H.synth.synth_weights(cfg, 13)
  eng, params = H.make_engine(cfg, w)
  orc = H.make_oracle(cfg, w)
  t0, l0 = H.synth.synth_prompt(cfg.vocab_size, 32, 1)
  t1, l1 = H.synth.synth_prompt(cfg.vocab_size, 32, 2)
  # slot 1 decodes for a while so that it leaves a large stale next_pos behind, then is released
  p1, _ = eng.prefill(params=params, padded_tokens=t1, true_length=l1, slot=1)
  ds = eng.insert(p1, eng.init_decode_state(), 1)
  for _ in range(20):
    ds, _ = eng.generate(params, ds)
  eng.release_pages(1)
  # slot 0 active; slot 1 prefilled again (short prompt) but NOT inserted; generate runs in between
  p0, _ = eng.prefill(params=params, padded_tokens=t0, true_length=l0, slot=0)
  ds = eng.insert(p0, ds, 0)
  op0, _ = orc.prefill(t0, l0, slot=0)
  ods = orc.insert(op0, orc.init_decode_state(), 0)
  p1b, _ = eng.prefill(params=params, padded_tokens=t1, true_length=5, slot=1)
  op1b, _ = orc.prefill(t1, 5, slot=1)
  k_before = H.bits(eng.gather_kv(0, 1, 5)[0]).copy()
  for _ in range(3):
    ds, r = eng.generate(params, ds)
    orc.page_state.has_active_page[1] = False          # the oracle's view: slot 1 does not decode before its insert
    ods, orr = orc.generate(ods)
    orc.page_state.has_active_page[1] = True
    assert np.array_equal(r.data.cpu().numpy()[0], orr[0])
  assert np.array_equal(H.bits(eng.gather_kv(0, 1, 5)[0]), k_before)
  ds = eng.insert(p1b, ds, 1)
  orc.page_state = orc.pm.update_prefill_pages(orc.page_state, 1, 5)   # lengths as right after the prefill (what insert restores)
  ods = orc.insert(op1b, ods, 1)
  for step in range(10):
    ds, r = eng.generate(params, ds)
    ods, orr = orc.generate(ods)
    assert np.array_equal(r.data.cpu().numpy(), orr), step


def test_generate_past_max_target_length_is_harmless():
  """One generate() too many (the slot's sequence has reached max_target_length): the row idles, no K/V is written out of range, other
  slots are unaffected."""
  cfg = H.make_config(per_device_batch_size=2.0, max_prefill_predict_length=32, max_target_length=40, pagedattn_num_pages=16)
  w = H.synth.synth_weights(cfg, 14)
  eng, params = H.make_engine(cfg, w)
  orc = H.make_oracle(cfg, w)
  t0, _ = H.synth.synth_prompt(cfg.vocab_size, 32, 3)
  ds, ods = eng.init_decode_state(), orc.init_decode_state()
  for s, tl in ((0, 32), (1, 8)):
    p, _ = eng.prefill(params=params, padded_tokens=t0, true_length=tl, slot=s)
    op, _ = orc.prefill(t0, tl, slot=s)
    ds, ods = eng.insert(p, ds, s), orc.insert(op, ods, s)
  for step in range(12):                                # slot 0 is full after 8 steps
    ds, r = eng.generate(params, ds)
    ods, orr = orc.generate(ods) if step < 8 else (ods, None)
    if step < 8:
      assert np.array_equal(r.data.cpu().numpy(), orr), step
  k0, _ = eng.gather_kv(0, 0, 40)
  ko, _ = orc.model.gather_kv(0, orc.page_state.page_map[0], 40)
  assert np.array_equal(H.bits(k0), ko)                 # nothing was written after the sequence filled up


def test_page_pool_exhaustion_is_reported_and_harmless():
  """A pool too small for the reservation: the reference's page manager leaves the group empty without a word (page_manager.py:305-311).
  Here prefill raises, and with the check off the rows of the empty group are simply not computed: the neighbour's K/V and tokens stay
  exactly what they are without the failed request."""
  cfg = H.make_config(per_device_batch_size=2.0, pagedattn_num_pages=6, pagedattn_tokens_per_page=8)   # 5 usable pages of 8 tokens
  w = H.synth.synth_weights(cfg, 15)
  eng, params = H.make_engine(cfg, w)
  orc = H.make_oracle(cfg, w)
  t0, _ = H.synth.synth_prompt(cfg.vocab_size, 32, 4)
  p0, _ = eng.prefill(params=params, padded_tokens=t0, true_length=20, slot=0)     # 3 pages
  op0, _ = orc.prefill(t0, 20, slot=0)
  with pytest.raises(ValueError, match="page pool"):
    eng.prefill(params=params, padded_tokens=t0, true_length=30, slot=1)           # needs 4, 2 are free
  orc.page_state = orc.pm.update_prefill_pages(orc.page_state, 1, 30)              # the reference: silently unchanged
  assert_pages_equal(eng, orc)
  ds, ods = eng.insert(p0, eng.init_decode_state(), 0), orc.insert(op0, orc.init_decode_state(), 0)
  for step in range(30):                                                            # slot 0 grows to 50 tokens: needs 7 pages, gets 5
    ds, r = eng.generate(params, ds)
    ods, orr = orc.generate(ods)
    if step < 20:                                                                   # 40 tokens = 5 pages: identical while pages last
      assert np.array_equal(r.data.cpu().numpy()[0], orr[0]), step
  import ctypes
  out = (ctypes.c_int32 * 4)()
  H.lib.check(eng.lib.ixt_pages_status(eng._ctx, out, 0, None))
  assert out[0] == 1 and out[1] == 1 and out[2] >= 1                                # one failed reservation (group 1), starved decode rows
  k0, _ = eng.gather_kv(0, 0, 40)
  ko, _ = orc.model.gather_kv(0, orc.page_state.page_map[0], 40)
  assert np.array_equal(H.bits(k0), ko)


def test_prefill_multisampling_greedy():
  """maxengine.py:595-717 under greedy sampling: num_samples equal first tokens, the reference's result shapes, one cache."""
  import torch
  cfg = H.make_config(per_device_batch_size=2.0)
  w = H.synth.synth_weights(cfg, 5)
  eng, params = H.make_engine(cfg, w)
  orc = H.make_oracle(cfg, w)
  toks, tl = H.synth.synth_prompt(cfg.vocab_size, 32, 41)
  pre, res = eng.prefill_multisampling(params=params, padded_tokens=toks, true_length=tl, num_samples=3, slot=1)
  opre, ores = orc.prefill(toks, tl, slot=1, want_logits=True)
  data = res.data.cpu().numpy()
  assert data.shape == (3, 3) and res.samples_per_slot == 3
  assert np.array_equal(data, np.repeat(ores, 3, axis=0))
  assert pre["tokens"].shape == (3, 1) and pre["generated_tokens"].shape == (3, 1) and int(pre["next_pos"][0, 0]) == tl
  assert np.array_equal(pre["logits"].cpu().numpy().reshape(-1).view(np.uint32), opre["logits"].reshape(-1).view(np.uint32))
  with pytest.raises(ValueError):
    eng.prefill_multisampling(params=params, padded_tokens=toks, true_length=tl, num_samples=0)
  del eng
  torch.cuda.synchronize()


def test_engine_factory_helpers(tmp_path):
  """create_engine_from_config_flags (maxengine.py:1509-1545: positional batch / lengths + "k=v" string, load_parameters_path required) and
  set_engine_vars_from_base_engine (:1490-1506): the second engine decodes exactly like the first."""
  import torch
  maxengine = H.importlib.import_module("maxtext-indextts2_amd.maxengine")
  cfg = H.make_config(per_device_batch_size=2.0)
  w = H.synth.synth_weights(cfg, 5)
  path = str(tmp_path / "w.npz")
  np.savez(path, **w)
  with pytest.raises(AssertionError):
    maxengine.create_engine_from_config_flags(None, 2, 32, 96, "model_name=ixt-tiny")
  a = maxengine.create_engine_from_config_flags(None, 2, 32, 96, f"model_name=ixt-tiny load_parameters_path={path} pagedattn_num_pages=16")
  assert a.max_concurrent_decodes == 2 and a.max_prefill_length == 32
  pa = a.load_params()
  b = maxengine.MaxEngine(a.config)
  with pytest.raises(ValueError):
    maxengine.set_engine_vars_from_base_engine(b, maxengine.MaxEngine(a.config))
  maxengine.set_engine_vars_from_base_engine(b, a)
  toks, tl = H.synth.synth_prompt(a.config.vocab_size, 32, 7)
  outs = []
  for eng in (a, b):
    pre, res = eng.prefill(params=pa, padded_tokens=toks, true_length=tl, slot=0)
    ds = eng.insert(pre, eng.init_decode_state(), 0)
    seq = [int(res.data[0, 0])]
    for _ in range(8):
      ds, r = eng.generate(pa, ds)
      seq.append(int(r.data[0, 0]))
    outs.append(seq)
  assert outs[0] == outs[1]
  del a, b
  torch.cuda.synchronize()


def test_return_log_prob_outputs():
  """config.return_log_prob (maxengine.py:502-516, 801-815, 907-926): ResultTokens.log_prob = log_softmax(logits)[token] of the first token
  (prefill, prefill_concat, prefill_multisampling) and of every generated token, bit-exact against the oracle's restatement of
  inference_utils.log_prob_of_chosen_token over the same logits."""
  import torch
  from oracle import oracle as O
  cfg = H.make_config(per_device_batch_size=3.0, return_log_prob=True, ixt_staging_groups=2, ixt_max_packed_rows=64, pagedattn_num_pages=40)
  w = H.synth.synth_weights(cfg, 5)
  eng, params = H.make_engine(cfg, w)
  orc = H.make_oracle(cfg, w)
  toks, tl = H.synth.synth_prompt(cfg.vocab_size, 32, 77)
  pre, res = eng.prefill(params=params, padded_tokens=toks, true_length=tl, slot=0)
  opre, ores = orc.prefill(toks, tl, slot=0, want_logits=True)
  want = O.logprob_rows(opre["logits"].reshape(1, -1), ores[:, 0])
  assert res.log_prob.shape == (1, 1)
  assert np.array_equal(res.log_prob.cpu().numpy().reshape(-1).view(np.uint32), want.view(np.uint32))
  _, resm = eng.prefill_multisampling(params=params, padded_tokens=toks, true_length=tl, num_samples=2, slot=1)
  assert resm.log_prob.shape == (2, 1) and np.array_equal(resm.log_prob.cpu().numpy().reshape(-1).view(np.uint32), np.repeat(want, 2).view(np.uint32))
  ds = eng.insert(pre, eng.init_decode_state(), 0)
  ods = orc.insert(opre, orc.init_decode_state(), 0)
  for step in range(4):
    ds, r = eng.generate(params, ds)
    ods, orr = orc.generate(ods, want_logits=True)
    assert np.array_equal(r.data.cpu().numpy()[0], orr[0])
    want = O.logprob_rows(ods["logits"][0:1].reshape(1, -1), orr[0:1, 0])
    assert r.log_prob.shape == (3, 1)
    assert np.array_equal(r.log_prob.cpu().numpy()[0].view(np.uint32), want.view(np.uint32)), step
  from test_gpu_engine import _packed_inputs as _pi
  prompts = [(H.synth.synth_prompt(cfg.vocab_size, 32, 900 + s)[0], 9 + 5 * s) for s in range(2)]
  t2, pos, seg, starts, ln = _pi(prompts)
  _, pres, results = eng.prefill_concat(params=params, padded_tokens=t2, decoder_positions=pos, decoder_segment_ids=seg, start_pos=starts, true_lengths=ln,
                                        num_prompts=2)
  lg = pres["logits"].reshape(2, -1).cpu().numpy()
  want = O.logprob_rows(lg, pres["tokens"].reshape(2).cpu().numpy())
  got = np.concatenate([r.log_prob.cpu().numpy().reshape(-1) for r in results])
  assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
  # return_prompt_logp of the packed form (maxengine.py:777-778): every packed position's log-probability == the one-by-one prefill's == the oracle's
  _, pres2, _ = eng.prefill_concat(params=params, padded_tokens=t2, decoder_positions=pos, decoder_segment_ids=seg, start_pos=starts, true_lengths=ln,
                                   num_prompts=2, return_prompt_logp=True)
  plp = pres2["prompt_logp"].cpu().numpy()
  assert plp.shape == (1, len(t2))
  covered = np.zeros(len(t2), bool)
  for i, (ptoks, tl) in enumerate(prompts):
    orc2 = H.make_oracle(cfg, w)
    opre2, _ = orc2.prefill(ptoks, tl, slot=0, want_prompt_logp=True)
    st0 = int(starts[i])
    assert np.array_equal(plp[0, st0:st0 + tl].view(np.uint32), opre2["prompt_logp"].view(np.uint32)), i
    covered[st0:st0 + tl] = True
  assert not plp[0, ~covered].any()                                                       # padding positions hold 0
  del eng
  torch.cuda.synchronize()


@pytest.mark.parametrize("model,slots", [("ixt-tiny", 1), ("ixt-tiny", 7), ("gemma2-tiny", 2), ("mixtral-tiny", 3)])
def test_return_prompt_logp(model, slots):
  """prefill(return_prompt_logp=True) (maxengine.py:477-478, 529): log_softmax(logits[t])[tokens[t]] for every prompt position, bit-exact against
  the oracle (all rows' logits, inference_utils.log_prob_of_chosen_token restated); also chunk by chunk, where each chunk returns its own
  positions, and with the rows unembedded 1, 2, 3 or 7 at a time (the workspace holds max_slots rows)."""
  import torch
  cfg = H.make_config(model, per_device_batch_size=float(slots), max_prefill_predict_length=32, max_target_length=96, use_chunked_prefill=True,
                      prefill_chunk_size=32, pagedattn_num_pages=64)
  w = H.synth.synth_weights(cfg, 21)
  eng, params = H.make_engine(cfg, w)
  orc = H.make_oracle(cfg, w)
  toks = np.random.Generator(np.random.PCG64(5)).integers(1, cfg.vocab_size, size=64).astype(np.int32)
  first = np.zeros((32,), np.int32)
  first[:29] = toks[:29]
  pre, res = eng.prefill(params=params, padded_tokens=first, true_length=29, slot=0, return_prompt_logp=True)
  opre, ores = orc.prefill(first, 29, slot=0, want_logits=True, want_prompt_logp=True)
  got = pre["prompt_logp"].cpu().numpy()
  assert got.shape == (1, 32) and np.all(got[0, 29:] == 0.0)
  assert np.array_equal(got[0, :29].view(np.uint32), opre["prompt_logp"].view(np.uint32))
  assert np.array_equal(res.data.cpu().numpy(), ores) and np.array_equal(pre["logits"].cpu().numpy().reshape(-1).view(np.uint32), opre["logits"].reshape(-1).view(np.uint32))
  assert np.all(got[0, :29] < 0.0)
  # second chunk of the same sequence (positions 29 .. 52)
  maxengine = H.importlib.import_module("maxtext-indextts2_amd.maxengine")
  nxt = np.zeros((32,), np.int32)
  nxt[:24] = toks[29:53]
  ep = maxengine.ExistingPrefix(cache=pre["cache"], common_prefix_tokens=torch.from_numpy(toks[:29].copy()))
  pre2, _ = eng.prefill(params=params, padded_tokens=nxt, true_length=24, slot=0, existing_prefix=ep, return_prompt_logp=True)
  opre2, _ = orc.prefill(nxt, 24, slot=0, existing_prefix_tokens=toks[:29], want_prompt_logp=True)
  assert np.array_equal(pre2["prompt_logp"].cpu().numpy()[0, :24].view(np.uint32), opre2["prompt_logp"].view(np.uint32))
  del eng
  torch.cuda.synchronize()


def test_packed_prefill_pool_shortage_surfaces_before_the_pages_move():
  """prefill_concat checks its reservations without stopping the host (the status words travel to pinned memory in stream order); a reservation
  the pool could not serve raises from insert_partial, before any page list is handed to a slot, and the served prompts' results stand."""
  import torch
  cfg = H.make_config(per_device_batch_size=3.0, pagedattn_num_pages=6, pagedattn_tokens_per_page=8, ixt_staging_groups=2, ixt_max_packed_rows=64)
  w = H.synth.synth_weights(cfg, 15)
  eng, params = H.make_engine(cfg, w)
  orc = H.make_oracle(cfg, w)
  t0 = H.synth.synth_prompt(cfg.vocab_size, 32, 4)[0]
  t1 = H.synth.synth_prompt(cfg.vocab_size, 32, 5)[0]
  lens = np.array([20, 30], np.int32)                 # 3 pages + 4 pages, 5 usable
  starts = np.array([0, 20], np.int32)
  toks = np.concatenate([t0[:20], t1[:30]]).astype(np.int32)
  pos = np.concatenate([np.arange(20), np.arange(30)]).astype(np.int32)
  seg = np.concatenate([np.full(20, 1), np.full(30, 2)]).astype(np.int32)
  cache, pres, results = eng.prefill_concat(params=params, padded_tokens=toks, decoder_positions=pos, decoder_segment_ids=seg, start_pos=starts,
                                            true_lengths=lens, num_prompts=2)
  _, ores = orc.prefill(t0, 20, slot=0)
  assert int(results[0].data.cpu().numpy()[0, 0]) == int(ores[0, 0])          # the prompt that got its pages was computed
  with pytest.raises(ValueError, match="prefill_concat.*page pool"):
    eng.insert_partial(pres, eng.init_decode_state(), cache, [0, 1], num_prompts=2)
  ds = eng.insert_partial(pres, eng.init_decode_state(), cache, [0], num_prompts=1)   # the failure was reported once; the served prompt inserts
  ds, r = eng.generate(params, ds)
  torch.cuda.synchronize()
  assert int(r.data.cpu().numpy()[0, 1]) == 1
