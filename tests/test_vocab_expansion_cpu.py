"""SURVEY.md section 8(f) item 1 on the CPU: the audio-id mapping (pinned by the reference's own data file, in array form under
tests/golden/), the row-append tool, and the oracle's range-restricted greedy sampling."""
import importlib
import os

import numpy as np
import pytest

import helpers as H

VE = importlib.import_module("maxtext-indextts2_amd.vocab_expansion")


@pytest.fixture(scope="module")
def mapping():
  return VE.AudioTokenMapping.from_npz(os.path.join(H.GOLDEN, "audio_token_mapping_adjusted.npz"))


def test_mapping_table_equals_rule(mapping):
  """The data file of the reference (audio_token_mapping_adjusted.json -> tests/golden/*.npz by tests/golden/make_golden.py)
  against the generator's rule (extend_tokenizer.py:29-46): a < 99 -> 6 + a, else 256001 + (a - 99); 98 padding rows."""
  rule = VE.AudioTokenMapping.from_rule()
  for f in ("audio_to_embedding", "embedding_rows", "embedding_audio_id"):
    assert np.array_equal(getattr(mapping, f), getattr(rule, f)), f
  assert mapping.num_audio == 8192 and len(mapping.embedding_rows) == 8290
  assert int((mapping.embedding_audio_id == -1).sum()) == 98 and int(mapping.embedding_rows.max()) == 264191


def test_mapping_round_trip_and_ranges(mapping):
  a = np.arange(8192)
  rows = mapping.to_rows(a)
  assert np.array_equal(mapping.to_audio(rows), a)
  assert len(np.unique(rows)) == 8192
  assert mapping.audio_row_ranges() == [(6, 105), (256001, 264094)]
  assert np.array_equal(mapping.to_audio([0, 5, 105, 256000, 264094, 264191, 300000]), [-1] * 7)   # text rows, padding rows, out of table
  assert np.array_equal(mapping.to_rows([0, 98, 99, 8191]), [6, 104, 256001, 264093])
  with pytest.raises(ValueError):
    mapping.to_rows([8192])
  # the decoder-side helper of the synthetic prompts follows the same table
  assert np.array_equal(H.synth.audio_id_to_row(a), rows)


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_expand_embedder_appends_rows_and_columns(dtype):
  rng = np.random.default_rng(0)
  V, E, add = 1000, 64, 37
  emb = rng.standard_normal((V, E)).astype(np.float32)
  head = rng.standard_normal((E, V)).astype(np.float32)
  if dtype == "bf16":
    emb, head = H.synth.f32_to_bf16_bits(emb), H.synth.f32_to_bf16_bits(head)
  params = {VE.expand_embedder.__globals__["EMBEDDING"]: emb, VE.expand_embedder.__globals__["LOGITS"]: head,
            "params/decoder/decoder_norm/scale": np.ones((E,), np.float32)}
  out = VE.expand_embedder(params, num_tokens=add)
  ne, nh = out["params/token_embedder/embedding"], out["params/decoder/logits_dense/kernel"]
  Vp = (V + add + 15) // 16 * 16
  assert ne.shape == (Vp, E) and nh.shape == (E, Vp) and ne.dtype == emb.dtype and nh.dtype == head.dtype
  assert np.array_equal(ne[:V], emb) and np.array_equal(nh[:, :V], head)                 # the old vocabulary is untouched, bit for bit
  f = (lambda x: H.synth.bf16_bits_to_f32(x)) if dtype == "bf16" else (lambda x: x)
  new_rows, new_cols = f(ne[V:V + add]), f(nh[:, V:V + add])
  for x in (new_rows, new_cols):
    assert np.abs(x).max() <= 2 * 0.02 * 1.01 and 0.015 < x.std() < 0.02                 # truncated normal(+-2 sigma) * 0.02
  assert not f(ne[V + add:]).any() and not f(nh[:, V + add:]).any()                      # tile padding is zero
  assert out["params/decoder/decoder_norm/scale"] is params["params/decoder/decoder_norm/scale"]
  again = VE.expand_embedder(params, num_tokens=add)
  assert np.array_equal(again["params/token_embedder/embedding"], ne)                     # deterministic (seed 67)
  tied = VE.expand_embedder({"token_embedder/embedding": emb}, num_tokens=add)            # tied head / un-prefixed path
  assert tied["token_embedder/embedding"].shape == (Vp, E) and len(tied) == 1


def test_expand_checkpoint_file_safetensors(tmp_path):
  from safetensors.numpy import load_file, save_file
  rng = np.random.default_rng(1)
  src = {"params/token_embedder/embedding": rng.standard_normal((48, 32)).astype(np.float32),
         "params/decoder/logits_dense/kernel": rng.standard_normal((32, 48)).astype(np.float32)}
  a, b = str(tmp_path / "in.safetensors"), str(tmp_path / "out.safetensors")
  save_file(src, a)
  shapes = VE.expand_checkpoint_file(a, b, num_tokens=16)
  got = load_file(b)
  assert shapes["params/token_embedder/embedding"] == (64, 32) and got["params/decoder/logits_dense/kernel"].shape == (32, 64)
  assert np.array_equal(got["params/token_embedder/embedding"][:48], src["params/token_embedder/embedding"])


def test_oracle_sampling_ranges():
  """Range-restricted greedy sampling in the oracle: every token lies in the ranges, the logits are the unrestricted ones, and
  lifting the restriction restores the plain argmax."""
  from oracle import oracle as O
  cfg = H.make_config(per_device_batch_size=2.0)
  w = H.synth.synth_weights(cfg, 0)
  toks, tl = H.synth.synth_prompt(cfg.vocab_size, 32, 1234)
  ranges = [(6, 105), (600, 900)]

  def run(r):
    orc = H.make_oracle(cfg, w)
    O.OracleEngine.set_sampling_ranges(r)
    try:
      pre, res = orc.prefill(toks, tl, slot=0, want_logits=True)
      ds = orc.insert(pre, orc.init_decode_state(), 0)
      out = [int(res[0, 0])]
      for _ in range(12):
        ds, rr = orc.generate(ds)
        out.append(int(rr[0, 0]))
      return np.array(out), pre["logits"]
    finally:
      O.OracleEngine.set_sampling_ranges([])
  free, lg_free = run([])
  lim, lg_lim = run(ranges)
  assert all(any(lo <= t < hi for lo, hi in ranges) for t in lim)
  assert np.array_equal(lg_free.view(np.uint32), lg_lim.view(np.uint32))
  row = lg_lim.reshape(-1)
  allowed = np.zeros(row.shape, bool)
  for lo, hi in ranges:
    allowed[lo:hi] = True
  assert lim[0] == int(np.flatnonzero(allowed)[np.argmax(row[allowed])]) and free[0] == int(np.argmax(row))
  assert not np.array_equal(free, lim)


def test_other_reference_mapping_files_agree(mapping):
  """The reference ships three mapping data files; the adjusted one drives this build.  The other two
  (vocab_expansion/audio_token_mapping.json, sha256 abf2cbca..., and audio_token_mapping_complete.json, sha256 6644cd61...) are pinned
  here in array form and must tell the same story: same audio-id -> embedding-row table, and tokenizer indices that differ from
  embedding rows by exactly the removed soft token at 262144 (extend_tokenizer.py:40-46)."""
  g = np.load(os.path.join(H.GOLDEN, "audio_token_mappings_other.npz"))
  assert str(g["simple_sha256"]).startswith("abf2cbca") and str(g["complete_sha256"]).startswith("6644cd61")
  # complete file: the embedding-side tables are the adjusted file's tables
  assert np.array_equal(g["complete_audio_to_embedding"], mapping.audio_to_embedding)
  assert np.array_equal(g["complete_embedding_rows"], mapping.embedding_rows)
  assert np.array_equal(g["complete_embedding_audio_id"], mapping.embedding_audio_id)
  # tokenizer index -> embedding row: indices above the soft token (262144) shift down by one, the soft token itself never appears
  for tix, aid in ((g["simple_tokenizer_index"], g["simple_audio_id"]), (g["complete_tokenizer_index"], g["complete_tokenizer_audio_id"])):
    assert 262144 not in set(tix.tolist())
    rows = np.where(tix > 262144, tix - 1, tix)
    assert np.array_equal(rows, mapping.embedding_rows) and np.array_equal(aid, mapping.embedding_audio_id)
  assert np.array_equal(np.where(g["complete_detail_original_token_idx"] > 262144, g["complete_detail_original_token_idx"] - 1,
                                 g["complete_detail_original_token_idx"]), mapping.embedding_rows)
  assert int(g["simple_vocab_size"]) == 264193            # tokenizer size; the embedding table has one row less (264192)
  import json
  st = json.loads(str(g["complete_stats"]))
  assert st["adjusted_embedding_size"] == 264192 and st["total_audio_tokens"] == 8192 and st["padding_added"] == 98
