"""The example_audio_training.py counterpart (maxtext-indextts2_amd/audio_training.py): the reference's override list parses, the tokenizer
callback contract (List[bytes] -> List[List[int]] in [0, 8192)) is honoured, and batches carry the six int32 [B, max_target_length] keys of
maxtext_utils.get_shaped_batch (maxtext_utils.py:85-99)."""
import importlib
import os

import numpy as np
import pytest

import helpers as H

AT = importlib.import_module("maxtext-indextts2_amd.audio_training")
VE = importlib.import_module("maxtext-indextts2_amd.vocab_expansion")


def test_example_override_list_parses():
  cfg = AT.setup_audio_training_config()
  assert cfg.dataset_type == "emilia_audio" and cfg.emilia_language == "EN" and cfg.audio_batch_size == 32
  assert cfg.vocab_size == 264192 and cfg.max_target_length == 2048 and cfg.steps == 100000 and cfg.learning_rate == 1e-4
  assert cfg.global_batch_size_to_load == 128 and cfg.per_device_batch_size == 4.0 and cfg.emb_dim == 2048
  with pytest.raises(ValueError):   # read-only, pyconfig.py:1163-1164 (the example's `config.audio_tokenizer_fn = ...` cannot work there either)
    cfg.audio_tokenizer_fn = AT.example_audio_tokenizer
  with pytest.raises(ValueError, match="264192"):
    AT.setup_audio_training_config([o if not o.startswith("vocab_size") else "vocab_size=264191" for o in AT.EXAMPLE_OVERRIDES])
  shaped = AT.get_shaped_batch(cfg)
  assert list(shaped) == list(AT.BATCH_KEYS) and all(v == ((128, 2048), np.int32) for v in shaped.values())


def test_batches_follow_the_contract():
  cfg = AT.setup_audio_training_config(["dataset_type=emilia_audio", "vocab_size=264192", "global_batch_size_to_load=4", "max_target_length=16",
                                        "per_device_batch_size=4"])
  mapping = VE.AudioTokenMapping.from_npz(os.path.join(H.GOLDEN, "audio_token_mapping_adjusted.npz"))
  lens = [5, 30, 16, 17]
  calls = []

  def tok(clips):
    calls.append(len(clips))
    return [list(range(100 * i, 100 * i + n)) for i, n in zip(range(len(clips)), lens)]
  it = AT.AudioBatchIterator(cfg, tok, [[b"a", b"b"], [b"c", b"d"]], mapping=mapping)
  b = next(it)
  assert calls == [2, 2]
  for k in AT.BATCH_KEYS:
    assert b[k].shape == (4, 16) and b[k].dtype == np.int32, k
  # row 0: bos + 5 audio ids, padded; targets are inputs shifted by one
  r0 = np.concatenate([[264094], mapping.to_rows(np.arange(0, 5))])
  assert np.array_equal(b["inputs"][0, :6], r0) and not b["inputs"][0, 6:].any()
  assert np.array_equal(b["targets"][0, :5], r0[1:]) and np.array_equal(b["inputs_segmentation"][0], [1] * 6 + [0] * 10)
  assert np.array_equal(b["targets_segmentation"][0], [1] * 5 + [0] * 11) and np.array_equal(b["inputs_position"][0, :6], np.arange(6))
  # row 1 (30 ids) is cut to max_target_length; every id is an audio row of the expanded vocabulary
  assert b["inputs_segmentation"][1].all() and b["targets_segmentation"][1].all()
  rows = b["targets"][b["targets_segmentation"] == 1]
  assert all(any(lo <= r < hi for lo, hi in mapping.audio_row_ranges()) for r in rows)
  with pytest.raises(StopIteration):
    next(it)
  bad = AT.AudioBatchIterator(cfg, lambda clips: [[8192]] * len(clips), [[b"a"] * 4], mapping=mapping)
  with pytest.raises(ValueError):
    next(bad)


def test_placeholder_tokenizer_and_synthetic_batch():
  ids = AT.example_audio_tokenizer([b"x", b"y"], seq_length=64)
  assert len(ids) == 2 and all(len(s) == 64 and 0 <= min(s) and max(s) < 8192 for s in ids)
  assert ids == AT.example_audio_tokenizer([b"x", b"y"], seq_length=64)
  cfg = H.make_config(per_device_batch_size=2.0, max_target_length=32)
  b = AT.synthetic_batch(cfg, 3)
  assert np.array_equal(b["inputs"][:, 1:], b["targets"][:, :-1]) and b["inputs_segmentation"].all()
