"""Counterpart of the reference's example_audio_training.py (repo root of the reference) and of the batch contract its training path
expects (MaxText/maxtext_utils.py:85-99 `get_shaped_batch`; MaxText/input_pipeline/synthetic_data_processing.py:66-78 for how the six arrays
relate).  The reference's own Emilia iterators raise NotImplementedError (input_pipeline/_emilia_audio_processing.py:44-49) and its example
calls a `pyconfig.load_config` that does not exist (header facts), so "counterpart" means: the SAME override list parses, the
SAME tokenizer callback type is accepted, and the batches have the SAME six int32 [B, max_target_length] keys.

  audio bytes --(audio_tokenizer_fn: List[bytes] -> List[List[int]], ids in [0, 8192), example_audio_training.py:16-43)-->
  audio ids --(AudioTokenMapping, audio_token_mapping_adjusted.json)--> embedding rows --> inputs / targets shifted by one.
"""
from __future__ import annotations

import os
from typing import Callable, Dict, Iterable, Iterator, List, Optional, Sequence

import numpy as np

from . import pyconfig
from .vocab_expansion import AudioTokenMapping

AudioTokenizerFn = Callable[[List[bytes]], List[List[int]]]
BATCH_KEYS = ("inputs", "inputs_position", "inputs_segmentation", "targets", "targets_position", "targets_segmentation")

# example_audio_training.py:54-79, verbatim keys; vocab_size is 264192 here: the example's 264191 cannot hold max_embedding_index 264191
# (SURVEY.md section 8, C2 note) and is refused below with that explanation
EXAMPLE_OVERRIDES = [
    "dataset_type=emilia_audio", "audio_token_mapping_path=vocab_expansion/audio_token_mapping_adjusted.json", "audio_batch_size=32",
    "emilia_language=EN", "base_emb_dim=2048", "vocab_size=264192", "global_batch_size_to_load=128", "max_target_length=2048", "steps=100000",
    "learning_rate=1e-4", "per_device_batch_size=4", "base_output_directory=gs://your-bucket/audio-training", "run_name=emilia_audio_experiment",
]


def example_audio_tokenizer(audio_batch: List[bytes], seq_length: int = 1024, seed: int = 0) -> List[List[int]]:
  """Placeholder with the reference placeholder's contract (example_audio_training.py:16-43): one list of audio ids in [0, 8192) per clip.
  Deterministic here (seeded by the clip's bytes) so that tests can pin it."""
  out = []
  for clip in audio_batch:
    rng = np.random.Generator(np.random.PCG64(seed + (hash(bytes(clip)) & 0x7FFFFFFF)))
    out.append(rng.integers(0, 8192, size=seq_length).tolist())
  return out


def setup_audio_training_config(overrides: Optional[Sequence[str]] = None, config_path: Optional[str] = None):
  """example_audio_training.py:46-88: base.yml + the audio override list -> read-only config."""
  config_path = config_path or os.path.join(pyconfig.PKG_DIR, "configs", "base.yml")
  cfg = pyconfig.initialize(["audio_training", config_path] + list(EXAMPLE_OVERRIDES if overrides is None else overrides))
  if cfg.dataset_type == "emilia_audio" and cfg.vocab_size < 264192:
    raise ValueError(f"vocab_size={cfg.vocab_size} cannot hold the audio rows of audio_token_mapping_adjusted.json (max_embedding_index 264191): "
                     "use 264192 (example_audio_training.py:65 says 264191, one short)")
  return cfg


def get_shaped_batch(config) -> Dict[str, tuple]:
  """maxtext_utils.py:85-99: name -> (shape, dtype) of the batch the train step consumes."""
  shape = (int(config.global_batch_size_to_load), int(config.max_target_length))
  return {k: (shape, np.int32) for k in BATCH_KEYS}


class AudioBatchIterator:
  """Batches of the contract above from clips of audio bytes and the caller's tokenizer.

  `audio_batches`: iterable of List[bytes] (each list = `audio_batch_size` clips).  Every clip becomes one row: its audio ids, mapped to
  embedding rows, preceded by `bos_row`; inputs = row[:-1], targets = row[1:] (next-token prediction), positions 0.., segmentation 1 on
  real tokens and 0 on padding -- the relations of synthetic_data_processing.py:66-78.  Rows are cut / padded to max_target_length."""

  def __init__(self, config, audio_tokenizer_fn: AudioTokenizerFn, audio_batches: Iterable[List[bytes]], mapping: Optional[AudioTokenMapping] = None,
               bos_row: int = 264094):
    self.config, self.fn, self.src = config, audio_tokenizer_fn, iter(audio_batches)
    if mapping is None:
      path = config.audio_token_mapping_path
      mapping = AudioTokenMapping.from_npz(path) if path.endswith(".npz") else (AudioTokenMapping.from_json(path) if os.path.isfile(path) else AudioTokenMapping.from_rule())
    self.mapping, self.bos_row = mapping, bos_row
    self.B, self.T = int(config.global_batch_size_to_load), int(config.max_target_length)
    self._rows: List[np.ndarray] = []

  def __iter__(self) -> Iterator[Dict[str, np.ndarray]]:
    return self

  def _fill(self):
    while len(self._rows) < self.B:
      clips = next(self.src)                    # StopIteration ends the epoch
      ids = self.fn(list(clips))
      if len(ids) != len(clips):
        raise ValueError("audio_tokenizer_fn must return one token list per clip")
      for seq in ids:
        a = np.asarray(seq, np.int64)
        if a.size and (a.min() < 0 or a.max() >= 8192):
          raise ValueError("audio_tokenizer_fn returned ids outside [0, 8192)")
        self._rows.append(np.concatenate([[self.bos_row], self.mapping.to_rows(a)]).astype(np.int32))

  def __next__(self) -> Dict[str, np.ndarray]:
    self._fill()
    rows, self._rows = self._rows[:self.B], self._rows[self.B:]
    T = self.T
    tok = np.zeros((self.B, T + 1), np.int32)
    seg = np.zeros((self.B, T + 1), np.int32)
    for i, r in enumerate(rows):
      n = min(len(r), T + 1)
      tok[i, :n], seg[i, :n] = r[:n], 1
    pos = np.broadcast_to(np.arange(T + 1, dtype=np.int32), (self.B, T + 1)) * seg
    return {"inputs": tok[:, :-1].copy(), "inputs_position": pos[:, :-1].copy(), "inputs_segmentation": seg[:, :-1].copy(),
            "targets": tok[:, 1:].copy(), "targets_position": pos[:, 1:].copy(), "targets_segmentation": (seg[:, :-1] * seg[:, 1:]).copy()}


def synthetic_batch(config, seed: int = 0) -> Dict[str, np.ndarray]:
  """SyntheticDataIterator's batch (synthetic_data_processing.py:48-78): uniform token ids, full-length rows, segmentation all ones."""
  B, T = int(config.global_batch_size_to_load), int(config.max_target_length)
  rng = np.random.Generator(np.random.PCG64(seed))
  tok = rng.integers(0, int(config.vocab_size), size=(B, T + 1), dtype=np.int32)
  pos = np.broadcast_to(np.arange(T + 1, dtype=np.int32), (B, T + 1))
  one = np.ones((B, T), np.int32)
  return {"inputs": tok[:, :-1].copy(), "inputs_position": pos[:, :-1].copy(), "inputs_segmentation": one, "targets": tok[:, 1:].copy(),
          "targets_position": pos[:, 1:].copy(), "targets_segmentation": one.copy()}
