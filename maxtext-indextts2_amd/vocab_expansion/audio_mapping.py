"""The audio-code <-> embedding-row mapping of the expanded vocabulary.

Reference data file: audio_token_mapping_adjusted.json (generator vocab_expansion/extend_tokenizer.py:29-46, :49-257):
  audio id a in [0, 8192):  a < 99 -> row 6 + a (the tokenizer's unused slots);  a >= 99 -> row 256001 + (a - 99)
  rows 264094..264191 are padding entries (audio id -1); the soft token at original index 262144 is removed and every index
  above it shifts down by one (extend_tokenizer.py:40-46).
The table is DATA and is read from the JSON (or the array form of it); `from_rule` builds the same table from the rule above and
exists so that tests can check the rule against the data file.
"""
from __future__ import annotations

import json
from dataclasses import dataclass
from typing import List, Tuple

import numpy as np


@dataclass
class AudioTokenMapping:
  audio_to_embedding: np.ndarray   # int32 [num_audio]  audio id -> embedding row
  embedding_rows: np.ndarray       # int32 [M] every mapped row, ascending (audio rows and padding rows)
  embedding_audio_id: np.ndarray   # int32 [M] audio id of embedding_rows[i], -1 for padding rows

  # ------------------------------------------------------------------ constructors
  @classmethod
  def from_json(cls, path: str) -> "AudioTokenMapping":
    """The reference's JSON: {"audio_to_embedding": {"<audio id>": row}, "embedding_to_audio": {"<row>": audio id}, "stats": ...}."""
    with open(path, "rb") as f:
      d = json.load(f)
    n = 1 + max(int(k) for k in d["audio_to_embedding"])
    a2e = np.full((n,), -1, np.int32)
    for k, v in d["audio_to_embedding"].items():
      a2e[int(k)] = int(v)
    rows = np.array(sorted(int(k) for k in d["embedding_to_audio"]), np.int32)
    aud = np.array([int(d["embedding_to_audio"][str(r)]) for r in rows], np.int32)
    return cls(a2e, rows, aud)

  @classmethod
  def from_npz(cls, path: str) -> "AudioTokenMapping":
    with np.load(path) as z:
      return cls(z["audio_to_embedding"].astype(np.int32), z["embedding_rows"].astype(np.int32), z["embedding_audio_id"].astype(np.int32))

  @classmethod
  def from_rule(cls, num_audio: int = 8192, low_slots: int = 99, low_base: int = 6, high_base: int = 256001, padded_rows_end: int = 264192):
    a = np.arange(num_audio, dtype=np.int32)
    a2e = np.where(a < low_slots, low_base + a, high_base + (a - low_slots)).astype(np.int32)
    pad = np.arange(int(a2e.max()) + 1, padded_rows_end, dtype=np.int32)
    rows = np.concatenate([a2e, pad])
    aud = np.concatenate([a, np.full(pad.shape, -1, np.int32)])
    order = np.argsort(rows, kind="stable")
    return cls(a2e, rows[order], aud[order])

  # ------------------------------------------------------------------ lookups
  @property
  def num_audio(self) -> int:
    return int(self.audio_to_embedding.shape[0])

  def to_rows(self, audio_ids) -> np.ndarray:
    """audio ids -> embedding rows (= the token ids the decoder consumes and produces)."""
    a = np.asarray(audio_ids)
    if a.size and (a.min() < 0 or a.max() >= self.num_audio):
      raise ValueError(f"audio id outside [0, {self.num_audio})")
    return self.audio_to_embedding[a]

  def to_audio(self, rows) -> np.ndarray:
    """embedding rows -> audio ids; -1 for padding rows and for rows that are not audio tokens at all."""
    r = np.asarray(rows)
    idx = np.searchsorted(self.embedding_rows, r)
    idx = np.clip(idx, 0, len(self.embedding_rows) - 1)
    hit = self.embedding_rows[idx] == r
    return np.where(hit, self.embedding_audio_id[idx], -1).astype(np.int32)

  def audio_row_ranges(self) -> List[Tuple[int, int]]:
    """Half-open runs [lo, hi) of rows that carry a real audio code (padding rows excluded), ascending: the columns a decoder
    that must emit audio codes may sample from ([(6, 105), (256001, 264094)] for the reference table)."""
    rows = np.sort(self.audio_to_embedding[self.audio_to_embedding >= 0])
    if rows.size == 0:
      return []
    brk = np.flatnonzero(np.diff(rows) != 1)
    starts = np.concatenate([[0], brk + 1])
    ends = np.concatenate([brk, [rows.size - 1]])
    return [(int(rows[s]), int(rows[e]) + 1) for s, e in zip(starts, ends)]
