"""The slice of `jetstream.engine.engine_api` that MaxEngine's callers touch (third-party in the reference,
imported at /root/reference/MaxText/maxengine.py:39-44, not vendored there).  Restated from its call sites:
`ResultTokens(data, tokens_idx, valid_idx, length_idx, log_prob, samples_per_slot)` is built at
maxengine.py:506-518 / :915-926 and read as `t.get_result_at_slot(i).tokens.item()` at decode.py:178 and
`first_token.data`, `.log_prob` at tests/maxengine_test.py:111-164."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Optional, Tuple

import numpy as np


@dataclass
class SlotData:
  tokens: Any
  valid: Any
  lengths: Any
  log_prob: Any = None


@dataclass
class ResultTokens:
  """data is [batch, 3 * samples]: tokens | valid | lengths along axis 1 (slices given by the *_idx pairs)."""
  data: Any
  tokens_idx: Tuple[int, int]
  valid_idx: Tuple[int, int]
  length_idx: Tuple[int, int]
  samples_per_slot: int
  log_prob: Optional[Any] = None

  def convert_to_numpy(self) -> "ResultTokens":
    d = self.data
    if hasattr(d, "detach"):
      d = d.detach().cpu().numpy()
    lp = self.log_prob
    if lp is not None and hasattr(lp, "detach"):
      lp = lp.detach().cpu().numpy()
    return ResultTokens(np.asarray(d), self.tokens_idx, self.valid_idx, self.length_idx, self.samples_per_slot, lp)

  def get_result_at_slot(self, slot: int) -> SlotData:
    s0, s1 = slot * self.samples_per_slot, (slot + 1) * self.samples_per_slot
    lp = None if self.log_prob is None else self.log_prob[s0:s1]
    return SlotData(
        tokens=self.data[s0:s1, self.tokens_idx[0]:self.tokens_idx[1]],
        valid=self.data[s0:s1, self.valid_idx[0]:self.valid_idx[1]],
        lengths=self.data[s0:s1, self.length_idx[0]:self.length_idx[1]][:, 0],
        log_prob=lp,
    )


class Engine:
  """Abstract surface MaxEngine implements (names only; semantics in maxengine.py)."""

  def prefill(self, *, params, existing_prefix=None, padded_tokens, true_length, **kwargs):
    raise NotImplementedError

  def generate(self, params, decode_state, sampler=None, rng=None):
    raise NotImplementedError

  def insert(self, prefix, decode_state, slot, request_id=None):
    raise NotImplementedError

  def init_decode_state(self, *args, **kwargs):
    raise NotImplementedError

  def load_params(self, *args, **kwargs):
    raise NotImplementedError
