"""The attention-mask rules of the path, exact and boolean (the reference's own mask tests, tests/attention_test.py:45-260, pin its mask
builders with literal expected matrices; the cases that touch THIS path -- plain causality, causality inside packed segments, the
chunk offset -- are restated here against the oracle's restatement of `generate_attention_mask` and of `ref_ragged_paged_attention`,
and against what the build's row descriptors (slot, position) express)."""
import numpy as np
import pytest

from oracle import oracle as O


def test_causal_cases_of_the_reference_tests():
  # tests/attention_test.py:216-229 `test_full_length_chunk`: one chunk covering the sequence is the lower-triangular causal mask
  assert np.array_equal(O.reference_prefill_mask(6, 6), np.tril(np.ones((6, 6), bool)))
  # :138-141 `test_combine_with_causal_mask`: the causal operand it builds, (col_ids <= row_ids), literally
  row, col = np.arange(6)[:, None], np.arange(6)[None, :]
  assert np.array_equal(O.reference_prefill_mask(6, 6), col <= row)
  # its expected matrices with the image (bidirectional) blocks removed = the causal part: first column all True, strictly upper part False
  m = O.reference_prefill_mask(6, 6)
  assert m[:, 0].all() and not np.triu(m, 1).any()
  # :231-243 `test_single_token_chunk` is the identity: a row descriptor at position 0 of its own slot sees exactly itself
  vis = O.row_visibility([0, 1, 2], [0, 0, 0], 4)
  assert np.array_equal(vis[:, :, 0], np.eye(3, dtype=bool)) and not vis[:, :, 1:].any()


@pytest.mark.parametrize("q_len,prev", [(1, 0), (4, 0), (4, 3), (7, 9), (16, 16), (1, 31)])
def test_chunk_offset_rule_equals_ragged_paged_rule(q_len, prev):
  """SARATHI offset (attentions.py:611-614, `col <= row + next_pos`, next_pos = previous chunks) == ref_ragged_paged_attention's
  (kv_len - q_len) + row >= col with kv_len = prev + q_len (paged_attention_kernel_v2.py:100-103) == rows at positions prev..prev+q_len-1."""
  kv_len = prev + q_len
  a = O.reference_prefill_mask(q_len, kv_len, next_pos=prev)
  b = O.ragged_paged_mask(q_len, kv_len)
  c = O.row_visibility(np.zeros(q_len, int), prev + np.arange(q_len), kv_len)[:, 0, :]
  assert np.array_equal(a, b) and np.array_equal(a, c)
  # decode is the q_len == 1 case: the new token sees every cached position and itself (attentions.py:594-596: next_pos = kv_len - 1)
  if q_len == 1:
    assert a.all()


def test_packed_segments_equal_per_slot_rows():
  """prefill_concat packs prompts into one sequence with segment ids 1, 2, .. and per-prompt positions (maxengine.py:719-760): the mask is
  segment equality AND causality on the PACKED index.  The build computes the same prompts as rows (slot = segment, position inside the
  prompt): both say 'token j of prompt i sees tokens 0..j of prompt i and nothing else'."""
  lens = [5, 1, 7, 3]
  seg = np.concatenate([np.full(n, i + 1) for i, n in enumerate(lens)])
  S = len(seg)
  m = O.reference_prefill_mask(S, S, 0, seg, seg)
  slot = np.concatenate([np.full(n, i) for i, n in enumerate(lens)])
  pos = np.concatenate([np.arange(n) for n in lens])
  vis = O.row_visibility(slot, pos, max(lens))
  starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
  dense = np.zeros((S, S), bool)
  for r in range(S):
    for s_, st in enumerate(starts):
      dense[r, st:st + lens[s_]] = vis[r, s_, :lens[s_]]
  assert np.array_equal(m, dense)


def test_oracle_attention_follows_the_mask():
  """The oracle's paged attention over rows (orc_attend_row through OracleEngine.prefill_many) gives, for packed prompts, exactly what
  separate prompts give: no row sees another prompt's keys."""
  import helpers as H
  cfg = H.make_config(per_device_batch_size=3.0)
  w = H.synth.synth_weights(cfg, 0)
  o1, o2 = H.make_oracle(cfg, w), H.make_oracle(cfg, w)
  prompts = [(H.synth.synth_prompt(cfg.vocab_size, 32, 10 + s)[0], tl) for s, tl in enumerate([5, 17, 32])]
  _, firsts, lg = o1.prefill_many(prompts, want_logits=True)
  for s, (t, tl) in enumerate(prompts):
    p, r = o2.prefill(t, tl, slot=s, want_logits=True)
    assert r[0, 0] == firsts[s] and np.array_equal(p["logits"].view(np.uint32), lg[s:s + 1].view(np.uint32))
  assert np.array_equal(o1.model.kpages, o2.model.kpages) and np.array_equal(o1.model.vpages, o2.model.vpages)
