"""The HIP path against the INDEPENDENT restatement (oracle/ref_algo_cpu.py: the reference's dense two-segment cache, full-length masked attention,
torch f32 matmuls in torch's own order) -- not through the paged C oracle the kernels share their operation order with.  Same rules as
tests/test_cross_oracle_cpu.py: logits within the reference's own tolerance for the equivalent comparison (tests/model_test.py:115-191:
atol = rtol = 1e-1), greedy ids equal wherever the top-2 gap exceeds the disagreement observed; the dense side is fed the HIP ids, so both see
the same context at every step."""
import numpy as np
import pytest
import torch

import helpers as H
from oracle import ref_algo_cpu as RA
from test_cross_oracle_cpu import LOGIT_TOL, _check, _close, _top2_gap

pytestmark = pytest.mark.gpu


def _hip_against_dense(cfg, w, prompts, steps):
  B, P, T = len(prompts), cfg.max_prefill_predict_length, cfg.max_target_length
  eng, params = H.make_engine(cfg, w)
  dense = RA.RefAlgoCPU(cfg, w, B, P, T, matmul="f32", threads=4)
  ds = eng.init_decode_state(materialize_logits=True)
  worst, ids_equal, ids_compared, flips = 0.0, 0, 0, []
  for s, (toks, tl) in enumerate(prompts):
    pre, res = eng.prefill(params=params, padded_tokens=toks, true_length=tl, slot=s)
    first = dense.prefill(toks, tl, s)
    lh, ld = pre["logits"].cpu().numpy().reshape(-1), dense.last_logits.numpy().reshape(-1)
    assert _close(ld, lh, LOGIT_TOL), f"prefill logits of prompt {s}: max abs diff {np.abs(ld - lh).max()}"
    worst = max(worst, float(np.abs(ld - lh).max()))
    tok = int(res.data.cpu().numpy()[0, 0])
    ids_compared += 1
    if first == tok:
      ids_equal += 1
    else:
      flips.append(("prefill", s, _top2_gap(lh)))
    dense.tokens[s] = tok
    ds = eng.insert(pre, ds, s)
  for step in range(steps):
    ds, r = eng.generate(params, ds)
    got = dense.generate()
    hip_ids = r.data.cpu().numpy()[:, 0]
    lh, ld = ds["logits"].cpu().numpy().reshape(B, -1), dense.last_logits.numpy()
    assert _close(ld, lh, LOGIT_TOL), f"step {step}: max abs diff {np.abs(ld - lh).max()}"
    worst = max(worst, float(np.abs(ld - lh).max()))
    for s in range(B):
      ids_compared += 1
      if int(got[s]) == int(hip_ids[s]):
        ids_equal += 1
      else:
        flips.append((step, s, _top2_gap(lh[s])))
    dense.tokens = torch.from_numpy(hip_ids.astype(np.int64))
  assert eng.exchange_status() == 0
  eng.close()
  return {"worst_abs": worst, "ids_equal": ids_equal, "ids_compared": ids_compared, "flips": flips}


def test_hip_agrees_with_the_dense_restatement_tiny():
  cfg = H.make_config(per_device_batch_size=2.0, max_prefill_predict_length=32, max_target_length=96, pagedattn_num_pages=64)
  w = H.synth.synth_weights(cfg, 0)
  prompts = [H.synth.synth_prompt(cfg.vocab_size, 32, 1234), H.synth.synth_prompt(cfg.vocab_size, 32, 99)]
  stats = _hip_against_dense(cfg, w, prompts, 40)
  _check(stats)
  assert stats["worst_abs"] < LOGIT_TOL


def test_hip_agrees_with_the_dense_restatement_c1():
  """BASELINE.json configs[0]'s 134M decoder: one 64-token prompt, 32 greedy steps."""
  cfg = H.make_config(model_name="c1-134m", per_device_batch_size=1.0, max_prefill_predict_length=64, max_target_length=128,
                      pagedattn_tokens_per_page=32, pagedattn_num_pages=8)
  w = H.synth.synth_weights(cfg, 0)
  stats = _hip_against_dense(cfg, w, [H.synth.synth_prompt(cfg.vocab_size, 64, 1234)], 32)
  _check(stats)
  assert stats["worst_abs"] < LOGIT_TOL
