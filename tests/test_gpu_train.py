"""GPU parity of the training step (BASELINE.json configs[2] at test size): Trainer.train_step (HIP, include/ixt_train.h)
against the f32 PyTorch oracle (oracle/train_oracle.py) on identical batches.  Bars: loss within 1e-4 relative at every step (north_star's
training criterion); raw gradient norm within 3e-3 relative and every gradient tensor within 2 % in L2 of the oracle's (the product rounds its
backward signals to bf16 where they feed the matrix cores; the oracle does not)."""
import importlib

import numpy as np
import pytest

import helpers as H
from oracle import train_oracle as TO

pytestmark = pytest.mark.gpu

AT = importlib.import_module("maxtext-indextts2_amd.audio_training")


def train_config(**kw):
  base = dict(model_name="default", base_emb_dim=64, base_num_query_heads=2, base_num_kv_heads=2, head_dim=32, base_mlp_dim=128, base_num_decoder_layers=2,
              vocab_size=256, per_device_batch_size=2.0, max_target_length=16, steps=10, learning_rate=1e-2, warmup_steps_fraction=0.2)
  base.update(kw)
  return H.make_config(**base)


def f32_params(cfg, seed=0):
  return {k: H.synth.bf16_bits_to_f32(v).astype(np.float32) for k, v in H.synth.synth_weights(cfg, seed).items()}


def packed_batch(cfg, seed):
  """Rows with two packed segments and trailing padding: segment ids 1, 2, 0; positions restart per segment (sequence_packing layout)."""
  b = AT.synthetic_batch(cfg, seed)
  B, S = b["inputs"].shape
  for r in range(B):
    n1, n2 = S // 2 - r, S // 4 + r
    seg = np.array([1] * n1 + [2] * n2 + [0] * (S - n1 - n2), np.int32)
    pos = np.concatenate([np.arange(n1), np.arange(n2), np.zeros(S - n1 - n2)]).astype(np.int32)
    b["inputs_segmentation"][r] = seg
    b["inputs_position"][r] = pos
    b["targets_segmentation"][r] = seg
    b["targets_segmentation"][r, n1 - 1] = 0          # the last token of a segment has no target inside the segment
    b["targets_position"][r] = pos
  return b


@pytest.mark.parametrize("shape", ["tiny", "wide", "gqa", "gqa4"])
def test_train_steps_match_oracle(shape):
  import torch
  tr = importlib.import_module("maxtext-indextts2_amd.train")
  if shape == "tiny":
    cfg = train_config()
  elif shape == "gqa":      # grouped-query attention, two query heads per kv head (every Gemma / Mixtral configuration has groups)
    cfg = train_config(base_emb_dim=128, base_num_query_heads=4, base_num_kv_heads=2, head_dim=32, base_mlp_dim=256, vocab_size=512, max_target_length=40,
                       per_device_batch_size=2.0, learning_rate=3e-3)
  elif shape == "gqa4":     # four query heads per kv head (Mixtral-8x7B's 32 / 8), one kv head only
    cfg = train_config(base_emb_dim=128, base_num_query_heads=4, base_num_kv_heads=1, head_dim=32, base_mlp_dim=256, vocab_size=512, max_target_length=24,
                       per_device_batch_size=3.0, learning_rate=3e-3)
  else:
    cfg = train_config(base_emb_dim=256, base_num_query_heads=4, base_num_kv_heads=4, head_dim=64, base_mlp_dim=512, vocab_size=1024, max_target_length=136,
                       per_device_batch_size=3.0, learning_rate=3e-3)
  params = f32_params(cfg, 4)
  trainer = tr.Trainer(cfg, logits_chunk=128)     # several logits chunks at the wide shape (3 x 136 rows)
  trainer.load_params(params)
  orc = TO.TrainOracle(cfg, params)
  batches = [AT.synthetic_batch(cfg, 10), packed_batch(cfg, 11), AT.synthetic_batch(cfg, 12), packed_batch(cfg, 13)]
  for step, b in enumerate(batches):
    # identical state going into the step: parameters and AdamW moments of the oracle (Adam turns gradient noise at the 1e-3 level into
    # update differences at the level of lr, so free-running trajectories part at 1e-4..1e-3 in the loss after a few steps: second test)
    trainer.load_params({k: v.detach().numpy() for k, v in orc.p.items()})
    for buf, src in ((trainer.mu, orc.mu), (trainer.nu, orc.nu)):
      for k, v in trainer.param_views(buf).items():
        v.copy_(src[k])
    trainer.step = orc.step
    before = {k: v.detach().clone() for k, v in orc.p.items()}
    m = trainer.train_step(b)
    g_hip = {k: v.clone() for k, v in trainer.param_views(trainer.grads).items()}
    o = orc.train_step(b)
    assert m["learning/total_weights"] == o["learning/total_weights"]
    assert abs(m["learning/loss"] - o["learning/loss"]) <= 1e-4 * abs(o["learning/loss"]), (step, m["learning/loss"], o["learning/loss"])
    assert abs(m["learning/raw_grad_norm"] - o["learning/raw_grad_norm"]) <= 3e-3 * o["learning/raw_grad_norm"], (step, m, o["learning/raw_grad_norm"])
    assert abs(m["learning/grad_norm"] - o["learning/grad_norm"]) <= 3e-3 * o["learning/grad_norm"]
    assert abs(m["learning/current_learning_rate"] - o["learning/current_learning_rate"]) < 1e-12
    denom = o["learning/total_weights"] + 1e-8
    for k, g in g_hip.items():
      ref = o["grads"][k]
      got = g.cpu() / denom                           # the buffer holds the gradient of total_loss (train.py:307-322)
      err = float((got - ref).norm()) / (float(ref.norm()) + 1e-12)
      assert err < 2e-2, (step, k, err)
    # the AdamW update of this step: same direction and size as the oracle's, tensor by tensor
    if o["learning/current_learning_rate"] > 0:
      for k, v in trainer.param_views().items():
        d_hip, d_ref = v.cpu() - before[k], orc.p[k].detach() - before[k]
        if float(d_ref.norm()) > 0:
          cos = float((d_hip * d_ref).sum()) / (float(d_hip.norm()) * float(d_ref.norm()) + 1e-30)
          assert cos > 0.98 and abs(float(d_hip.norm()) / float(d_ref.norm()) - 1) < 2e-2, (step, k, cos)
  del trainer
  torch.cuda.synchronize()


def test_free_running_trajectory_stays_close():
  """Six steps on a repeated batch without re-synchronising (the loss falls five-fold): the trajectory stays within 1 % of the oracle's
  (Adam amplifies the bf16 noise of the product's backward signals: the per-step bar with identical state is the 1e-4 of the test above)."""
  import torch
  tr = importlib.import_module("maxtext-indextts2_amd.train")
  cfg = train_config(learning_rate=5e-3, warmup_steps_fraction=0.1)
  params = f32_params(cfg, 5)
  trainer = tr.Trainer(cfg)
  trainer.load_params(params)
  orc = TO.TrainOracle(cfg, params)
  b = packed_batch(cfg, 21)
  losses = []
  for step in range(6):
    m, o = trainer.train_step(b), orc.train_step(b)
    assert abs(m["learning/loss"] - o["learning/loss"]) <= 1e-2 * abs(o["learning/loss"]), (step, m["learning/loss"], o["learning/loss"])
    losses.append(m["learning/loss"])
  assert losses[-1] < losses[1] < losses[0] + 1e-6
  del trainer
  torch.cuda.synchronize()


def test_train_step_at_the_c3_widths():
  """BASELINE.json configs[2]'s model at its real widths -- emb 2048, 16 heads of 128, mlp 7168, the 264 192-row audio-expanded vocabulary, 2048-token
  rows -- with 2 of its 16 layers and one row per step (the CPU oracle is an f32 torch autograd pass: the logits alone are 2 GB): loss within 1e-4
  relative with synchronised state, raw gradient norm within 3e-3, every gradient tensor within 2 %."""
  import torch
  tr = importlib.import_module("maxtext-indextts2_amd.train")
  cfg = H.make_config(model_name="indextts2-1b", base_num_decoder_layers=2, per_device_batch_size=1.0, max_target_length=2048, steps=10, learning_rate=1e-4,
                      warmup_steps_fraction=0.2)
  torch.set_num_threads(16)
  params = f32_params(cfg, 6)
  trainer = tr.Trainer(cfg)
  trainer.load_params(params)
  orc = TO.TrainOracle(cfg, params)
  b = packed_batch(cfg, 31)
  m = trainer.train_step(b)
  g_hip = {k: v.clone() for k, v in trainer.param_views(trainer.grads).items()}
  o = orc.train_step(b)
  assert m["learning/total_weights"] == o["learning/total_weights"]
  assert abs(m["learning/loss"] - o["learning/loss"]) <= 1e-4 * abs(o["learning/loss"]), (m["learning/loss"], o["learning/loss"])
  assert abs(m["learning/raw_grad_norm"] - o["learning/raw_grad_norm"]) <= 3e-3 * o["learning/raw_grad_norm"], (m, o["learning/raw_grad_norm"])
  denom = o["learning/total_weights"] + 1e-8
  for k, g in g_hip.items():
    ref = o["grads"][k]
    err = float((g.cpu() / denom - ref).norm()) / (float(ref.norm()) + 1e-12)
    assert err < 2e-2, (k, err)
  del trainer
  torch.cuda.synchronize()


def test_trainer_rejects_what_it_does_not_build():
  tr = importlib.import_module("maxtext-indextts2_amd.train")
  with pytest.raises(ValueError):
    tr.Trainer(train_config(base_num_query_heads=3, base_num_kv_heads=2, base_emb_dim=96))     # query heads must be a multiple of the kv heads
  cfg = train_config()
  t = tr.Trainer(cfg)
  with pytest.raises(ValueError):
    t.train_step({k: np.zeros((1, 16), np.int32) for k in AT.BATCH_KEYS})
