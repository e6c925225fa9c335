"""CPU checks of the training oracle (oracle/train_oracle.py) against independent implementations, and of the data-parallel gradient
exchange (maxtext-indextts2_amd/dp.exchange_gradients) on two gloo ranks."""
import importlib
import math
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import helpers as H
from oracle import train_oracle as TO

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AT = importlib.import_module("maxtext-indextts2_amd.audio_training")


def train_config(**kw):
  base = dict(model_name="default", base_emb_dim=64, base_num_query_heads=2, base_num_kv_heads=2, head_dim=32, base_mlp_dim=128, base_num_decoder_layers=2,
              vocab_size=256, per_device_batch_size=2.0, max_target_length=16, steps=10, learning_rate=1e-2, warmup_steps_fraction=0.2)
  base.update(kw)
  return H.make_config(**base)


def f32_params(cfg, seed=0):
  return {k: H.synth.bf16_bits_to_f32(v).astype(np.float32) for k, v in H.synth.synth_weights(cfg, seed).items()}


def test_schedule_matches_closed_form_and_product():
  cfg = train_config(steps=100, learning_rate=3e-3, warmup_steps_fraction=0.1, cosine_learning_rate_final_fraction=0.1)
  tr = importlib.import_module("maxtext-indextts2_amd.train")
  sched = tr.create_learning_rate_schedule(cfg)
  assert sched(0) == 0.0 and abs(sched(5) - 1.5e-3) < 1e-12 and abs(sched(10) - 3e-3) < 1e-12          # optax.linear_schedule 0 -> lr over 10 steps
  assert abs(sched(55) - (3e-3 * 0.5 + 3e-4 * 0.5)) < 1e-12 and abs(sched(99) - TO.learning_rate(cfg, 99)) < 1e-15
  assert sched(100) == 0.0 and all(abs(sched(s) - TO.learning_rate(cfg, s)) < 1e-15 for s in range(0, 120))
  assert all(sched(s + 1) < sched(s) for s in range(10, 99))


def test_oracle_adamw_equals_torch_adamw():
  """optax.adamw (optimizers.py:28-38) == torch.optim.AdamW for eps_root = 0 (decoupled decay, eps outside the root, bias correction)."""
  cfg = train_config(gradient_clipping_threshold=0.0, warmup_steps_fraction=0.0, cosine_learning_rate_final_fraction=1.0, learning_rate=1e-2)
  orc = TO.TrainOracle(cfg, f32_params(cfg))
  ref = {k: v.detach().clone().requires_grad_(True) for k, v in orc.p.items()}
  opt = torch.optim.AdamW(ref.values(), lr=1e-2, betas=(cfg.adam_b1, cfg.adam_b2), eps=cfg.adam_eps, weight_decay=cfg.adam_weight_decay)
  batch = AT.synthetic_batch(cfg, 1)
  for _ in range(3):
    m = orc.train_step(batch)
    for k, v in ref.items():
      v.grad = m["grads"][k].clone()
    opt.step()
  for k in ref:
    assert torch.allclose(ref[k], orc.p[k].detach(), rtol=2e-6, atol=1e-8), k


def test_oracle_gradient_is_the_gradient():
  """Finite differences of the oracle's own loss on a few coordinates (f64 would be cleaner; the bf16 rounding makes the loss piecewise
  flat, so the check uses a step far above the rounding grain and a loose bar)."""
  cfg = train_config()
  orc = TO.TrainOracle(cfg, f32_params(cfg))
  batch = AT.synthetic_batch(cfg, 2)
  tl, tw = orc.loss(batch)
  tl.backward()
  name = "params/decoder/decoder_norm/scale"
  g = orc.p[name].grad.clone()
  with torch.no_grad():
    d = torch.sign(g) * 0.05
    orc.p[name] += d
  tl2, _ = orc.loss(batch)
  pred = float((g * d).sum())
  assert abs((float(tl2) - float(tl)) - pred) < 0.2 * abs(pred) + 1e-3


def _free_port():
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _dp_worker(rank, world, port, algo, q):
  sys.path.insert(0, ROOT)
  os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
  dp = importlib.import_module("maxtext-indextts2_amd.dp")
  dp.init("gloo")
  n = 1000 + 7                                    # not a multiple of the world size: the tail bucket path
  g = torch.arange(n, dtype=torch.float32) * (rank + 1)
  lw = torch.tensor([10.0 * (rank + 1), 3.0 + rank])
  dp.exchange_gradients(g, lw, algorithm=algo, bucket_elems=256)
  if rank == 0:
    q.put((g.tolist(), lw.tolist()))
  dp.finish()


@pytest.mark.parametrize("algo", ["rs_ag", "all_reduce"])
def test_gradient_exchange_world_size_2(algo):
  """The one collective of the build: after the exchange every rank holds the SUM of the gradient buffers and of [total_loss, total_weights]
  (loss = sum / sum, train.py:307-322); reduce-scatter + all-gather buckets and the all-reduce form agree."""
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, algo, q)) for r in range(2)]
  for p in procs:
    p.start()
  g, lw = q.get(timeout=120)
  for p in procs:
    p.join(timeout=120)
    assert p.exitcode == 0
  assert g == [3.0 * i for i in range(1007)] and lw == [30.0, 7.0]
