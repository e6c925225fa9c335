#!/usr/bin/env python3
"""BASELINE.json configs[2] on ONE GPU's share: the 1B IndexTTS2-style decoder (vocab 264192), per-device batch 8 x 2048 tokens
(example_audio_training.py:69: max_target_length=2048; 64 utterances over 8 GPUs), a few train steps on a synthetic batch.
Reports step time, tokens/s and TFLOP/s against SURVEY.md section 8(d)'s 155.4 TFLOP per step per GPU.  Run on the GPU box."""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench


def main():
  steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
  pdb, S = (int(sys.argv[2]) if len(sys.argv) > 2 else 8), (int(sys.argv[3]) if len(sys.argv) > 3 else 2048)
  pyconfig = importlib.import_module("maxtext-indextts2_amd.pyconfig")
  tr = importlib.import_module("maxtext-indextts2_amd.train")
  at = importlib.import_module("maxtext-indextts2_amd.audio_training")
  cfg = pyconfig.initialize(["train_bench", os.path.join(pyconfig.PKG_DIR, "configs", "base.yml")], model_name="indextts2-1b", per_device_batch_size=float(pdb),
                            max_target_length=S, steps=1000, learning_rate=1e-4, dataset_type="synthetic", run_name="train_bench")
  dev = torch.device("cuda", 0)
  torch.cuda.set_device(0)
  t0 = time.time()
  trainer = tr.Trainer(cfg)
  trainer.load_params(bench.gpu_weights(cfg, dev, seed=0))
  torch.cuda.synchronize()
  print(f"[train_bench] state ready in {time.time() - t0:.1f} s; {trainer.params.numel() / 1e9:.3f} B parameters", file=sys.stderr, flush=True)
  batch = {k: torch.from_numpy(v).to(dev) for k, v in at.synthetic_batch(cfg, 0).items()}
  out = []
  for i in range(steps):
    torch.cuda.synchronize()
    t1 = time.time()
    m = trainer.train_step(batch)
    torch.cuda.synchronize()
    m["step_s"] = time.time() - t1
    out.append(m)
    print(f"[train_bench] step {i}: loss {m['learning/loss']:.4f} raw_grad_norm {m['learning/raw_grad_norm']:.4f} lr {m['learning/current_learning_rate']:.2e} "
          f"{m['step_s']:.3f} s", file=sys.stderr, flush=True)
  best = min(m["step_s"] for m in out[1:]) if steps > 1 else out[0]["step_s"]
  # the reference's own count (maxtext_utils.calculate_tflops_training_per_device; SURVEY.md 8(d): 155.4 TFLOP per step per GPU at 8 x 2048)
  tflop = importlib.import_module("maxtext-indextts2_amd.maxtext_utils").calculate_tflops_training_per_device(cfg, log=False)[0]
  print(json.dumps({"config": f"C3 share: indextts2-1b, per_device_batch {pdb} x {S} tokens, 1 GPU", "steps": steps, "step_s": best, "tokens_per_s": pdb * S / best,
                    "tflop_per_s": tflop / best, "mfu_bf16": tflop / best / 2500.0, "losses": [m["learning/loss"] for m in out], "ln_vocab": float(np.log(cfg.vocab_size))}))


if __name__ == "__main__":
  main()
