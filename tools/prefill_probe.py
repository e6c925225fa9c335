#!/usr/bin/env python3
"""Time the packed prefill of the bench workload (8 prompts x 256 tokens per call) under engine options:
   python tools/prefill_probe.py attn_tile=0 attn_tile=1 ...   (each argument = one variant; "a=1,b=2" sets several options)."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
  import torch
  args = bench.parse(["--batch", "16", "--decode-steps", "64"])
  pyconfig = importlib.import_module("maxtext-indextts2_amd.pyconfig")
  maxengine = importlib.import_module("maxtext-indextts2_amd.maxengine")
  synth = importlib.import_module("maxtext-indextts2_amd.synth")
  lib = importlib.import_module("maxtext-indextts2_amd._lib")
  device = torch.device("cuda", 0)
  cfg, eng, pack = bench.make_engine(args, pyconfig, maxengine, args.batch)
  params = eng.load_params(params=bench.gpu_weights(cfg, device, seed=0))
  prompts = [synth.synth_prompt(cfg.vocab_size, args.prefill_len, 1234 + s) for s in range(args.batch)]
  prompts_dev = [(torch.from_numpy(t).to(device), tl) for t, tl in prompts]
  wl = bench.Workload(eng, params, prompts_dev, args.decode_steps, pack)
  for variant in sys.argv[1:] or ["attn_tile=1"]:
    for kv in variant.split(","):
      k, v = kv.split("=")
      lib.check(eng.lib.ixt_set_option(eng._ctx, k.encode(), int(v)))
    best = 1e9
    for it in range(4):
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      wl.prefill_all(eng.init_decode_state())
      torch.cuda.synchronize()
      best = min(best, time.perf_counter() - t0)
      wl.release_all()
    print(f"{variant:40s} {best * 1e3 / args.batch:8.3f} ms per utterance ({best * 1e3 / (args.batch / pack):.2f} ms per packed call)", flush=True)


if __name__ == "__main__":
  main()
