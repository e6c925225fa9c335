#!/usr/bin/env python3
"""Do two half-batch engines on two streams of ONE GPU beat one full-batch engine?  The decode step alternates an HBM-bound attention launch
with latency-bound GEMM launches; two independent step graphs in flight let one engine's GEMMs run beside the other's attention.
   python tools/two_engine_probe.py [total_utterances=256] [decode_steps=256]"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def build(args, B, device, weights, seed0):
  import torch
  pyconfig = importlib.import_module("maxtext-indextts2_amd.pyconfig")
  maxengine = importlib.import_module("maxtext-indextts2_amd.maxengine")
  synth = importlib.import_module("maxtext-indextts2_amd.synth")
  cfg, eng, pack = bench.make_engine(args, pyconfig, maxengine, B)
  if weights is None:
    weights = bench.gpu_weights(cfg, device, seed=0)
  params = eng.load_params(params=weights)
  prompts = [synth.synth_prompt(cfg.vocab_size, args.prefill_len, seed0 + s) for s in range(B)]
  prompts_dev = [(torch.from_numpy(t).to(device), tl) for t, tl in prompts]
  return bench.Workload(eng, params, prompts_dev, args.decode_steps, pack), weights


def main():
  import torch
  total = int(sys.argv[1]) if len(sys.argv) > 1 else 256
  steps = int(sys.argv[2]) if len(sys.argv) > 2 else 256
  args = bench.parse(["--batch", str(total), "--decode-steps", str(steps)])
  device = torch.device("cuda", 0)
  one, weights = build(args, total, device, None, 1234)

  def run_one():
    ds = one.prefill_all(one.eng.init_decode_state())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    one.decode_all(ds)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    one.release_all()
    return dt
  run_one()
  t_one = run_one()
  print(f"one engine, {total} utterances: {t_one / steps * 1e3:.3f} ms per step, {total * steps / t_one:.0f} tokens/s", flush=True)
  del one
  torch.cuda.empty_cache()
  half = total // 2
  a, _ = build(args, half, device, weights, 1234)
  b, _ = build(args, half, device, weights, 1234 + half)
  sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

  def run_two(stagger):
    with torch.cuda.stream(sa):
      da = a.prefill_all(a.eng.init_decode_state())
    with torch.cuda.stream(sb):
      db = b.prefill_all(b.eng.init_decode_state())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
      with torch.cuda.stream(sa):
        da, _ = a.eng.generate(a.params, da)
      if stagger and i == 0:
        time.sleep(0.002)
      with torch.cuda.stream(sb):
        db, _ = b.eng.generate(b.params, db)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    a.release_all(); b.release_all()
    return dt
  run_two(False)
  for stagger in (False, True):
    t_two = run_two(stagger)
    print(f"two engines x {half} on two streams{' (second started 2 ms late)' if stagger else ''}: {t_two / steps * 1e3:.3f} ms per step of both, "
          f"{total * steps / t_two:.0f} tokens/s ({t_one / t_two:.3f} x)", flush=True)


if __name__ == "__main__":
  main()
