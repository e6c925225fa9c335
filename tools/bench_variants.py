#!/usr/bin/env python3
"""Rates of the two other model families of SURVEY.md section 8(f) at their BASELINE.json sizes (configs[3] and configs[4]), one GPU, synthetic
weights drawn ON the device (the trees are 5 GB and 93 GB; the host-side generator of synth.py would take minutes):

  gemma2   Gemma-2 2B (configs/models/gemma2-2b.yml) + 2048 appended vocabulary rows: one 4096-token prompt through the chunked prefill in
           256-row chunks, then greedy decode steps on 8 slots at that context;
  mixtral  Mixtral-8x7B (configs/models/mixtral-8x7b.yml, all 32 layers, 8 experts, 2 per token): 256-token prompts, greedy decode on 8 slots.

Prints one JSON line per family.  Not the headline bench (bench.py); parity of both blocks is the GPU suite's job (tests/test_gpu_gemma2.py,
tests/test_gpu_mixtral.py) -- this only times them.  Usage: python3 tools/bench_variants.py [gemma2] [mixtral] [--layers N]"""
import argparse
import importlib
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

pyconfig = importlib.import_module("maxtext-indextts2_amd.pyconfig")
maxengine = importlib.import_module("maxtext-indextts2_amd.maxengine")
synth = importlib.import_module("maxtext-indextts2_amd.synth")
BASE = os.path.join(pyconfig.PKG_DIR, "configs", "base.yml")


DeviceTree = synth.DeviceTree                      # weights drawn on the device, tensor by tensor (synth.py)
gemma2_spec = mixtral_spec = synth.device_tree_spec


OPTS = []          # library options name=value (ixt_set_option) given with --opt: experiments


def run(name, cfg, spec, prompt_len, chunk, slots, steps):
  t0 = time.perf_counter()
  eng = maxengine.MaxEngine(cfg)
  for kv in OPTS:
    k, v = kv.split("=")
    importlib.import_module("maxtext-indextts2_amd._lib").check(eng.lib.ixt_set_option(eng._ctx, k.encode(), int(v)))
  params = eng.load_params(params=DeviceTree(spec))
  torch.cuda.synchronize()
  t_load = time.perf_counter() - t0
  rng = np.random.default_rng(7)
  toks = [torch.from_numpy(rng.integers(1, cfg.vocab_size, size=prompt_len).astype(np.int32)).to(eng.device) for _ in range(slots)]

  def prefill(slot):
    done, pre = 0, None
    while done < prompt_len:
      n = min(chunk, prompt_len - done)
      ep = None if done == 0 else maxengine.ExistingPrefix(cache=pre["cache"], common_prefix_tokens=toks[slot][:done])
      pre, _ = eng.prefill(params=params, padded_tokens=toks[slot][done:done + n], true_length=n, slot=slot, existing_prefix=ep)
      done += n
    return pre

  pre = prefill(0)                       # warm-up (kernel load), then the timed prompt
  torch.cuda.synchronize()
  eng.release_pages(0)
  t1 = time.perf_counter()
  pre = prefill(0)
  torch.cuda.synchronize()
  t_prefill = time.perf_counter() - t1
  ds = eng.init_decode_state()
  ds = eng.insert(pre, ds, 0)
  for s in range(1, slots):
    ds = eng.insert(prefill(s), ds, s)
  for _ in range(4):
    ds, _ = eng.generate(params, ds)
  torch.cuda.synchronize()
  t2 = time.perf_counter()
  for _ in range(steps):
    ds, res = eng.generate(params, ds)
  torch.cuda.synchronize()
  t_dec = (time.perf_counter() - t2) / steps
  nparams = sum(int(np.prod(s)) for _, s, std in spec if std is not None)
  out = {"workload": name, "params": nparams, "weights_gb": nparams * 2 / 1e9, "load_s": round(t_load, 1), "prompt_tokens": prompt_len, "prefill_chunk": chunk,
         "prefill_s": t_prefill, "prefill_tokens_per_s": prompt_len / t_prefill, "decode_slots": slots, "decode_steps": steps,
         "options": list(OPTS), "quantization": str(cfg.quantization or "bf16"), "decode_ms_per_step": t_dec * 1e3, "decode_tokens_per_s": slots / t_dec, "weights_hbm_frac": nparams * 2 / t_dec / 8e12,
         "build_id": eng.lib.ixt_build_id().decode() if hasattr(eng.lib, "ixt_build_id") else None}
  print(json.dumps(out), flush=True)
  del eng
  torch.cuda.empty_cache()


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("which", nargs="*", default=["gemma2", "mixtral"])
  ap.add_argument("--layers", type=int, default=0, help="override base_num_decoder_layers (0: the model file's)")
  ap.add_argument("--steps", type=int, default=64)
  ap.add_argument("--opt", action="append", default=[], help="library option name=value (ixt_set_option)")
  ap.add_argument("--fp8", action="store_true", help="quantization: fp8 (IXT-FP8-v1: e4m3 projection kernels on the fp8 matrix cores)")
  args = ap.parse_args()
  OPTS.extend(args.opt)
  common = dict(attention="paged", scan_layers=False, pagedattn_tokens_per_page=32, run_name="bench_variants", use_chunked_prefill=True, override_model_config=True)
  if args.fp8:
    common["quantization"] = "fp8"
  if "gemma2" in args.which:
    kw = dict(base_num_decoder_layers=args.layers) if args.layers else {}
    cfg = pyconfig.initialize(["bench", BASE], model_name="gemma2-2b", vocab_size=256128 + 2048, per_device_batch_size=8.0, max_prefill_predict_length=256,
                              prefill_chunk_size=256, max_target_length=4096 + 128, pagedattn_num_pages=8 * 132 + 2, **common, **kw)
    run("gemma2-2b + 2048 rows: 4096-token chunked prefill (256-row chunks), decode on 8 slots", cfg, gemma2_spec(cfg), 4096, 256, 8, args.steps)
  if "mixtral" in args.which:
    kw = dict(base_num_decoder_layers=args.layers) if args.layers else {}
    cfg = pyconfig.initialize(["bench", BASE], model_name="mixtral-8x7b", per_device_batch_size=8.0, max_prefill_predict_length=256, prefill_chunk_size=256,
                              max_target_length=256 + 128, pagedattn_num_pages=8 * 12 + 2, **common, **kw)
    run("mixtral-8x7b: 256-token prompts, decode on 8 slots (routed experts: every copy in the rows of its expert)", cfg, mixtral_spec(cfg), 256, 256, 8, args.steps)


if __name__ == "__main__":
  main()
