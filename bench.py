#!/usr/bin/env python3
"""Headline benchmark: audio tokens / second over prefill + decode of the 1B IndexTTS2-style decoder (BASELINE.json
configs[1], "C2"), data-parallel over N GPUs of one node (one process per GPU, no collective on the data path).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic input: `--batch` (default 256) utterances per GPU,
each a 256-token prompt (4 s reference speech + 128-char text) prefilled into its slot, then
`--decode-steps` (default 1024) greedy generate calls for all slots.  Tokens counted = generated audio tokens
including the first one from prefill (decode.py:162-178: T-P generate calls yield T-P+1 tokens per utterance).
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK = 8.0e12          # B/s, MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s spec"
MFMA_BF16_PEAK = 2.5e15    # FLOP/s dense bf16 (same guide)
MFMA_F32_PEAK = 157.3e12   # FLOP/s f32-input MFMA (same guide, "Peak FP32 (matrix)")


def parse():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=2)
  ap.add_argument("--warmup", type=int, default=1)
  ap.add_argument("--batch", type=int, default=256, help="utterances (decode slots) per GPU")
  ap.add_argument("--decode-steps", type=int, default=1024)
  ap.add_argument("--prefill-len", type=int, default=256)
  ap.add_argument("--model", default="indextts2-1b")
  ap.add_argument("--no-graph", action="store_true")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--no-roofline", action="store_true")
  ap.add_argument("--cpu-decode-steps", type=int, default=4)
  return ap.parse_args()


def gpu_weights(cfg, device, seed=0):
  """Random-init weights of the architecture, drawn on the GPU (fast); same distributions as synth.synth_weights."""
  synth = importlib.import_module("maxtext-indextts2_amd.synth")
  g = torch.Generator(device=device)
  g.manual_seed(seed)
  E, Q, KV, D, F, V, L = (cfg.emb_dim, cfg.num_query_heads, cfg.num_kv_heads, cfg.head_dim, cfg.mlp_dim, cfg.vocab_size, cfg.num_decoder_layers)

  def dense(shape, fan_in, div=1.0):
    t = torch.empty(shape, dtype=torch.float32, device=device)
    torch.nn.init.trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0, generator=g)
    return (t * ((1.0 / fan_in) ** 0.5 / 0.87962566103423978 / div)).to(torch.bfloat16)

  w = {"params/token_embedder/embedding": torch.randn((V, E), generator=g, device=device, dtype=torch.float32).to(torch.bfloat16)}
  one = torch.ones((E,), dtype=torch.bfloat16, device=device)
  for i in range(L):
    p = synth.layer_paths(i)
    w[p["ln1"]] = one
    w[p["q"]] = dense((E, Q, D), E, D ** 0.5)
    w[p["k"]] = dense((E, KV, D), E)
    w[p["v"]] = dense((E, KV, D), E)
    w[p["o"]] = dense((Q, D, E), Q * D)
    w[p["ln2"]] = one
    w[p["wi0"]] = dense((E, F), E)
    w[p["wi1"]] = dense((E, F), E)
    w[p["wo"]] = dense((F, E), F)
  w["params/decoder/decoder_norm/scale"] = one
  w["params/decoder/logits_dense/kernel"] = dense((E, V), E)
  return w


def model_numbers(cfg):
  E, Q, KV, D, F, V, L = (cfg.emb_dim, cfg.num_query_heads, cfg.num_kv_heads, cfg.head_dim, cfg.mlp_dim, cfg.vocab_size, cfg.num_decoder_layers)
  body = L * (E * (Q * D + 2 * KV * D) + Q * D * E + 2 * E * F + F * E)
  unembed = E * V
  kv_bytes_per_token = 2 * KV * D * 2 * L
  return {"body_params": body, "unembed_params": unembed, "kv_bytes_per_token": kv_bytes_per_token}


def main():
  args = parse()
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  world = int(os.environ.get("WORLD_SIZE", "1"))
  if world > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
  torch.cuda.set_device(local_rank)
  device = torch.device("cuda", local_rank)

  pyconfig = importlib.import_module("maxtext-indextts2_amd.pyconfig")
  maxengine = importlib.import_module("maxtext-indextts2_amd.maxengine")
  synth = importlib.import_module("maxtext-indextts2_amd.synth")
  B, P, T = args.batch, args.prefill_len, args.prefill_len + args.decode_steps
  pages_per_seq = (T + 31) // 32
  cfg = pyconfig.initialize(
      ["bench", os.path.join(pyconfig.PKG_DIR, "configs", "base.yml")], model_name=args.model, attention="paged", scan_layers=False,
      per_device_batch_size=float(B), max_prefill_predict_length=P, max_target_length=T, pagedattn_tokens_per_page=32,
      pagedattn_num_pages=B * pages_per_seq + 1, ixt_use_hip_graph=not args.no_graph, run_name="bench")
  eng = maxengine.MaxEngine(cfg)
  weights = gpu_weights(cfg, device, seed=0)
  params = eng.load_params(params=weights)
  # distinct utterances per rank and slot: seeds 1234 + global utterance index
  prompts = [synth.synth_prompt(cfg.vocab_size, P, 1234 + rank * B + s) for s in range(B)]
  prompts_dev = [(torch.from_numpy(t).to(device), tl) for t, tl in prompts]

  def one_pass(collect=False):
    prefixes = []
    for s, (t, tl) in enumerate(prompts_dev):
      pre, _ = eng.prefill(params=params, padded_tokens=t, true_length=tl, slot=s)
      prefixes.append(pre)
    ds = eng.init_decode_state()
    for s in range(B):
      ds = eng.insert(prefixes[s], ds, s)
    outs = []
    for _ in range(args.decode_steps):
      ds, res = eng.generate(params, ds)
      if collect:
        outs.append(res.data)
    for s in range(B):
      eng.release_pages(s)
    return outs

  def sync_all():
    torch.cuda.synchronize()
    if world > 1:
      import torch.distributed as dist
      dist.barrier()
      torch.cuda.synchronize()

  for _ in range(args.warmup):
    one_pass()
  sync_all()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    one_pass()
  sync_all()
  elapsed = time.perf_counter() - t0
  if world > 1:
    import torch.distributed as dist
    te = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(te, op=dist.ReduceOp.MAX)
    elapsed = float(te.item())

  tokens_per_pass = B * (args.decode_steps + 1)
  total_tokens = tokens_per_pass * args.steps * world
  value = total_tokens / elapsed
  nums = model_numbers(cfg)

  # ---- separate timing of prefill-only and decode-only (rank 0 reporting detail)
  detail = {}
  if rank == 0:
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    prefixes = []
    for s, (t, tl) in enumerate(prompts_dev):
      pre, _ = eng.prefill(params=params, padded_tokens=t, true_length=tl, slot=s)
      prefixes.append(pre)
    torch.cuda.synchronize()
    t_prefill = time.perf_counter() - t1
    ds = eng.init_decode_state()
    for s in range(B):
      ds = eng.insert(prefixes[s], ds, s)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(args.decode_steps):
      ds, res = eng.generate(params, ds)
    torch.cuda.synchronize()
    t_decode = time.perf_counter() - t2
    for s in range(B):
      eng.release_pages(s)
    ms_step = t_decode / args.decode_steps * 1e3
    mean_ctx = P + args.decode_steps / 2.0
    step_bytes = (nums["body_params"] + nums["unembed_params"]) * 2 + B * mean_ctx * nums["kv_bytes_per_token"] + B * nums["kv_bytes_per_token"]
    step_flops = B * 2.0 * (nums["body_params"] + nums["unembed_params"]) + B * mean_ctx * nums["kv_bytes_per_token"]
    detail = {
        "prefill_ms_per_utterance": t_prefill / B * 1e3,
        "decode_ms_per_step": ms_step,
        "decode_tokens_per_s": B / (ms_step * 1e-3),
        "decode_step_algorithmic_bytes": step_bytes,
        "decode_hbm_frac": step_bytes / (ms_step * 1e-3) / HBM_PEAK,
        "decode_mfu": step_flops / (ms_step * 1e-3) / MFMA_BF16_PEAK,
    }

  # ---- roofline of the dominant kernel, measured live with HIP events on the launch stream (eager launches of 16 decode steps).
  # Two kernel families carry (almost) every HBM byte of a decode step: the one-launch decode attention (K/V pages) and the
  # canonical dot (weights).  The one with the larger share of the step is reported; the other rides along in "other".
  roofline = None
  if rank == 0 and not args.no_roofline:
    lib = eng.lib
    import ctypes
    prefixes = []
    for s, (t, tl) in enumerate(prompts_dev):
      pre, _ = eng.prefill(params=params, padded_tokens=t, true_length=tl, slot=s)
      prefixes.append(pre)
    ds = eng.init_decode_state()
    for s in range(B):
      ds = eng.insert(prefixes[s], ds, s)
    # advance to the middle of the decode range (graph replay), so that the measured contexts are the average ones
    mid = max(0, args.decode_steps // 2 - 8)
    for _ in range(mid):
      ds, res = eng.generate(params, ds)
    torch.cuda.synchronize()
    lib.ixt_timing_enable(eng._ctx, 1)
    lib.ixt_timing_reset(eng._ctx)
    nmeas = 16
    for _ in range(nmeas):
      ds, res = eng.generate(params, ds)

    def fam(name):
      avg_us, n, bpl = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
      lib.ixt_timing_get(eng._ctx, name, ctypes.byref(avg_us), ctypes.byref(n), ctypes.byref(bpl))
      return avg_us.value, int(n.value), bpl.value
    dot_us, dot_n, dot_bpl = fam(b"dot_decode")
    att_us, att_n, _ = fam(b"attn_decode")
    lib.ixt_timing_enable(eng._ctx, 0)
    for s in range(B):
      eng.release_pages(s)
    # contexts during the measured steps: every utterance has true_length tl_s + mid + i + 1 keys at measured step i
    ctx_sum = sum(tl + mid + (nmeas + 1) / 2.0 for (_, tl) in prompts)
    kv_layer = nums["kv_bytes_per_token"] / cfg.num_decoder_layers                 # K+V bytes per token per layer (8 192 at C2)
    att_bpl = ctx_sum * kv_layer + B * kv_layer                                    # read every cached key/value once + write the new one
    traffic, traffic_src = None, None
    for pmc in (os.path.join(ROOT, "profiles", "r01_pmc_traffic.json"), os.path.join(ROOT, "profiles", f"r01_pmc_traffic_b{B}.json")):
      if not os.path.exists(pmc):
        continue
      try:
        pj = json.load(open(pmc))
        if pj.get("batch") == B and pj.get("model") == args.model:
          traffic, traffic_src = pj["hbm_bytes_per_launch"], pj.get("source")
      except Exception:  # pylint: disable=broad-except
        pass
    dot = {"kernel": "k_gemm / k_dotb / k_unembed (canonical dot on v_mfma_f32_16x16x32_bf16), all launches of a decode step", "bound": "hbm",
           "achieved": dot_bpl / (dot_us * 1e-6) / 1e9 if dot_us > 0 else 0.0, "unit": "GB/s", "avg_launch_us": dot_us, "launches": dot_n,
           "algorithmic_bytes_per_launch": dot_bpl, "us_per_step": dot_us * dot_n / nmeas}
    att = {"kernel": "k_attn_decode (one launch per layer: QKV finalize, scores, softmax, PV over the paged KV cache)", "bound": "hbm",
           "achieved": att_bpl / (att_us * 1e-6) / 1e9 if att_us > 0 else 0.0, "unit": "GB/s", "avg_launch_us": att_us, "launches": att_n,
           "algorithmic_bytes_per_launch": att_bpl, "us_per_step": att_us * att_n / nmeas,
           "mean_context": ctx_sum / B}
    for d in (dot, att):
      d["peak"] = HBM_PEAK / 1e9
      d["frac"] = d["achieved"] / d["peak"]
    main_k, other = (att, dot) if att["us_per_step"] >= dot["us_per_step"] else (dot, att)
    roofline = dict(main_k)
    roofline["traffic"] = traffic
    roofline["traffic_source"] = traffic_src
    roofline["other"] = other

  # ---- CPU baseline: the oracle (a port restating the reference algorithm) on a bounded sample, rank 0, N=1 only
  cpu_baseline = None
  if rank == 0 and world == 1 and not args.no_cpu_baseline:
    from oracle import oracle as O
    O.lib().orc_set_num_threads(min(16, os.cpu_count() or 1))   # the GPU box's CPU share for one GPU
    host_w = {k: v.detach().contiguous().view(torch.int16).cpu().numpy().view(np.uint16) for k, v in weights.items()}
    ocfg = O.OracleConfig(vocab_size=cfg.vocab_size, emb_dim=cfg.emb_dim, num_query_heads=cfg.num_query_heads, num_kv_heads=cfg.num_kv_heads,
                          head_dim=cfg.head_dim, mlp_dim=cfg.mlp_dim, num_decoder_layers=cfg.num_decoder_layers,
                          normalization_layer_epsilon=cfg.normalization_layer_epsilon, max_target_length=T, max_prefill_predict_length=P,
                          pagedattn_num_pages=pages_per_seq + 1, pagedattn_tokens_per_page=32, max_slots=1)
    orc = O.OracleEngine(O.OracleModel(ocfg, host_w))
    toks, tl = prompts[0]
    c0 = time.perf_counter()
    opre, ores = orc.prefill(toks, tl, slot=0)
    c1 = time.perf_counter()
    ods = orc.insert(opre, orc.init_decode_state(), 0)
    otoks = [int(ores[0, 0])]
    for _ in range(args.cpu_decode_steps):
      ods, orr = orc.generate(ods)
      otoks.append(int(orr[0, 0]))
    c2 = time.perf_counter()
    t_pre, t_step = c1 - c0, (c2 - c1) / max(1, args.cpu_decode_steps)
    cpu_tps = (args.decode_steps + 1) / (t_pre + args.decode_steps * t_step)
    # the same utterance on the GPU must give the same tokens (checker, not the thing measured)
    pre, r0 = eng.prefill(params=params, padded_tokens=prompts_dev[0][0], true_length=tl, slot=0)
    ds = eng.insert(pre, eng.init_decode_state(), 0)
    gtoks = [int(r0.data[0, 0])]
    for _ in range(args.cpu_decode_steps):
      ds, r = eng.generate(params, ds)
      gtoks.append(int(r.data[0, 0]))
    eng.release_pages(0)
    cpu_baseline = {"value": cpu_tps, "unit": "tokens/s", "cores": O.lib().orc_num_threads(), "kind": "port",
                    "sample": f"oracle (C port of the reference algorithm; JAX unavailable): 1 utterance, {tl}-token prefill ({t_pre:.2f} s) + "
                              f"{args.cpu_decode_steps} decode steps at ctx {tl}.. ({t_step:.3f} s/step), extrapolated to {args.decode_steps} steps at that step time",
                    "host_cpus": os.cpu_count(), "tokens_match_gpu": otoks == gtoks}

  if rank == 0:
    out = {
        "metric": "audio-tokens/sec (prefill+decode) @ 1B IndexTTS2", "value": value, "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16 (bf16 matrix-core dots, f32 accumulate; f32 fma chains in attention/norms)", "data": "synthetic",
        "config": {"workload": f"C2 {args.model}: {P}-token prefill -> {args.decode_steps} greedy audio-token decode steps, {B} utterances/GPU, paged KV (32 tok/page)",
                   "utterances_per_gpu": B, "prefill_len": P, "decode_steps": args.decode_steps, "parallelism": f"dp{world}", "hip_graph": not args.no_graph},
        "per_gpu_tokens_per_s": value / world,
        "detail": detail, "roofline": roofline, "cpu_baseline": cpu_baseline,
    }
    print(json.dumps(out), flush=True)
  if world > 1:
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
