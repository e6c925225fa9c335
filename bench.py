#!/usr/bin/env python3
"""Headline benchmark: audio tokens / second over prefill + decode of the 1B IndexTTS2-style decoder (BASELINE.json
configs[1], "C2"), data-parallel over N GPUs of one node (one process per GPU, no collective on the data path).

  python bench.py --gpus N --steps K --warmup W            (N > 1: this process starts N rank processes itself, before any GPU call)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic input: `--batch` (default 256) utterances per GPU,
each a 256-token prompt (4 s reference speech + 128-char text) prefilled -- `--prefill-pack` (default 8)
prompts per pass over the weights through the engine's prefill_concat / insert_partial (maxengine.py:719-836, 1206-1318) -- then
`--decode-steps` (default 1024) greedy generate calls for all slots, then ONE fetch of all generated ids to the host (decode.py:162-184
reads them; SURVEY.md section 8e "results gathered on host as int32 ids").  Tokens counted = generated audio tokens
including the first one from prefill (decode.py:162-178: T-P generate calls yield T-P+1 tokens per utterance).
Protocol = the reference's micro-benchmark (inference_microbenchmark.py:149-197): warm-up passes, then a timed loop closed by a device sync.

The driver's command is `python3 bench.py --gpus 1 --steps 20 --warmup 5` under a 600 s limit: 25 passes of the default batch take about
175 s.  Everything after the timed region runs against a wall-clock budget (`--budget-s`, counted from process start): `roofline` and
`cpu_baseline` always, the extra rows under `detail` (one / eight utterances, the 4096-token chunked prefill, one pass each at 512 and 1024
utterances per GPU) only while the budget allows; rows left out are named in `detail.skipped`.
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

_T0 = time.time()


def _log(msg):
  """Progress on stderr (stdout carries the one JSON line)."""
  print(f"[bench +{time.time() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


HBM_PEAK = 8.0e12          # B/s, MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s spec"
MFMA_BF16_PEAK = 2.5e15    # FLOP/s dense bf16 (same guide)


def parse(argv=None):
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=2)
  ap.add_argument("--warmup", type=int, default=1)
  ap.add_argument("--batch", type=int, default=256, help="utterances (decode slots) per GPU; 1024 (172 GB of paged K/V at 1280 tokens each) is the engine's maximum")
  ap.add_argument("--budget-s", type=float, default=420.0, help="wall-clock budget of the whole process; optional detail rows are skipped once it would be exceeded")
  ap.add_argument("--big-rows", default="1024,512", help="utterances per GPU of the extra one-pass rows under detail.big_batch_rows (budget permitting)")
  ap.add_argument("--decode-steps", type=int, default=1024)
  ap.add_argument("--prefill-len", type=int, default=256)
  ap.add_argument("--prefill-pack", type=int, default=8, help="prompts per packed prefill (prefill_concat); 1 = one prefill call per utterance")
  ap.add_argument("--model", default="indextts2-1b")
  ap.add_argument("--no-graph", action="store_true")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--no-roofline", action="store_true")
  ap.add_argument("--no-batch-rows", action="store_true", help="skip the B=1 / B=8 rows of SURVEY.md section 8(d)")
  ap.add_argument("--no-detail", action="store_true", help="no prefill / decode split events inside the timed passes (profiling runs)")
  ap.add_argument("--no-big-rows", action="store_true", help="skip the one-pass rows at larger batches")
  ap.add_argument("--no-fetch", action="store_true", help="do not fetch the generated ids to the host at the end of a pass (counter runs)")
  ap.add_argument("--cpu-batch", type=int, default=16, help="utterances of the CPU baseline sample")
  ap.add_argument("--cpu-slice-steps", type=int, default=2)
  ap.add_argument("--cpu-check-steps", type=int, default=4, help="decode steps of utterance 0 the parity oracle replays to check the fetched ids")
  ap.add_argument("--cpu-threads", type=int, default=16, help="threads of the CPU baseline (the GPU box's CPU share for one GPU is 16)")
  ap.add_argument("--sync-every", type=int, default=0, help="synchronise the stream every N engine calls (profiling runs: bounds the launch backlog)")
  ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for the CPU rehearsal)")
  ap.add_argument("--opt", action="append", default=[], help="library option name=value (ixt_set_option), e.g. kv_head_major=1; experiments")
  ap.add_argument("--stub-engine", action="store_true", help="CPU rehearsal of the multi-process plumbing (tests/test_dp_cpu.py): no GPU, no model")
  return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------------------- model numbers
def model_numbers(cfg):
  E, Q, KV, D, F, V, L = (cfg.emb_dim, cfg.num_query_heads, cfg.num_kv_heads, cfg.head_dim, cfg.mlp_dim, cfg.vocab_size, cfg.num_decoder_layers)
  body = L * (E * (Q * D + 2 * KV * D) + Q * D * E + 2 * E * F + F * E)
  return {"body_params": body, "unembed_params": E * V, "kv_bytes_per_token": 2 * KV * D * 2 * L}


def gpu_weights(cfg, device, seed=0):
  """Random-init weights of the architecture, drawn on the GPU (fast); same distributions as synth.synth_weights."""
  import torch
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


# ------------------------------------------------------------------------------------------------------------------- the workload
class Workload:
  """One GPU's share: B utterances through the engine's own prefill / insert / generate surface."""

  def __init__(self, eng, params, prompts_dev, decode_steps, pack, sync_every=0, sync=None, fetch=True, split_events=False):
    self.eng, self.params, self.prompts, self.decode_steps = eng, params, prompts_dev, decode_steps
    self.pack, self.sync_every, self.sync, self.calls = max(1, pack), sync_every, sync, 0
    self._packed = None
    self.fetch, self.split_events = fetch, split_events
    self.last_ids = None        # host int32 [B, decode_steps + 1] of the last pass (first token from prefill, then one per generate call)
    self.splits = []            # per pass (start, prefill done, decode done) events on the launch stream

  def _tick(self):
    self.calls += 1
    if self.sync_every and self.sync and self.calls % self.sync_every == 0:
      self.sync()

  def _packed_inputs(self):
    """Device arrays of each packed prefill, built once (they are inputs, like the prompts themselves)."""
    if self._packed is None:
      import numpy as np
      import torch
      self._packed = []
      B = len(self.prompts)
      for s0 in range(0, B, self.pack):
        group = self.prompts[s0:s0 + self.pack]
        lens = np.array([tl for _, tl in group], np.int32)
        dev = group[0][0].device
        toks = torch.cat([t[:tl] for t, tl in group])
        pos = torch.cat([torch.arange(tl, dtype=torch.int32, device=dev) for tl in lens])
        seg = torch.cat([torch.full((int(tl),), i + 1, dtype=torch.int32, device=dev) for i, tl in enumerate(lens)])
        starts = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int32)
        self._packed.append((toks, pos, seg, starts, lens, np.arange(s0, s0 + len(group))))
    return self._packed

  def prefill_all(self, ds):
    eng = self.eng
    if self.pack == 1:
      for s, (t, tl) in enumerate(self.prompts):
        pre, _ = eng.prefill(params=self.params, padded_tokens=t, true_length=tl, slot=s)
        ds = eng.insert(pre, ds, s)
        self._tick()
      return ds
    for toks, pos, seg, starts, lens, slots in self._packed_inputs():
      cache, pres, _ = eng.prefill_concat(params=self.params, padded_tokens=toks, decoder_positions=pos, decoder_segment_ids=seg, start_pos=starts,
                                          true_lengths=lens, num_prompts=len(lens))
      ds = eng.insert_partial(pres, ds, cache, slots, start_indices=starts, num_prompts=len(lens), seq_len=int(lens.max()))
      self._tick()
    return ds

  def decode_all(self, ds, steps=None, collect=None):
    for _ in range(self.decode_steps if steps is None else steps):
      ds, res = self.eng.generate(self.params, ds)
      if collect is not None:
        collect.append(res.data)
      self._tick()
    return ds

  def release_all(self):
    for s in range(len(self.prompts)):
      self.eng.release_pages(s)

  def _event(self):
    import torch
    ev = torch.cuda.Event(enable_timing=True)
    ev.record(torch.cuda.current_stream())     # the engine launches on torch's current stream (maxengine._stream_ptr)
    return ev

  def one_pass(self):
    """Prefill every utterance, decode, hand the ids to the host in ONE copy at the end (the only host synchronisation of a pass)."""
    ev = [self._event()] if self.split_events else None
    ds = self.prefill_all(self.eng.init_decode_state())
    if ev:
      ev.append(self._event())
    collect = None
    if self.fetch:
      collect = [ds["tokens"].clone()]           # [B, 1] first tokens (from prefill), in slot order
    self.decode_all(ds, collect=collect)
    if ev:
      ev.append(self._event())
      self.splits.append(ev)
    if self.fetch:
      import torch
      self.last_ids = torch.cat([collect[0]] + [c[:, 0:1] for c in collect[1:]], dim=1).cpu()
    self.release_all()

  def split_ms(self, last_n):
    """(prefill ms, decode ms) means over the last `last_n` passes, from the events recorded inside them."""
    rows = self.splits[-last_n:]
    if not rows:
      return None
    pre = sum(a.elapsed_time(b) for a, b, _ in rows) / len(rows)
    dec = sum(b.elapsed_time(c) for _, b, c in rows) / len(rows)
    return pre, dec


class _StubEngine:
  """CPU stand-in with the engine's call surface (tests/test_dp_cpu.py: the multi-process plumbing of this file without a GPU)."""

  def __init__(self, batch):
    import torch
    self.B, self.torch = batch, torch

  def init_decode_state(self):
    return {"tokens": self.torch.zeros((self.B, 1), dtype=self.torch.int32)}

  def prefill(self, *, params, padded_tokens, true_length, slot):
    return {"tokens": self.torch.tensor([[int(padded_tokens[0]) % 97]])}, None

  def insert(self, pre, ds, slot):
    ds["tokens"][slot] = pre["tokens"][0]
    return ds

  def generate(self, params, ds):
    time.sleep(0.0005)
    ds["tokens"] = (ds["tokens"] * 31 + 7) % 1009

    class R:
      data = self.torch.cat([ds["tokens"], self.torch.ones_like(ds["tokens"]), self.torch.zeros_like(ds["tokens"])], dim=1)
    return ds, R

  def release_pages(self, slot):
    pass


# ------------------------------------------------------------------------------------------------------------------- one rank
def make_engine(args, pyconfig, maxengine, B, device_weights=None):
  P, T = args.prefill_len, args.prefill_len + args.decode_steps
  pages_per_seq = (T + 31) // 32
  pack = max(1, min(args.prefill_pack, B))
  cfg = pyconfig.initialize(
      ["bench", os.path.join(pyconfig.PKG_DIR, "configs", "base.yml")], model_name=args.model, attention="paged", scan_layers=False,
      per_device_batch_size=float(B), max_prefill_predict_length=P, max_target_length=T, pagedattn_tokens_per_page=32,
      pagedattn_num_pages=(B + (pack if pack > 1 else 0)) * pages_per_seq + 1, ixt_use_hip_graph=not args.no_graph, run_name="bench",
      ixt_staging_groups=pack if pack > 1 else 0, ixt_max_packed_rows=pack * P if pack > 1 else 0)
  eng = maxengine.MaxEngine(cfg)
  lib = importlib.import_module("maxtext-indextts2_amd._lib")
  for kv in args.opt:
    k, v = kv.split("=")
    lib.check(eng.lib.ixt_set_option(eng._ctx, k.encode(), int(v)))
  return cfg, eng, pack


def run_rank(args):
  import numpy as np
  import torch
  dp = importlib.import_module("maxtext-indextts2_amd.dp")
  rank, local_rank, world = dp.env_world()
  if world != args.gpus:
    raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch N ranks (python bench.py --gpus N does it itself)")
  stub = args.stub_engine
  device = torch.device("cpu") if stub else torch.device("cuda", local_rank)
  if not stub:
    torch.cuda.set_device(local_rank)
  if world > 1:
    dp.init(args.backend, None if stub else device)
  sync_dev = (lambda: None) if stub else torch.cuda.synchronize
  B, P = args.batch, args.prefill_len
  synth = importlib.import_module("maxtext-indextts2_amd.synth")

  if stub:
    cfg, eng, params, weights, pack = None, _StubEngine(B), None, None, 1
    prompts = [(np.full((P,), rank * B + s, np.int32), P) for s in range(B)]
    prompts_dev = [(torch.from_numpy(t), tl) for t, tl in prompts]
  else:
    pyconfig = importlib.import_module("maxtext-indextts2_amd.pyconfig")
    maxengine = importlib.import_module("maxtext-indextts2_amd.maxengine")
    cfg, eng, pack = make_engine(args, pyconfig, maxengine, B)
    weights = gpu_weights(cfg, device, seed=0)
    params = eng.load_params(params=weights)
    # distinct utterances per rank and slot: seeds 1234 + global utterance index
    prompts = [synth.synth_prompt(cfg.vocab_size, P, 1234 + rank * B + s) for s in range(B)]
    prompts_dev = [(torch.from_numpy(t).to(device), tl) for t, tl in prompts]
  wl = Workload(eng, params, prompts_dev, args.decode_steps, pack, args.sync_every, sync_dev, fetch=not args.no_fetch,
                split_events=not (stub or args.no_detail))

  _log(f"rank {rank}: engine and weights ready")
  # ---- the timed region: W warm-up passes, then exactly K passes between barriers; MAX over ranks (dp.timed_region)
  elapsed = dp.timed_region(wl.one_pass, args.steps, args.warmup, sync_dev, None if stub else device)
  tokens_per_pass = B * (args.decode_steps + 1)
  value = tokens_per_pass * args.steps * world / elapsed
  _log(f"rank {rank}: timed region done: {elapsed / args.steps * 1e3:.1f} ms per pass")
  if not stub and hasattr(eng, "exchange_status") and eng.exchange_status() != 0:
    raise RuntimeError("a bounded wait inside the split decode attention gave up during the timed passes: the run's tokens are not to be trusted")

  out = {"metric": "audio-tokens/sec (prefill+decode) @ 1B IndexTTS2", "value": value, "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
         "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
         "dtype": "bf16 (bf16 matrix-core dots, f32 accumulate; f32 fma chains in attention/norms)", "data": "synthetic",
         "config": {"workload": f"C2 {args.model}: {P}-token prefill -> {args.decode_steps} greedy audio-token decode steps, {B} utterances/GPU, "
                                f"paged KV (32 tok/page), {pack} prompts per packed prefill",
                    "utterances_per_gpu": B, "prefill_len": P, "decode_steps": args.decode_steps, "parallelism": f"dp{world}", "hip_graph": not args.no_graph,
                    "prefill_pack": pack},
         "per_gpu_tokens_per_s": value / world}
  if stub:
    if rank == 0:
      print(json.dumps(out), flush=True)
    dp.finish()
    return

  nums = model_numbers(cfg)
  if rank == 0:
    budget = Budget(args.budget_s)
    out["detail"] = {"skipped": budget.skipped, "budget_s": args.budget_s}
    sp = wl.split_ms(args.steps)
    if sp:
      out["detail"].update(detail_split(args, nums, B, P, sp[0] * 1e-3, sp[1] * 1e-3))
    gpu_ids = wl.last_ids
    def guarded(name, fn):
      """roofline / cpu_baseline: an exception becomes {"error": ...} under the key -- the headline line must print whatever happens here."""
      try:
        r = fn()
        _log(f"{name} done")
        return r
      except Exception as ex:  # pylint: disable=broad-except
        _log(f"{name} FAILED: {type(ex).__name__}: {ex}")
        return {"error": f"{type(ex).__name__}: {ex}"[:300]}
    out["roofline"] = None if args.no_roofline else guarded("live roofline", lambda: roofline(args, wl, eng, cfg, nums, prompts, B))
    out["cpu_baseline"] = None if (world > 1 or args.no_cpu_baseline) else guarded("cpu baseline", lambda: cpu_baseline(args, cfg, weights, prompts, gpu_ids))
    wl.eng = None
    eng.close()                            # the rows below build their own engines: give the page pool back first
    del wl, eng, params
    torch.cuda.empty_cache()

    def optional(name, estimate_s, fn):
      """One optional row: skipped when the budget does not cover its estimate; a failure is recorded, never raised (the line must print)."""
      if not budget.allows(name, estimate_s):
        return None
      try:
        r = fn()
        _log(f"{name} done")
        return r
      except Exception as ex:  # pylint: disable=broad-except
        budget.skipped.append({"row": name, "error": f"{type(ex).__name__}: {ex}"[:300]})
        _log(f"{name} FAILED: {ex}")
        torch.cuda.empty_cache()
        return None

    if world > 1:                            # the rows below describe ONE GPU: on a multi-rank run the other ranks would only wait for them
      args.no_batch_rows = args.no_big_rows = True
      budget.skipped.append({"row": "batch_rows, long_prefill, big_batch_rows", "reason": f"single-GPU rows are not run at n_gpus = {world}"})
    if not args.no_batch_rows:
      rows = [optional(f"batch_rows[{b}]", 12.0 + 3.2 * args.decode_steps / 1024, lambda b=b: batch_row(args, b, weights, prompts_dev, nums)) for b in (1, 8)]
      out["detail"]["batch_rows"] = [r for r in rows if r is not None]
      for key, chunk in (("long_prefill", None), ("long_prefill_chunk1024", 1024)):
        r = optional(key, 8.0, lambda chunk=chunk: long_prefill_row(args, weights, nums, chunk=chunk))
        if r is not None:
          out["detail"][key] = r
    if not args.no_big_rows:
      out["detail"]["big_batch_rows"] = []
      for b in [int(v) for v in args.big_rows.split(",") if v]:
        if b == B:
          continue
        # one pass at b utterances: decode time grows a little faster than linearly below the attention-bound range; 1.5x margin + set-up
        est = 15.0 + 1.5 * (out["ms_per_step"] * 1e-3) * b / B
        r = optional(f"big_batch_rows[{b}]", est, lambda b=b: big_row(args, b, weights, cfg, nums, rank))
        if r is not None:
          out["detail"]["big_batch_rows"].append(r)
    out["wall_s"] = time.time() - _T0
    print(json.dumps(out), flush=True)
  dp.finish()


class Budget:
  """Wall-clock budget of the optional rows: a row runs only if its estimated cost still fits; what was left out is listed in the JSON."""

  def __init__(self, total_s):
    self.total, self.skipped = total_s, []

  def left(self):
    return self.total - (time.time() - _T0)

  def allows(self, name, estimate_s):
    if self.left() >= estimate_s:
      return True
    self.skipped.append({"row": name, "estimate_s": round(estimate_s, 1), "left_s": round(self.left(), 1)})
    _log(f"skipping {name}: needs ~{estimate_s:.0f} s, {self.left():.0f} s of the budget left")
    return False


def detail_split(args, nums, B, P, t_prefill, t_decode):
  """Prefill / decode split of the TIMED passes (means of the events recorded inside them on the launch stream)."""
  ms_step = t_decode / args.decode_steps * 1e3
  mean_ctx = P + args.decode_steps / 2.0
  step_bytes = (nums["body_params"] + nums["unembed_params"]) * 2 + B * mean_ctx * nums["kv_bytes_per_token"] + B * nums["kv_bytes_per_token"]
  step_flops = B * 2.0 * (nums["body_params"] + nums["unembed_params"]) + B * mean_ctx * nums["kv_bytes_per_token"]
  prefill_flops = 2.0 * P * nums["body_params"] + 4.0 * 16 * 128 * 16 * P * P / 2 + 2.0 * nums["unembed_params"]   # last-row logits only
  return {"prefill_ms_per_utterance": t_prefill / B * 1e3, "prefill_mfu": prefill_flops / (t_prefill / B) / MFMA_BF16_PEAK,
          "decode_ms_per_step": ms_step, "decode_tokens_per_s": B / (ms_step * 1e-3),
          "decode_step_algorithmic_bytes": step_bytes, "decode_hbm_frac": step_bytes / (ms_step * 1e-3) / HBM_PEAK,
          "decode_mfu": step_flops / (ms_step * 1e-3) / MFMA_BF16_PEAK, "source": "HIP events inside the timed passes"}


def big_row(args, b, weights, cfg0, nums, rank):
  """ONE full pass (after a short warm-up that captures the step graph) at `b` utterances per GPU: the same workload at a larger batch,
  where the decode attention reaches the read ceiling of its access pattern.  Reported beside the headline, never as `value`."""
  import torch
  pyconfig = importlib.import_module("maxtext-indextts2_amd.pyconfig")
  maxengine = importlib.import_module("maxtext-indextts2_amd.maxengine")
  synth = importlib.import_module("maxtext-indextts2_amd.synth")
  sub = argparse.Namespace(**vars(args))
  sub.prefill_pack = 16
  cfg, eng, pack = make_engine(sub, pyconfig, maxengine, b)
  params = eng.load_params(params=weights)
  dev = eng.device
  prompts_dev = [(torch.from_numpy(t).to(dev), tl) for t, tl in (synth.synth_prompt(cfg.vocab_size, args.prefill_len, 1234 + rank * b + s) for s in range(b))]
  wl = Workload(eng, params, prompts_dev, args.decode_steps, pack, split_events=True)
  ds = wl.prefill_all(eng.init_decode_state())          # warm-up: every packed prefill once, 8 decode steps (graph capture), release
  wl.decode_all(ds, steps=8)
  wl.release_all()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  wl.one_pass()
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  pre_ms, dec_ms = wl.split_ms(1)
  wl.eng = None
  eng.close()
  del wl, eng
  torch.cuda.empty_cache()
  return {"utterances": b, "prefill_pack": pack, "passes": 1, "seconds_per_pass": dt, "tokens_per_s": b * (args.decode_steps + 1) / dt,
          "prefill_ms_per_utterance": pre_ms / b, "decode_ms_per_step": dec_ms / args.decode_steps}


def roofline(args, wl, eng, cfg, nums, prompts, B):
  """Roofline of the dominant kernel, measured live with HIP events on the launch stream (eager launches of 16 decode steps in the middle
  of the decode range).  Two kernel families carry (almost) every HBM byte of a decode step: the one-launch decode attention (K/V pages)
  and the canonical dot (weights).  The one with the larger share of the step is reported; the other rides along in "other"."""
  import ctypes
  import torch
  lib = eng.lib
  ds = wl.prefill_all(eng.init_decode_state())
  mid = max(0, args.decode_steps // 2 - 8)
  ds = wl.decode_all(ds, steps=mid)
  torch.cuda.synchronize()
  lib.ixt_timing_enable(eng._ctx, 1)
  lib.ixt_timing_reset(eng._ctx)
  nmeas = 16
  ds = wl.decode_all(ds, steps=nmeas)

  def fam(name):
    avg_us, n, bpl = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
    lib.ixt_timing_get(eng._ctx, name, ctypes.byref(avg_us), ctypes.byref(n), ctypes.byref(bpl))
    return avg_us.value, int(n.value), bpl.value
  dot_us, dot_n, dot_bpl = fam(b"dot_decode")
  att_us, att_n, _ = fam(b"attn_decode")
  lib.ixt_timing_enable(eng._ctx, 0)
  wl.release_all()
  # contexts during the measured steps: every utterance has true_length tl_s + mid + i + 1 keys at measured step i
  ctx_sum = sum(tl + mid + (nmeas + 1) / 2.0 for (_, tl) in prompts)
  kv_layer = nums["kv_bytes_per_token"] / cfg.num_decoder_layers                 # K+V bytes per token per layer (8 192 at C2)
  att_bpl = ctx_sum * kv_layer + B * kv_layer                                    # read every cached key/value once + the new one
  build_id = lib.ixt_build_id().decode()
  traffic, traffic_src, mfma, stale = None, None, None, None
  import glob
  for pmc in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc*.json")), reverse=True):      # newest round first
    try:
      pj = json.load(open(pmc))
    except Exception:  # pylint: disable=broad-except
      continue
    if pj.get("batch") != B or pj.get("model") != args.model:
      continue
    if pj.get("build_id") == build_id:      # counters of THIS build only
      traffic, mfma = pj.get("attn_hbm_bytes_per_launch"), pj.get("mfma")
      traffic_src = (f"{os.path.relpath(pmc, ROOT)} -- NOT this run: separate rocprofv3 --pmc passes of the same build on the workload "
                     f"'{pj.get('workload', pj.get('source'))}'; " + str(pj.get("source")))
      stale = None
      break
    stale = stale or f"{os.path.relpath(pmc, ROOT)} was measured on build {pj.get('build_id')}, this is {build_id}"
  dot = {"kernel": "k_gemm_ring / k_gemm / k_dotb / k_unembed (canonical dot on v_mfma_f32_16x16x32_bf16), all launches of a decode step", "bound": "hbm",
         "achieved": dot_bpl / (dot_us * 1e-6) / 1e9 if dot_us > 0 else 0.0, "unit": "GB/s", "avg_launch_us": dot_us, "launches": dot_n,
         "algorithmic_bytes_per_launch": dot_bpl, "us_per_step": dot_us * dot_n / nmeas}
  att = {"kernel": "k_attn_decode (one launch per layer: scores, softmax, PV over the paged KV cache)", "bound": "hbm",
         "achieved": att_bpl / (att_us * 1e-6) / 1e9 if att_us > 0 else 0.0, "unit": "GB/s", "avg_launch_us": att_us, "launches": att_n,
         "algorithmic_bytes_per_launch": att_bpl, "us_per_step": att_us * att_n / nmeas, "mean_context": ctx_sum / B}
  for d in (dot, att):
    d["peak"] = HBM_PEAK / 1e9
    d["frac"] = d["achieved"] / d["peak"]
  main_k, other = (att, dot) if att["us_per_step"] >= dot["us_per_step"] else (dot, att)
  r = dict(main_k)
  r["traffic"], r["traffic_source"], r["build_id"] = traffic, traffic_src, build_id
  if stale:
    r["traffic_stale"] = stale
  if mfma:
    r["mfma_utilisation"] = mfma
  r["other"] = other
  return r


def batch_row(args, b, weights, prompts_dev, nums):
  """the same model at B = 1 (latency) and B = 8 utterances per GPU against the HBM bounds of BASELINE.md section 4."""
  import torch
  pyconfig = importlib.import_module("maxtext-indextts2_amd.pyconfig")
  maxengine = importlib.import_module("maxtext-indextts2_amd.maxengine")
  sub = argparse.Namespace(**vars(args))
  sub.prefill_pack = 1
  cfg, eng, _ = make_engine(sub, pyconfig, maxengine, b)
  params = eng.load_params(params=weights)
  wl = Workload(eng, params, prompts_dev[:b], args.decode_steps, 1)
  wl.one_pass()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  ds = wl.prefill_all(eng.init_decode_state())
  torch.cuda.synchronize()
  t1 = time.perf_counter()
  wl.decode_all(ds)
  torch.cuda.synchronize()
  t2 = time.perf_counter()
  wl.release_all()
  wl.eng = None
  eng.close()
  del eng
  ms_step = (t2 - t1) / args.decode_steps * 1e3
  mean_ctx = args.prefill_len + args.decode_steps / 2.0
  step_bytes = (nums["body_params"] + nums["unembed_params"]) * 2 + b * (mean_ctx + 1) * nums["kv_bytes_per_token"]
  bound = b / (step_bytes / HBM_PEAK)
  return {"utterances": b, "tokens_per_s": b * (args.decode_steps + 1) / (t2 - t0), "decode_ms_per_step": ms_step, "decode_tokens_per_s": b / (ms_step * 1e-3),
          "prefill_ms_per_utterance": (t1 - t0) / b * 1e3, "decode_hbm_frac": step_bytes / (ms_step * 1e-3) / HBM_PEAK,
          "hbm_bound_tokens_per_s": bound}


def long_prefill_row(args, weights, nums, total=4096, chunk=None):
  """BASELINE.json configs[3]'s prefill shape on the C2 model: ONE prompt of `total` tokens through the engine's chunked prefill (chunks of
  `chunk` rows, default --prefill-len; each chunk attends to everything before it through the pages: k_attn_tile with a growing key range)."""
  import numpy as np
  import torch
  pyconfig = importlib.import_module("maxtext-indextts2_amd.pyconfig")
  maxengine = importlib.import_module("maxtext-indextts2_amd.maxengine")
  synth = importlib.import_module("maxtext-indextts2_amd.synth")
  P = chunk or args.prefill_len
  cfg = pyconfig.initialize(["bench", os.path.join(pyconfig.PKG_DIR, "configs", "base.yml")], model_name=args.model, attention="paged", scan_layers=False,
                            per_device_batch_size=1.0, max_prefill_predict_length=P, max_target_length=total + 32, pagedattn_tokens_per_page=32,
                            pagedattn_num_pages=(total + 32) // 32 + 2, ixt_use_hip_graph=False, run_name="bench", use_chunked_prefill=True,
                            prefill_chunk_size=P)
  eng = maxengine.MaxEngine(cfg)
  params = eng.load_params(params=weights)
  rng = np.random.default_rng(4096)
  toks = torch.from_numpy(rng.integers(1, cfg.vocab_size, size=total).astype(np.int32)).to(eng.device)

  def run():
    done, pre = 0, None
    while done < total:
      n = min(P, total - done)
      ep = None if done == 0 else maxengine.ExistingPrefix(cache=pre["cache"], common_prefix_tokens=toks[:done])
      pre, _ = eng.prefill(params=params, padded_tokens=toks[done:done + n], true_length=n, slot=0, existing_prefix=ep)
      done += n
    torch.cuda.synchronize()
    eng.release_pages(0)
  run()
  t0 = time.perf_counter()
  run()
  dt = time.perf_counter() - t0
  eng.close()
  del eng
  flops = 2.0 * total * nums["body_params"] + 4.0 * 16 * 128 * 16 * total * total / 2 + (total // P) * 2.0 * nums["unembed_params"]
  return {"prompt_tokens": total, "chunk": P, "seconds": dt, "prefill_tokens_per_s": total / dt, "mfu": flops / dt / MFMA_BF16_PEAK}


def cpu_baseline(args, cfg, weights, prompts, gpu_ids=None):
  """The reference ALGORITHM on this host's CPU cores (oracle/ref_algo_cpu.py: dense two-segment cache, both full buffers read at every step,
  [1, P, V] prefill logits, bf16 values / f32 accumulation), on a bounded sample of the workload.  JAX (the north star's `jax.devices('cpu')`)
  is not installed anywhere in this pipeline; this torch-CPU restatement is the labelled stand-in of BASELINE.md section 3.
  `gpu_ids` (the ids the last timed pass fetched) are CHECKED here against the parity oracle (oracle/ixt_oracle.c: utterance 0, the first
  1 + `--cpu-check-steps` ids, bit-exact contract) and against the independent f32 restatement's first token."""
  import numpy as np
  import torch
  from oracle import ref_algo_cpu as RA
  host_w = {k: v.detach().contiguous().view(torch.int16).cpu().numpy().view(np.uint16) for k, v in weights.items()}
  cb = max(1, min(args.cpu_batch, args.batch))
  # the GPU box gives one GPU's job 16 CPUs whatever os.cpu_count() says: more threads than that only thrash
  threads = min(args.cpu_threads, os.cpu_count() or 1)
  _log(f"cpu baseline: {cb} utterances on {threads} threads")
  r = RA.measure(cfg, host_w, prompts, cb, args.prefill_len, args.decode_steps, slice_steps=args.cpu_slice_steps, threads=threads, log=_log)
  model = ""
  try:
    model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo", encoding="utf-8") if l.startswith("model name")][0]
  except Exception:  # pylint: disable=broad-except
    pass
  r.update({"kind": "port", "host_cpus": os.cpu_count(), "cpu_model": model,
            "sample": f"torch-cpu restatement of the reference algorithm (JAX unavailable), {r['matmul']} matmul, {r['cores']} threads: {cb} utterances, one "
                      f"{args.prefill_len}-token prefill timed ({r['prefill_s_per_utterance']:.2f} s, charged {cb} times) + {args.cpu_slice_steps}-step decode slices at "
                      f"three ring positions ({', '.join(f'{s:.3f}' for s in r['decode_slices_s_per_step'])} s/step), integrated over {args.decode_steps} steps"})
  if gpu_ids is not None:
    n = max(0, min(args.cpu_check_steps, args.decode_steps))
    from oracle import oracle as O
    O.lib().orc_set_num_threads(threads)
    T = args.prefill_len + args.decode_steps
    ocfg = O.OracleConfig(vocab_size=cfg.vocab_size, emb_dim=cfg.emb_dim, num_query_heads=cfg.num_query_heads, num_kv_heads=cfg.num_kv_heads,
                          head_dim=cfg.head_dim, mlp_dim=cfg.mlp_dim, num_decoder_layers=cfg.num_decoder_layers,
                          normalization_layer_epsilon=cfg.normalization_layer_epsilon, max_target_length=T, max_prefill_predict_length=args.prefill_len,
                          pagedattn_num_pages=(T + 31) // 32 + 1, pagedattn_tokens_per_page=32, max_slots=1)
    orc = O.OracleEngine(O.OracleModel(ocfg, host_w))
    toks, tl = prompts[0]
    opre, ores = orc.prefill(toks, tl, slot=0)
    ods = orc.insert(opre, orc.init_decode_state(), 0)
    otoks = [int(ores[0, 0])]
    for _ in range(n):
      ods, orr = orc.generate(ods)
      otoks.append(int(orr[0, 0]))
    gtoks = [int(v) for v in gpu_ids[0, :n + 1]]
    r.update({"tokens_match_gpu": otoks == gtoks, "tokens_compared": n + 1,
              "tokens_checker": "oracle/ixt_oracle.c (parity oracle, bit-exact contract) on utterance 0 against the ids the last timed pass fetched",
              "ref_algo_first_token_matches_gpu": int(r["first_token"]) == gtoks[0]})
    _log(f"token check: oracle {otoks} gpu {gtoks}")
  return r


def main():
  args = parse()
  if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
    # Start the N rank processes from here, before anything has touched the GPU (a process that has initialised HIP must never
    # be replaced or forked into ranks); the children are fresh interpreters running this file with the rank environment set.
    dp_spawn = importlib.import_module("maxtext-indextts2_amd.dp_spawn")
    sys.exit(dp_spawn.spawn_local_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus))
  run_rank(args)


if __name__ == "__main__":
  main()
