"""N > 1 path on CPU: world_size-2 gloo processes exercise the data-parallel plumbing (sharding, barrier, MAX of the
elapsed time, gather of token ids) that bench.py uses with RCCL on the GPU node."""
import importlib
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _worker(rank, world, port, n_total, q):
  sys.path.insert(0, ROOT)
  os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
  dp = importlib.import_module("maxtext-indextts2_amd.dp")
  dp.init("gloo")
  ids = dp.shard_utterances(n_total, rank, world)
  # stand-in for the per-utterance decode: token t of utterance u is 1000*u + t
  toks = torch.tensor([[1000 * u + t for t in range(5)] for u in ids], dtype=torch.int32).reshape(len(ids), 5)
  dp.barrier()
  elapsed = dp.max_over_ranks(0.5 + rank)  # rank 1 is the slow one
  allt = dp.gather_token_ids(ids, toks, n_total)
  if rank == 0:
    q.put((ids, elapsed, allt.tolist()))
  dp.barrier()
  torch.distributed.destroy_process_group()


@pytest.mark.parametrize("n_total", [7, 16])
def test_world_size_2_gloo(n_total):
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
  for p in procs:
    p.start()
  ids0, elapsed, allt = q.get(timeout=120)
  for p in procs:
    p.join(timeout=120)
    assert p.exitcode == 0
  assert ids0 == list(range(0, n_total, 2))
  assert elapsed == 1.5  # MAX over ranks
  assert allt == [[1000 * u + t for t in range(5)] for u in range(n_total)]


def test_shard_covers_everything_once():
  dp = importlib.import_module("maxtext-indextts2_amd.dp")
  for world in (1, 2, 4, 8):
    seen = sorted(u for r in range(world) for u in dp.shard_utterances(64, r, world))
    assert seen == list(range(64))
    sizes = {len(dp.shard_utterances(64, r, world)) for r in range(world)}
    assert sizes == {64 // world}


def _bench_cmd(extra):
  return [sys.executable, os.path.join(ROOT, "bench.py"), "--stub-engine", "--backend", "gloo", "--batch", "6", "--decode-steps", "20", "--prefill-len", "8",
          "--steps", "2", "--warmup", "1"] + extra


def test_bench_spawns_its_own_ranks():
  """`python bench.py --gpus 2` (no torchrun): the parent starts two rank processes itself, they meet over gloo, run bench.py's own timed
  region (dp.timed_region: barrier + MAX over ranks) around the stub engine, and rank 0 prints ONE JSON line with n_gpus = 2 and the
  whole-job token count."""
  import json
  import subprocess
  env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
  r = subprocess.run(_bench_cmd(["--gpus", "2"]), capture_output=True, text=True, timeout=300, env=env)
  assert r.returncode == 0, r.stderr[-2000:]
  lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
  assert len(lines) == 1, r.stdout
  out = json.loads(lines[0])
  assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
  tokens = 2 * 2 * 6 * 21                                   # ranks x steps x utterances x (decode steps + 1)
  assert abs(out["value"] - tokens / (out["ms_per_step"] * 1e-3 * 2)) < 1e-6 * out["value"]
  assert out["ms_per_step"] >= 20 * 0.5                     # the stub sleeps 0.5 ms per generate call


def test_bench_refuses_mismatched_world():
  """--gpus must agree with the launcher's WORLD_SIZE (a plain `--gpus 8` under a 1-rank environment must not report one GPU's number
  as eight)."""
  import subprocess
  env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
  r = subprocess.run(_bench_cmd(["--gpus", "4"]), capture_output=True, text=True, timeout=120, env=env)
  assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_failed_rank_fails_the_launch():
  dp_spawn = importlib.import_module("maxtext-indextts2_amd.dp_spawn")
  code = "import os, sys, time; sys.exit(3) if os.environ['RANK'] == '1' else time.sleep(30)"
  rc = dp_spawn.spawn_local_ranks([sys.executable, "-c", code], 2)
  assert rc == 3
