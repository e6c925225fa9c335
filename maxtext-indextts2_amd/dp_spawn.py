"""Start one rank process per GPU of this node from a plain `python bench.py --gpus N` (no torch import here: the parent must not
touch the GPU -- a process that has initialised HIP may neither be replaced by another program nor share its state with ranks).

Each child is a fresh interpreter running the same command line with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set,
exactly what `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` would give it.  Rank 0 inherits this process's stdout
(the one JSON line of the bench); the other ranks' stdout goes to stderr.  A rank that fails takes the others down with it (by PID)
and makes the exit code non-zero."""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
from typing import List


def free_port() -> int:
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  port = s.getsockname()[1]
  s.close()
  return port


def spawn_local_ranks(cmd: List[str], n: int, timeout_s: float = 0.0) -> int:
  port = free_port()
  procs = []
  for r in range(n):
    env = dict(os.environ)
    env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL needs it)
    procs.append(subprocess.Popen(cmd, env=env, stdout=None if r == 0 else sys.stderr))
  t0 = time.time()
  rc = 0
  alive = list(procs)
  while alive:
    for p in list(alive):
      code = p.poll()
      if code is None:
        continue
      alive.remove(p)
      if code != 0 and rc == 0:
        rc = code if code > 0 else 1
        for q in alive:          # one rank failed: the others would wait in a barrier forever
          q.terminate()
    if timeout_s and time.time() - t0 > timeout_s and alive:
      rc = rc or 124
      for q in alive:
        q.terminate()
    time.sleep(0.05)
  return rc
