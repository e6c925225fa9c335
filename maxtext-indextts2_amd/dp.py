"""Data-parallel plumbing for N GPUs of one node: one process per GPU, utterances sharded across ranks, NO collective
on the data path.  The only exchanges are host-side: a barrier around the timed region, a MAX
of the elapsed time, and a gather of the generated token ids.  Backend "nccl" is RCCL on ROCm; the CPU tests use gloo."""
from __future__ import annotations

import os
from typing import List, Sequence

import torch
import torch.distributed as dist


def env_world():
  """(rank, local_rank, world_size) from the torch.distributed.run environment."""
  return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend: str, device: torch.device | None = None):
  if not dist.is_initialized():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend == "nccl" and device is not None:
      dist.init_process_group(backend, device_id=device)
    else:
      dist.init_process_group(backend)


def shard_utterances(n_total: int, rank: int, world: int) -> List[int]:
  """Utterance u goes to rank u mod world (the assignment SURVEY.md section 8e names); returns this rank's global ids."""
  return [u for u in range(n_total) if u % world == rank]


def max_over_ranks(seconds: float, device: torch.device | None = None) -> float:
  if not dist.is_initialized() or dist.get_world_size() == 1:
    return seconds
  t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
  dist.all_reduce(t, op=dist.ReduceOp.MAX)
  return float(t.item())


def barrier():
  if dist.is_initialized() and dist.get_world_size() > 1:
    dist.barrier()


def timed_region(one_pass, steps: int, warmup: int, sync_device, device: torch.device | None = None) -> float:
  """The bench contract: `warmup` untimed passes, then exactly `steps` passes bracketed by barrier + device synchronise on both
  sides; returns the MAX over ranks of the elapsed seconds.  `sync_device` = torch.cuda.synchronize (or a no-op on CPU)."""
  import time  # pylint: disable=import-outside-toplevel
  for _ in range(warmup):
    one_pass()
  sync_device()
  barrier()
  sync_device()
  t0 = time.perf_counter()
  for _ in range(steps):
    one_pass()
  sync_device()
  barrier()
  sync_device()
  return max_over_ranks(time.perf_counter() - t0, device)


def finish():
  if dist.is_initialized():
    dist.barrier()
    dist.destroy_process_group()


def gather_token_ids(local_ids: Sequence[int], local_tokens: torch.Tensor, n_total: int) -> torch.Tensor | None:
  """local_tokens int32 [len(local_ids), T] -> on rank 0 an int32 [n_total, T] matrix in global utterance order."""
  if not dist.is_initialized() or dist.get_world_size() == 1:
    out = torch.zeros((n_total, local_tokens.shape[1]), dtype=torch.int32)
    for i, u in enumerate(local_ids):
      out[u] = local_tokens[i].cpu()
    return out
  payload = (list(local_ids), local_tokens.cpu())
  gathered = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
  dist.gather_object(payload, gathered, dst=0)
  if dist.get_rank() != 0:
    return None
  out = torch.zeros((n_total, local_tokens.shape[1]), dtype=torch.int32)
  for ids, toks in gathered:
    for i, u in enumerate(ids):
      out[u] = toks[i]
  return out
