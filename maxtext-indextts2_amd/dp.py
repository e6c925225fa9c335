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


def exchange_gradients(grads: torch.Tensor, loss_and_weights: torch.Tensor, algorithm: str = "auto", bucket_elems: int = 64 << 20) -> None:
  """The data-parallel step of train_step: every rank ends with the SUM over ranks of the flat gradient buffer and of
  [total_loss, total_weights] (GSPMD emits this all-reduce implicitly in the reference; loss = sum / sum as in
  train.py:307-322).  In place.  algorithm "rs_ag" (what "auto" picks): per bucket a reduce-scatter followed by an all-gather -- on the
  8 fully connected GPUs of a node every rank exchanges 1/8 of the bucket with each of its 7 peers at once, all xGMI links busy, where a
  ring all-reduce is bound by one link; "all_reduce": one all-reduce per bucket."""
  if not dist.is_initialized() or dist.get_world_size() == 1:
    return
  world = dist.get_world_size()
  dist.all_reduce(loss_and_weights, op=dist.ReduceOp.SUM)
  algo = "rs_ag" if algorithm == "auto" else algorithm
  n = grads.numel()
  step = max(world, (bucket_elems // world) * world)
  for b0 in range(0, n, step):
    bucket = grads[b0:min(n, b0 + step)]
    if algo == "all_reduce" or bucket.numel() % world:      # a ragged tail bucket: plain all-reduce
      dist.all_reduce(bucket, op=dist.ReduceOp.SUM)
      continue
    shard = torch.empty((bucket.numel() // world,), dtype=bucket.dtype, device=bucket.device)
    dist.reduce_scatter_tensor(shard, bucket, op=dist.ReduceOp.SUM)
    dist.all_gather_into_tensor(bucket, shard)


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
