"""Expert-parallel exchange of the Mixtral block: the one place of the inference path with a real exchange
step.  Experts are split over the ranks of a node (expert e lives on rank e // experts_per_rank); every rank routes its own rows
(layers/moe.py:420-453 `permute`), sends each (row, k) copy to the rank that owns its expert, receives the copies for its experts,
applies them, and sends the results back the way they came (moe.py:855-931: the reference does this with two `jax.lax.ragged_all_to_all`
calls whose offsets come from `get_all_to_all_params`, :571-658).  Here it is two `torch.distributed.all_to_all_single` calls with split
sizes -- RCCL over xGMI on a GPU node ("nccl" backend), gloo in the CPU tests; the sizes are exchanged first with a small all-to-all.  On
the 8 fully connected GPUs of an MI355X node an all-to-all keeps all 7 links of a GPU busy at once (each peer gets its own slice), which is
the collective shape xGMI wants; nothing here is a ring.

`get_all_to_all_params` and `permute` restate the reference's integer bookkeeping (same argument meaning and results; numpy instead of
jnp) and are pinned by the cases the reference's own tests hold (tests/moe_test.py:788-900), restated in tests/test_moe_ep_cpu.py.  The
expert MLP itself is the caller's function (on a GPU: the context's per-expert dots), so this module is plumbing: device memory in, device
memory out, no arithmetic beyond the exchange."""
from __future__ import annotations

from typing import Callable, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


def get_all_to_all_params(all_shards_group_sizes, shard_id: int, num_expert_parallelism: int, is_batch_sharded: bool = True):
  """layers/moe.py:571-658.  Batch sharded: `all_shards_group_sizes[i, j]` = rows batch shard i sends to expert shard j.  Batch replicated:
  a vector, entry i = rows expert shard i is responsible for.  Returns (input_offsets, send_sizes, output_offsets, recv_sizes)."""
  g = np.asarray(all_shards_group_sizes)
  if is_batch_sharded:
    local = g[shard_id]
    input_offsets = np.concatenate(([0], np.cumsum(local)[:-1])).astype(g.dtype)
    send_sizes = local.copy()
    above = np.concatenate((np.zeros((1,) + g.shape[1:], g.dtype), g), axis=0)
    output_offsets = np.cumsum(above, axis=0, dtype=g.dtype)[shard_id]
    recv_sizes = g[:, shard_id].copy()
    return input_offsets, send_sizes, output_offsets, recv_sizes
  input_offsets = np.zeros((num_expert_parallelism,), g.dtype)
  send_sizes = np.repeat(g[shard_id], num_expert_parallelism)
  start = np.concatenate(([0], np.cumsum(g[:-1])))[shard_id]
  output_offsets = np.repeat(start, num_expert_parallelism).astype(g.dtype)
  recv_sizes = g.copy()
  return input_offsets, send_sizes, output_offsets, recv_sizes


def permute(selected_experts: torch.Tensor, num_experts: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
  """layers/moe.py:420-453 without the data movement: selected_experts int [rows, k] -> (sorted_selected_experts = the stable argsort of
  the flattened ids, source row of every sorted copy = that // k, group_size[num_experts])."""
  flat = selected_experts.reshape(-1)
  order = torch.argsort(flat, stable=True)
  rows = torch.div(order, selected_experts.shape[1], rounding_mode="floor")
  group_size = torch.bincount(flat, minlength=num_experts)
  return order, rows, group_size


def expert_parallel_apply(x: torch.Tensor, selected_experts: torch.Tensor, num_experts: int,
                          expert_fn: Callable[[int, torch.Tensor], torch.Tensor], group=None, always_exchange: bool = False) -> torch.Tensor:
  """x [rows, emb] and selected_experts int [rows, k] of THIS rank -> y [rows, k, emb] with y[r, j] = expert_fn(selected_experts[r, j], x[r]),
  every expert evaluated on the rank that owns it (experts_per_rank = num_experts / world consecutive ids per rank).  expert_fn(e, rows)
  is called once per local expert that received rows, with the global expert id.  always_exchange: run the collectives on a one-rank
  group too (exercises the RCCL calls on a one-GPU box)."""
  world = dist.get_world_size(group) if dist.is_initialized() else 1
  rank = dist.get_rank(group) if dist.is_initialized() else 0
  if num_experts % world:
    raise ValueError(f"num_experts ({num_experts}) must be a multiple of the expert-parallel degree ({world})")
  per_rank = num_experts // world
  R, K = selected_experts.shape
  order, src_rows, group_size = permute(selected_experts, num_experts)
  sorted_x = x.index_select(0, src_rows)                               # copies grouped by expert = grouped by owning rank
  send_counts = group_size.reshape(world, per_rank)                    # [dest rank][its local expert]
  exchange = world > 1 or (always_exchange and dist.is_initialized())
  if not exchange:
    recv_counts = send_counts.clone()
    recv_x = sorted_x
  else:
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts, group=group)      # [src rank][my local expert]
    send_split = send_counts.sum(dim=1).tolist()
    recv_split = recv_counts.sum(dim=1).tolist()
    recv_x = x.new_empty((sum(recv_split), x.shape[1]))
    dist.all_to_all_single(recv_x, sorted_x, recv_split, send_split, group=group)
  # received rows arrive grouped by source rank, inside a source by local expert: walk the blocks, one expert_fn call per local expert
  counts = recv_counts.tolist()
  recv_y = torch.empty_like(recv_x)
  starts = []
  off = 0
  for s in range(world):
    row = []
    for le in range(per_rank):
      row.append(off)
      off += counts[s][le]
    starts.append(row)
  for le in range(per_rank):
    pieces = [(starts[s][le], counts[s][le]) for s in range(world) if counts[s][le]]
    if not pieces:
      continue
    idx = torch.cat([torch.arange(a, a + n, device=x.device) for a, n in pieces])
    recv_y.index_copy_(0, idx, expert_fn(rank * per_rank + le, recv_x.index_select(0, idx)))
  if not exchange:
    back = recv_y
  else:
    back = torch.empty_like(sorted_x)
    dist.all_to_all_single(back, recv_y, send_split, recv_split, group=group)
  y = torch.empty((R * K, x.shape[1]), dtype=x.dtype, device=x.device)
  y.index_copy_(0, order, back)                                        # moe.py:455-466: undo the sort
  return y.reshape(R, K, x.shape[1])


def combine(y: torch.Tensor, weights: torch.Tensor) -> torch.Tensor:
  """moe.py:467-484 `unpermute`: einsum("BKE,BK -> BE") in f32 over the k copies in top_k order, as one fma chain (the canonical order of
  the single-GPU kernel k_moe_combine); result in y's dtype."""
  t = torch.zeros((y.shape[0], y.shape[2]), dtype=torch.float64 if y.dtype == torch.float64 else torch.float32, device=y.device)
  for j in range(y.shape[1]):
    t = torch.addcmul(t, y[:, j].to(t.dtype), weights[:, j:j + 1].to(t.dtype))
  return t.to(y.dtype)
