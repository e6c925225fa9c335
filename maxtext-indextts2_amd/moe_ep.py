"""Expert-parallel exchange of the Mixtral block: the one place of the inference path with a real exchange
step.  Experts are split over the ranks of a node (expert e lives on rank e // experts_per_rank); every rank routes its own rows
(layers/moe.py:420-453 `permute`), sends each (row, k) copy to the rank that owns its expert, receives the copies for its experts,
applies them, and sends the results back the way they came (moe.py:855-931: the reference does this with two `jax.lax.ragged_all_to_all`
calls whose offsets come from `get_all_to_all_params`, :571-658).  Here it is two `torch.distributed.all_to_all_single` calls with split
sizes -- RCCL over xGMI on a GPU node ("nccl" backend), gloo in the CPU tests; the sizes are exchanged first with a small all-to-all.  On
the 8 fully connected GPUs of an MI355X node an all-to-all keeps all 7 links of a GPU busy at once (each peer gets its own slice), which is
the collective shape xGMI wants; nothing here is a ring.

`get_all_to_all_params`, `permute` and `local_permute` restate the reference's integer bookkeeping (same argument meaning and results; numpy instead of
jnp) and are pinned by the cases the reference's own tests hold (tests/moe_test.py:634-900), restated in tests/test_moe_ep_cpu.py.  The
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


def local_permute(inputs: torch.Tensor, global_group_sizes: torch.Tensor, local_expert_size: int, shard_index: int, is_offset: bool = False,
                  global_sorted_experts: torch.Tensor | None = None):
  """layers/moe.py:485-568: the rows an expert shard holds, grouped by its local expert id.  `global_group_sizes` [batch shards, num_experts] =
  copies every batch shard made for every expert.  is_offset=False: `inputs` are what the all-to-all delivered -- source shard by source
  shard, inside a source by local expert.  is_offset=True: `inputs` are all copies sorted by global expert id (`global_sorted_experts`); rows
  of other shards' experts sort behind this shard's.  Returns (sorted_inputs, sorted_indices, local_group_size, sorted_experts_ids)."""
  g = global_group_sizes.reshape(-1, global_group_sizes.shape[-1])
  local = g[:, shard_index * local_expert_size:(shard_index + 1) * local_expert_size]
  local_group_size = local.sum(dim=0)
  if is_offset:
    owner = torch.div(global_sorted_experts, local_expert_size, rounding_mode="floor")
    expert_indices = torch.where(owner == shard_index, torch.remainder(global_sorted_experts, local_expert_size),
                                 torch.full_like(global_sorted_experts, local_expert_size))
  else:
    sizes = local.reshape(-1)
    base = torch.remainder(torch.arange(sizes.shape[0], device=sizes.device), local_expert_size)
    expert_indices = torch.repeat_interleave(base, sizes)
    pad = inputs.shape[0] - expert_indices.shape[0]          # jnp.repeat(total_repeat_length=...): pad with the last value / cut
    if pad > 0:
      expert_indices = torch.cat([expert_indices, expert_indices[-1:].expand(pad) if expert_indices.numel() else expert_indices.new_zeros(pad)])
    expert_indices = expert_indices[:inputs.shape[0]]
  sorted_indices = torch.argsort(expert_indices, stable=True)
  return inputs.index_select(0, sorted_indices), sorted_indices, local_group_size, expert_indices[sorted_indices]


def expert_parallel_apply(x: torch.Tensor, selected_experts: torch.Tensor, num_experts: int,
                          expert_fn: Callable[[int, torch.Tensor], torch.Tensor], group=None, always_exchange: bool = False) -> torch.Tensor:
  """x [rows, emb] and selected_experts int [rows, k] of THIS rank -> y [rows, k, emb] with y[r, j] = expert_fn(selected_experts[r, j], x[r]),
  every expert evaluated on the rank that owns it (experts_per_rank = num_experts / world consecutive ids per rank).  expert_fn(e, rows)
  is called once per local expert that received rows, with the global expert id.  always_exchange: run the collectives on a one-rank
  group too (exercises the RCCL calls on a one-GPU box).  The steps are the reference's (moe.py:799-942): permute -> all-gather of the
  group sizes -> all-to-all (get_all_to_all_params) -> local_permute -> experts -> undo -> all-to-all back -> unpermute."""
  exchange = dist.is_initialized() and (dist.get_world_size(group) > 1 or always_exchange)
  world = dist.get_world_size(group) if exchange else 1
  rank = dist.get_rank(group) if exchange else 0
  if num_experts % world:
    raise ValueError(f"num_experts ({num_experts}) must be a multiple of the expert-parallel degree ({world})")
  per_rank = num_experts // world
  R, K = selected_experts.shape
  order, src_rows, group_size = permute(selected_experts, num_experts)
  sorted_x = x.index_select(0, src_rows)                               # copies grouped by expert = grouped by owning rank
  if exchange:
    gathered = [torch.empty_like(group_size) for _ in range(world)]
    dist.all_gather(gathered, group_size, group=group)                 # moe.py:813-816: every shard learns every shard's group sizes
    all_sizes = torch.stack(gathered)                                  # [batch shard, expert]
    shard_sizes = all_sizes.reshape(world, world, per_rank).sum(dim=-1).cpu().numpy()     # [batch shard, expert shard]
    _, send_sizes, _, recv_sizes = get_all_to_all_params(shard_sizes, rank, world, is_batch_sharded=True)
    send_split, recv_split = [int(v) for v in send_sizes], [int(v) for v in recv_sizes]
    recv_x = x.new_empty((sum(recv_split), x.shape[1]))
    dist.all_to_all_single(recv_x, sorted_x, recv_split, send_split, group=group)
  else:
    all_sizes = group_size[None]
    recv_x = sorted_x
  local_x, local_idx, local_group_size, _ = local_permute(recv_x, all_sizes, per_rank, rank)
  local_y = torch.empty_like(local_x)
  off = 0
  for le, n in enumerate(local_group_size.tolist()):
    if n:
      local_y[off:off + n] = expert_fn(rank * per_rank + le, local_x[off:off + n])
      off += n
  recv_y = torch.empty_like(local_y)
  recv_y.index_copy_(0, local_idx, local_y)                            # moe.py:896-900: back to the order the rows arrived in
  if exchange:
    back = torch.empty_like(sorted_x)
    dist.all_to_all_single(back, recv_y, send_split, recv_split, group=group)
  else:
    back = recv_y
  y = torch.empty((R * K, x.shape[1]), dtype=x.dtype, device=x.device)
  y.index_copy_(0, order, back)                                        # moe.py:455-466: undo the sort
  return y.reshape(R, K, x.shape[1])


def expert_parallel_apply_grouped(x: torch.Tensor, selected_experts: torch.Tensor, num_experts: int,
                                  experts_fn: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], group=None, always_exchange: bool = False,
                                  capacity: int | None = None) -> torch.Tensor:
  """The hot-path form of `expert_parallel_apply`: same result, NO host synchronisation (no size exchange, no `.tolist()`, no tensor whose
  shape depends on the routing) and no per-expert Python loop.

  x [rows, emb], selected_experts int [rows, k] of THIS rank -> y [rows, k, emb].  `experts_fn(rows_x [n, emb], ids int32 [n]) -> [n, emb]` evaluates
  the local experts on packed rows in one call (on a GPU: OpContext.moe_experts = ixt_op_moe_experts); ids < 0 mark padding rows whose result
  is not read.  The exchange has a FIXED capacity: every rank sends `capacity` rows to every rank, so both all-to-alls have equal splits known
  without looking at the routing; the expert ids travel with the rows.  `capacity` must be the same on every rank; the default, rows * k, is
  dropless (the most one destination can receive from this rank) and needs the same number of rows on every rank (decode: max_slots).  A
  smaller capacity drops the copies beyond it -- their y rows are zero, as with the reference's capacity_factor (moe.py:1120-1190).  On the
  fully connected xGMI node the all-to-all gives each of a GPU's 7 links its own slice; the price of not synchronising is the padding
  (decode: rows * k <= a few dozen rows of emb bf16 per peer and layer)."""
  exchange = dist.is_initialized() and (dist.get_world_size(group) > 1 or always_exchange)
  world = dist.get_world_size(group) if exchange else 1
  if num_experts % world:
    raise ValueError(f"num_experts ({num_experts}) must be a multiple of the expert-parallel degree ({world})")
  per_rank = num_experts // world
  R, K = selected_experts.shape
  emb = x.shape[1]
  cap = int(capacity) if capacity is not None else R * K
  flat = selected_experts.reshape(-1).to(torch.int64)
  dest = torch.div(flat, per_rank, rounding_mode="floor")                              # owning rank of every copy
  # slot of a copy inside its destination's buffer = number of earlier copies with the same destination (stable, as the reference's sort)
  onehot = torch.nn.functional.one_hot(dest, world)
  slot = (torch.cumsum(onehot, dim=0) - onehot).gather(1, dest[:, None])[:, 0]
  keep = slot < cap
  trash = world * cap                                                                  # one extra row takes what does not fit
  lin = torch.where(keep, dest * cap + slot, torch.full_like(slot, trash))
  src_rows = torch.div(torch.arange(R * K, device=x.device), K, rounding_mode="floor")
  send_x = x.new_zeros((trash + 1, emb))
  send_id = torch.full((trash + 1,), -1, dtype=torch.int32, device=x.device)
  send_x.index_copy_(0, lin, x.index_select(0, src_rows))
  send_id.index_copy_(0, lin, flat.to(torch.int32))
  send_x, send_id = send_x[:trash], send_id[:trash].contiguous()
  if exchange:
    recv_x, recv_id = torch.empty_like(send_x), torch.empty_like(send_id)
    dist.all_to_all_single(recv_x, send_x, group=group)                                # equal splits: `cap` rows per peer
    dist.all_to_all_single(recv_id, send_id, group=group)
  else:
    recv_x, recv_id = send_x, send_id
  recv_y = experts_fn(recv_x, recv_id)
  if exchange:
    back = torch.empty_like(recv_y)
    dist.all_to_all_single(back, recv_y, group=group)
  else:
    back = recv_y
  y = back.index_select(0, torch.where(keep, lin, torch.zeros_like(lin)))
  y = torch.where(keep[:, None], y, torch.zeros_like(y))
  return y.reshape(R, K, emb)


def combine(y: torch.Tensor, weights: torch.Tensor) -> torch.Tensor:
  """moe.py:467-484 `unpermute`: einsum("BKE,BK -> BE") in f32 over the k copies in top_k order, as one fma chain (the canonical order of
  the single-GPU kernel k_moe_combine); result in y's dtype."""
  t = torch.zeros((y.shape[0], y.shape[2]), dtype=torch.float64 if y.dtype == torch.float64 else torch.float32, device=y.device)
  for j in range(y.shape[1]):
    t = torch.addcmul(t, y[:, j].to(t.dtype), weights[:, j:j + 1].to(t.dtype))
  return t.to(y.dtype)
