"""Expert-parallel exchange (maxtext-indextts2_amd/moe_ep.py): the integer bookkeeping against the cases the
reference's own tests hold (tests/moe_test.py:788-900), and the two-all-to-all exchange on 2 gloo ranks against the single-process result."""
import importlib
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers as H  # noqa: F401  (sys.path)

ep = importlib.import_module("maxtext-indextts2_amd.moe_ep")


def test_get_all_to_all_params_sharded_batch_reference_case():
  g = np.array([[1, 2, 0, 3], [4, 0, 1, 2], [0, 3, 2, 1], [2, 1, 4, 0]], np.int32)
  want_in = np.array([[0, 1, 3, 3], [0, 4, 4, 5], [0, 0, 3, 5], [0, 2, 3, 7]], np.int32)
  want_send = g
  want_out = np.array([[0, 0, 0, 0], [1, 2, 0, 3], [5, 2, 1, 5], [5, 5, 3, 6]], np.int32)
  want_recv = np.array([[1, 4, 0, 2], [2, 0, 3, 1], [0, 1, 2, 4], [3, 2, 1, 0]], np.int32)
  for shard in range(4):
    i, s, o, r = ep.get_all_to_all_params(g, shard, 4, is_batch_sharded=True)
    assert np.array_equal(i, want_in[shard]) and np.array_equal(s, want_send[shard])
    assert np.array_equal(o, want_out[shard]) and np.array_equal(r, want_recv[shard])


def test_get_all_to_all_params_unsharded_batch_reference_case():
  g = np.array([6, 7, 6, 7], np.int32)
  want_send = np.array([[6] * 4, [7] * 4, [6] * 4, [7] * 4], np.int32)
  want_out = np.array([[0] * 4, [6] * 4, [13] * 4, [19] * 4], np.int32)
  for shard in range(4):
    i, s, o, r = ep.get_all_to_all_params(g, shard, 4, is_batch_sharded=False)
    assert np.array_equal(i, np.zeros(4, np.int32)) and np.array_equal(s, want_send[shard])
    assert np.array_equal(o, want_out[shard]) and np.array_equal(r, g)


def test_permute_matches_the_reference_rule():
  sel = torch.tensor([[3, 1], [0, 3], [1, 0], [2, 2]], dtype=torch.int64)
  order, rows, group = ep.permute(sel, 4)
  flat = sel.reshape(-1).numpy()
  assert np.array_equal(order.numpy(), np.argsort(flat, kind="stable"))      # jnp.argsort is stable
  assert np.array_equal(rows.numpy(), np.argsort(flat, kind="stable") // 2)
  assert np.array_equal(group.numpy(), np.bincount(flat, minlength=4))


def test_local_permute_no_offset_reference_case():
  """tests/moe_test.py:634-712: 8 experts on 4 shards, expert e holds e rows; a shard's input slice is already in local-expert order."""
  sizes = torch.arange(8)
  total = int(sizes.sum())
  original = torch.arange(total * 5, dtype=torch.int32).reshape(total, 5)
  cum = torch.cumsum(sizes, 0)
  for shard in range(4):
    a = 0 if shard == 0 else int(cum[2 * shard - 1])
    b = int(cum[2 * shard + 1])
    si, idx, lgs, ids = ep.local_permute(original[a:b], sizes[None, :], 2, shard, is_offset=False)
    assert torch.equal(si, original[a:b]) and torch.equal(idx, torch.arange(b - a))
    assert torch.equal(lgs, sizes[2 * shard:2 * shard + 2])
    assert torch.equal(ids, torch.repeat_interleave(torch.arange(2), sizes[2 * shard:2 * shard + 2]))


def test_local_permute_offset_reference_case():
  """tests/moe_test.py:714-786: all copies sorted by global expert id; a shard takes its slice and its experts' local ids."""
  for counts in (torch.arange(8), torch.tensor([0, 0, 1, 1, 2, 0, 2, 2])):
    total = int(counts.sum())
    x = torch.arange(1, total + 1).reshape(-1, 1).repeat(1, 5)
    local_sizes = counts.reshape(4, 2).sum(-1)
    assign = torch.repeat_interleave(torch.arange(8), counts)
    offs = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(local_sizes, 0)[:-1]])
    for shard in range(4):
      px, idx, lcounts, lassign = ep.local_permute(x, counts[None, :], 2, shard, is_offset=True, global_sorted_experts=assign)
      n, o = int(local_sizes[shard]), int(offs[shard])
      assert torch.equal(px[:n], x[o:o + n]) and torch.equal(idx[:n], torch.arange(o, o + n))
      assert torch.equal(lcounts, counts[2 * shard:2 * shard + 2])
      assert torch.equal(lassign[:n], torch.remainder(assign[o:o + n], 2))


def test_local_permute_groups_what_two_sources_delivered():
  """Rows from 2 source shards for 2 local experts arrive as [s0e0 s0e1 s1e0 s1e1]; the shard wants [e0 rows | e1 rows], sources in order."""
  g = torch.tensor([[2, 1, 9, 9], [1, 3, 9, 9]])            # shard 0 owns experts 0, 1
  rows = torch.arange(7).reshape(7, 1)
  si, idx, lgs, ids = ep.local_permute(rows, g, 2, 0)
  assert si.reshape(-1).tolist() == [0, 1, 3, 2, 4, 5, 6] and lgs.tolist() == [3, 4] and ids.tolist() == [0, 0, 0, 1, 1, 1, 1]
  assert idx.tolist() == [0, 1, 3, 2, 4, 5, 6]


def _expert(e, rows):
  return rows * float(e + 1) + float(e)


def _case(rank, n_rows, k, num_experts, emb=16):
  g = torch.Generator().manual_seed(100 + rank)
  x = torch.randn((n_rows, emb), generator=g)
  sel = torch.stack([torch.randperm(num_experts, generator=g)[:k] for _ in range(n_rows)])
  w = torch.rand((n_rows, k), generator=g)
  return x, sel, w


def test_single_process_exchange_is_the_direct_evaluation():
  x, sel, w = _case(0, 13, 2, 8)
  y = ep.expert_parallel_apply(x, sel, 8, _expert)
  for r in range(13):
    for j in range(2):
      assert torch.equal(y[r, j], _expert(int(sel[r, j]), x[r]))
  out = ep.combine(y, w)
  ref = sum(y[:, j] * w[:, j:j + 1] for j in range(2))
  assert torch.allclose(out, ref, rtol=1e-6, atol=1e-6)


def _free_port():
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    return s.getsockname()[1]


def _worker(rank, world, port, q):
  dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
  try:
    calls = []

    def fn(e, rows):
      calls.append((e, rows.shape[0]))
      return _expert(e, rows)

    n_rows = [11, 5][rank]                      # ragged: the ranks hold different numbers of rows
    x, sel, w = _case(rank, n_rows, 2, 8)
    y = ep.expert_parallel_apply(x, sel, 8, fn)
    ok = all(torch.equal(y[r, j], _expert(int(sel[r, j]), x[r])) for r in range(n_rows) for j in range(2))
    owned = all(rank * 4 <= e < rank * 4 + 4 for e, _ in calls)       # only this rank's experts were evaluated here
    q.put((rank, ok, owned, sum(n for _, n in calls)))
  finally:
    dist.destroy_process_group()


def test_world_size_2_gloo_exchange():
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
  for p in procs:
    p.start()
  res = [q.get(timeout=120) for _ in procs]
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  assert all(ok and owned for _, ok, owned, _ in res)
  assert sum(n for *_, n in res) == (11 + 5) * 2                      # every (row, k) copy was evaluated exactly once somewhere


# ---------------------------------------------------------------- the hot-path form: fixed capacity, one call for all local experts
def _experts_packed(first, count):
  """experts_fn of expert_parallel_apply_grouped for experts [first, first + count): padding ids and other ranks' ids -> zero rows."""
  def fn(rows, ids):
    out = torch.zeros_like(rows)
    for i in range(rows.shape[0]):
      e = int(ids[i])
      if first <= e < first + count:
        out[i] = _expert(e, rows[i])
    return out
  return fn


def test_grouped_single_process_is_the_direct_evaluation():
  x, sel, _ = _case(0, 13, 2, 8)
  y = ep.expert_parallel_apply_grouped(x, sel, 8, _experts_packed(0, 8))
  for r in range(13):
    for j in range(2):
      assert torch.equal(y[r, j], _expert(int(sel[r, j]), x[r]))
  assert torch.equal(y, ep.expert_parallel_apply(x, sel, 8, _expert))


def test_grouped_capacity_drops_only_what_does_not_fit():
  x, sel, _ = _case(3, 9, 2, 8)
  y = ep.expert_parallel_apply_grouped(x, sel, 8, _experts_packed(0, 8), capacity=5)   # one destination (world 1): copies 0..4 fit
  flat = y.reshape(18, -1)
  for p in range(18):
    r, j = divmod(p, 2)
    want = _expert(int(sel[r, j]), x[r]) if p < 5 else torch.zeros_like(x[r])
    assert torch.equal(flat[p], want)


def _worker_grouped(rank, world, port, q):
  dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
  try:
    n_rows = [11, 5][rank]                      # ragged rows: the capacity is agreed beforehand (the larger rank's rows * k)
    x, sel, _ = _case(rank, n_rows, 2, 8)
    seen = []

    def fn(rows, ids):
      seen.append(int((ids >= 0).sum()))
      assert all(rank * 4 <= int(e) < rank * 4 + 4 for e in ids[ids >= 0])     # only rows of this rank's experts arrive
      return _experts_packed(rank * 4, 4)(rows, ids)

    y = ep.expert_parallel_apply_grouped(x, sel, 8, fn, capacity=22)
    ok = all(torch.equal(y[r, j], _expert(int(sel[r, j]), x[r])) for r in range(n_rows) for j in range(2))
    q.put((rank, ok, len(seen) == 1, seen[0]))
  finally:
    dist.destroy_process_group()


def test_grouped_world_size_2_gloo_exchange():
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_worker_grouped, args=(r, 2, port, q)) for r in range(2)]
  for p in procs:
    p.start()
  res = [q.get(timeout=120) for _ in procs]
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  assert all(ok and once for _, ok, once, _ in res)
  assert sum(n for *_, n in res) == (11 + 5) * 2                      # every (row, k) copy was evaluated exactly once somewhere
