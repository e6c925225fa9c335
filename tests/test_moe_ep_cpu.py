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
