"""Expert parallelism on TWO processes sharing the test GPU (layers/moe.py:785-938): each process builds a context that
holds HALF of the experts' kernels (ixt_config.ep_rank / ep_size), routes ITS OWN rows, and the fixed-capacity exchange of moe_ep.py
(expert_parallel_apply_grouped) carries the (row, k) copies to the process that owns their expert and the results back; the mixture every process
assembles must equal the oracle's (orc_moe_rows) bit for bit.  The collectives run on gloo (the box has one GPU: RCCL needs one device per rank), so
the rows cross the process boundary through host memory -- the exchange logic, the sharded contexts and ixt_op_moe_experts are the real ones."""
import ctypes
import importlib
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers as H

pytestmark = pytest.mark.gpu


def _free_port():
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    return s.getsockname()[1]


def _worker(rank, world, port, q):
  try:
    from oracle import oracle as O
    ep = importlib.import_module("maxtext-indextts2_amd.moe_ep")
    runtime = importlib.import_module("maxtext-indextts2_amd.runtime")
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    cfg_full = H.make_config("mixtral-tiny", max_prefill_predict_length=64, max_target_length=128)
    w = H.synth.synth_weights(cfg_full, 14)
    orc = H.make_oracle(cfg_full, w)
    cfg = H.make_config("mixtral-tiny", max_prefill_predict_length=64, max_target_length=128, ixt_expert_parallel_rank=rank, ixt_expert_parallel_size=world)
    eng, _ = H.make_engine(cfg, w)
    ctx = runtime.OpContext(engine=eng)
    NE, TK, E = int(cfg.num_experts), int(cfg.num_experts_per_tok), cfg.emb_dim
    R = 13                                                                          # this rank's rows (the same count on every rank: decode slots)
    rng = np.random.default_rng(100 + rank)
    xn = H.synth.f32_to_bf16_bits(rng.standard_normal((R, E)).astype(np.float32))
    want = np.zeros((R, E), np.uint16)
    routing = np.zeros((R, TK), np.int32)
    O.lib().orc_moe_rows(ctypes.byref(orc.model.c), 1, xn.ctypes.data_as(O._u16p), R, want.ctypes.data_as(O._u16p), routing.ctypes.data_as(O._i32p))
    x = H.dev_bf16(xn)
    experts, weights = ctx.moe_route(1, x)                                          # the gate kernel is replicated: every rank routes its own rows
    assert np.array_equal(experts.cpu().numpy(), routing)
    calls = []

    def experts_fn(rows_cpu, ids_cpu):                                              # what arrives here belongs to THIS rank's experts (or is padding)
      live = ids_cpu[ids_cpu >= 0]
      per = NE // world
      assert bool(((live >= rank * per) & (live < (rank + 1) * per)).all())
      calls.append(int(live.numel()))
      y = ctx.moe_experts(1, rows_cpu.cuda().contiguous(), ids_cpu.cuda().contiguous())
      return y.cpu()

    y = ep.expert_parallel_apply_grouped(x.cpu(), experts.cpu(), NE, experts_fn, always_exchange=True)     # [R, TK, E], host memory (gloo)
    out = ctx.moe_combine(y.cuda().contiguous(), weights)
    torch.cuda.synchronize()
    ok = bool(np.array_equal(H.bits(out), want))
    q.put((rank, ok, len(calls), calls[0] if calls else -1))
  except Exception as e:  # pylint: disable=broad-except
    q.put((rank, False, -1, repr(e)))
  finally:
    if dist.is_initialized():
      dist.destroy_process_group()


def test_two_processes_with_sharded_contexts_reproduce_the_mixture():
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
  for p in procs:
    p.start()
  res = [q.get(timeout=300) for _ in procs]
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  assert all(ok for _, ok, _, _ in res), res
  assert all(n == 1 for _, _, n, _ in res), res                          # one packed call per rank and layer
  assert sum(c for *_, c in res) == 2 * 13 * 2                            # every (row, k) copy of both ranks was evaluated exactly once somewhere
