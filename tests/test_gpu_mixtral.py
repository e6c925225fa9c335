"""GPU parity of the Mixtral block (layers/mixtral.py:52-170, layers/moe.py) against the oracle through the
engine's call order: tokens, f32 logits bits, K and V of every layer, PageState.  8 experts, 2 per token, 4 query heads per kv head."""
import numpy as np
import pytest

import helpers as H
from test_gpu_engine import packed_prefill_pair, run_pair

pytestmark = pytest.mark.gpu


def _cfg(**kw):
  base = dict(max_prefill_predict_length=48, max_target_length=128)
  base.update(kw)
  return H.make_config("mixtral-tiny", **base)


def _prompt(vocab, n, seed):
  return np.random.Generator(np.random.PCG64(seed)).integers(1, vocab, size=n).astype(np.int32)


def test_mixtral_single_slot():
  cfg = _cfg()
  w = H.synth.synth_weights(cfg, 0)
  run_pair(cfg, w, [(_prompt(cfg.vocab_size, 48, 1234), 41)], steps=40)


@pytest.mark.parametrize("use_graph", [True, False])
def test_mixtral_batched_ragged(use_graph):
  cfg = _cfg(per_device_batch_size=6.0, pagedattn_num_pages=160, ixt_use_hip_graph=use_graph)
  w = H.synth.synth_weights(cfg, 1)
  prompts = [(_prompt(cfg.vocab_size, 48, 100 + s), [48, 1, 25, 24, 33][s]) for s in range(5)]
  run_pair(cfg, w, prompts, steps=36)


@pytest.mark.parametrize("options", [{"moe_fused_gate": 0}, {"moe_route_blocks": 0}, {"small_attn_split": 0}, {"moe_mid_pairs": 2048}, {"moe_grouped": 0}])
def test_mixtral_kernel_forms_agree_with_the_oracle(options):
  """Every alternative form of a Mixtral step against the same oracle run: split-K wi + separate gate kernel, one-wave-per-expert router, one-launch
  decode attention instead of the split one, 64-row whole-K expert dots for prefill-sized copy counts, every expert over all rows."""
  cfg = _cfg(per_device_batch_size=3.0, pagedattn_num_pages=96)
  w = H.synth.synth_weights(cfg, 4)
  prompts = [(_prompt(cfg.vocab_size, 48, 300 + s), [48, 7, 29][s]) for s in range(3)]
  run_pair(cfg, w, prompts, steps=12, options=options)


def test_mixtral_chunked_prefill():
  cfg = _cfg(max_prefill_predict_length=16, per_device_batch_size=2.0, use_chunked_prefill=True, prefill_chunk_size=16)
  w = H.synth.synth_weights(cfg, 2)
  run_pair(cfg, w, [(_prompt(cfg.vocab_size, 48, 77), 44), (_prompt(cfg.vocab_size, 48, 78), 30)], steps=12, chunks=[[16, 16, 12], [16, 14]])


@pytest.mark.parametrize("ne,k", [(4, 1), (8, 8), (16, 4), (1, 1)])
def test_mixtral_expert_counts(ne, k):
  cfg = _cfg(per_device_batch_size=2.0, num_experts=ne, num_experts_per_tok=k)
  w = H.synth.synth_weights(cfg, 4)
  prompts = [(_prompt(cfg.vocab_size, 48, 50 + s), [37, 29][s]) for s in range(2)]
  run_pair(cfg, w, prompts, steps=10)


def test_mixtral_many_rows():
  """130 decode slots and a 150-row prefill: the per-expert dots take their >= 128-row forms."""
  cfg = H.make_config("mixtral-tiny", max_prefill_predict_length=160, max_target_length=192, per_device_batch_size=130.0, pagedattn_num_pages=130 * 24 + 8,
                      base_num_decoder_layers=2)
  w = H.synth.synth_weights(cfg, 6)
  prompts = [(_prompt(cfg.vocab_size, 160, 900 + s), 150 if s == 0 else [9, 17, 33, 2][s % 4]) for s in range(130)]
  run_pair(cfg, w, prompts, steps=3, check_logits=False)


@pytest.mark.parametrize("nprompts,hi", [(6, 32), (24, 12)])
def test_mixtral_packed_prefill(nprompts, hi):
  packed_prefill_pair(nprompts, hi, model_name="mixtral-tiny")


def test_mixtral_8x7b_widths():
  """The widths of configs/models/mixtral-8x7b.yml (emb 4096, 32 x 128 query heads over 8 kv heads, mlp 14336, 8 experts, 2 per token) with 1 of
  its 32 layers and a 2048-row vocabulary: every projection and expert kernel shape against the oracle."""
  cfg = H.make_config("mixtral-8x7b", base_num_decoder_layers=1, vocab_size=2048, max_prefill_predict_length=16, max_target_length=32,
                      per_device_batch_size=2.0, pagedattn_num_pages=16)
  w = H.synth.synth_weights(cfg, 10)
  prompts = [(_prompt(cfg.vocab_size, 16, 80 + s), [13, 6][s]) for s in range(2)]
  run_pair(cfg, w, prompts, steps=3)


@pytest.mark.parametrize("rccl", [False, True])
def test_routed_moe_module_with_exchange_matches_oracle(rccl):
  """layers.moe.RoutedMoE over the op-level entry points (route -> exchange -> per-expert MLP -> exchange -> combine) against the oracle's
  mixture, bit for bit; rccl=True runs the two all-to-alls on a one-rank RCCL group (the expert-parallel code path on this one-GPU box)."""
  import ctypes
  import importlib
  import torch
  import torch.distributed as dist
  from oracle import oracle as O
  cfg = _cfg(max_prefill_predict_length=64)
  w = H.synth.synth_weights(cfg, 12)
  eng, _ = H.make_engine(cfg, w)
  orc = H.make_oracle(cfg, w)
  moe = importlib.import_module("maxtext-indextts2_amd.layers.moe")
  rng = np.random.default_rng(3)
  R, E = 37, cfg.emb_dim
  xn = H.synth.f32_to_bf16_bits(rng.standard_normal((R, E)).astype(np.float32))
  want = np.zeros((R, E), np.uint16)
  routing = np.zeros((R, 2), np.int32)
  O.lib().orc_moe_rows(ctypes.byref(orc.model.c), 1, xn.ctypes.data_as(O._u16p), R, want.ctypes.data_as(O._u16p), routing.ctypes.data_as(O._i32p))
  started = False
  if rccl and not dist.is_initialized():
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{H.free_port()}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    started = True
  try:
    m = moe.get_routed_moe(cfg, cfg.num_experts, cfg.num_experts_per_tok, engine=eng, layer=1, always_exchange=rccl)
    out, loss = m(H.dev_bf16(xn).reshape(1, R, E))
    torch.cuda.synchronize()
    assert loss is None and np.array_equal(H.bits(out.reshape(R, E)), want)
    experts, _ = m.ctx.moe_route(1, H.dev_bf16(xn))
    assert np.array_equal(experts.cpu().numpy(), routing)
  finally:
    if started:
      dist.destroy_process_group()


def test_mixtral_transformer_from_modules_expert_parallel_path():
  """models.Transformer / decoders.Decoder / mixtral.MixtralDecoderLayer / moe.RoutedMoE assembled from the modules, the experts reached through
  the expert-parallel exchange (the two all-to-alls on a one-rank RCCL group): prefill logits of every row of two prompts and three decode
  steps against the oracle, bit for bit -- the launch sequence an expert-parallel decode over N GPUs runs on each rank."""
  import importlib
  import torch
  import torch.distributed as dist
  runtime = importlib.import_module("maxtext-indextts2_amd.runtime")
  layers_models = importlib.import_module("maxtext-indextts2_amd.layers.models")
  ct = importlib.import_module("maxtext-indextts2_amd.common_types")
  cfg = H.make_config("mixtral-tiny", per_device_batch_size=3.0, max_prefill_predict_length=32, max_target_length=64, pagedattn_num_pages=64)
  w = H.synth.synth_weights(cfg, 21)
  orc = H.make_oracle(cfg, w)
  saved = runtime._default
  ctx = runtime.set_context(runtime.OpContext(cfg))
  started = False
  if not dist.is_initialized():
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{H.free_port()}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    started = True
  try:
    model = layers_models.Transformer(cfg, model_mode=ct.MODEL_MODE_PREFILL)
    for layer in model.decoder.layers:
      layer.MoeBlock_0.always_exchange = True
    model.load_params(w)
    pm = ctx.page_manager
    prompts = [(_prompt(cfg.vocab_size, 32, 900), 27), (_prompt(cfg.vocab_size, 32, 901), 9)]
    nxt = []
    for slot, (toks, tl) in enumerate(prompts):
      ps = pm.update_prefill_pages(None, slot, tl)
      t = torch.from_numpy(toks[:tl][None]).cuda()
      pos = torch.arange(tl, dtype=torch.int32, device="cuda")[None]
      logits = model(t, pos, decoder_segment_ids=torch.ones((1, tl), dtype=torch.int32, device="cuda"), model_mode=ct.MODEL_MODE_PREFILL, slot=slot, page_state=ps)
      orc.page_state = orc.pm.update_prefill_pages(orc.page_state, slot, tl)
      o = orc.model.forward(toks[:tl], np.full((tl,), slot, np.int32), np.arange(tl, dtype=np.int32), np.ones((tl,), np.int32), orc.page_state.page_map,
                            logits_rows=list(range(tl)), want_logits=True)
      assert np.array_equal(logits[0].cpu().numpy().view(np.uint32), o["logits"].view(np.uint32)), slot
      nxt.append(int(o["argmax"][-1]))
    B = int(cfg.per_device_batch_size)
    tok = np.zeros((B,), np.int32)
    tok[:2] = nxt
    for step in range(3):
      ps = pm.update_decode_pages(None)
      orc.page_state = orc.pm.update_decode_pages(orc.page_state)
      pos = np.maximum(orc.page_state.sequence_lengths - 1, 0).astype(np.int32)
      logits = model(torch.from_numpy(tok[:, None]).cuda(), torch.from_numpy(pos[:, None]).cuda(), model_mode=ct.MODEL_MODE_AUTOREGRESSIVE, page_state=ps)
      act = orc.page_state.has_active_page.astype(np.int32)
      o = orc.model.forward(tok, np.arange(B, dtype=np.int32), pos, act, orc.page_state.page_map, logits_rows=list(range(B)), want_logits=True)
      assert np.array_equal(logits[:, 0].cpu().numpy()[:2].view(np.uint32), o["logits"][:2].view(np.uint32)), step
      tok[:2] = o["argmax"][:2]
  finally:
    if started:
      dist.destroy_process_group()
    runtime._default = saved


def test_expert_sharded_contexts_cover_the_mixture():
  """Expert parallelism below Python: two contexts that each keep HALF of the experts' kernels in HBM (ixt_config.ep_rank / ep_size) evaluate,
  with ixt_op_moe_experts, only the rows of their own experts (other rows come back as zeros: what the fixed-capacity exchange pads with);
  together they give every (row, k) copy exactly the oracle's per-expert output, and the mixture assembled from them equals orc_moe_rows bit
  for bit.  The fused engine entry points of a sharded context refuse."""
  import ctypes
  import importlib
  import torch
  from oracle import oracle as O
  runtime = importlib.import_module("maxtext-indextts2_amd.runtime")
  cfg = _cfg(max_prefill_predict_length=64)
  w = H.synth.synth_weights(cfg, 14)
  orc = H.make_oracle(cfg, w)
  full_eng, _ = H.make_engine(cfg, w)
  full = runtime.OpContext(engine=full_eng)
  NE, TK, E = int(cfg.num_experts), int(cfg.num_experts_per_tok), cfg.emb_dim
  shards = []
  for rank in range(2):
    c = _cfg(max_prefill_predict_length=64, ixt_expert_parallel_rank=rank, ixt_expert_parallel_size=2)
    eng, _ = H.make_engine(c, w)
    shards.append((eng, runtime.OpContext(engine=eng)))
  rng = np.random.default_rng(5)
  R = 29
  xn = H.synth.f32_to_bf16_bits(rng.standard_normal((R, E)).astype(np.float32))
  want = np.zeros((R, E), np.uint16)
  routing = np.zeros((R, TK), np.int32)
  O.lib().orc_moe_rows(ctypes.byref(orc.model.c), 1, xn.ctypes.data_as(O._u16p), R, want.ctypes.data_as(O._u16p), routing.ctypes.data_as(O._i32p))
  x = H.dev_bf16(xn)
  experts, weights = full.moe_route(1, x)
  assert np.array_equal(experts.cpu().numpy(), routing)
  rows = x.repeat_interleave(TK, dim=0).contiguous()                    # one row per (row, k) copy
  ids = experts.reshape(-1).to(torch.int32)
  ids_pad = torch.cat([ids, torch.full((3,), -1, dtype=torch.int32, device="cuda")])     # padding rows of the fixed-capacity exchange
  rows_pad = torch.cat([rows, rows[:3]])
  parts = [ctx.moe_experts(1, rows_pad, ids_pad) for _, ctx in shards]
  torch.cuda.synchronize()
  per = NE // 2
  y = torch.zeros_like(rows)
  for rank, part in enumerate(parts):
    mine = (ids >= rank * per) & (ids < (rank + 1) * per)
    assert not part[:R * TK][~mine].any() and not part[R * TK:].any()  # rows of other ranks' experts and padding rows: zeros
    y = torch.where(mine[:, None], part[:R * TK], y)
  whole = full.moe_experts(1, rows, ids)                                # the same through a context with every expert
  assert torch.equal(whole, y)
  for e in range(NE):                                                   # and through the one-expert op of round 2
    sel = (ids == e).nonzero()[:, 0]
    if sel.numel():
      assert torch.equal(full.moe_expert(1, e, rows[sel].contiguous()), y[sel])
  out = full.moe_combine(y.reshape(R, TK, E).contiguous(), weights)
  torch.cuda.synchronize()
  assert np.array_equal(H.bits(out), want)
  toks, tl = _prompt(cfg.vocab_size, 64, 7), 11
  with pytest.raises(NotImplementedError):
    shards[0][0].prefill(params=None, padded_tokens=toks, true_length=tl, slot=0)
