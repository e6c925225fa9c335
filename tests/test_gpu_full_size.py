"""Size-independent properties at the BASELINE.json sizes of the other two model families (configs[3] Gemma-2 2B with every layer, configs[4]
Mixtral-8x7B widths with four layers), where the CPU oracle would take minutes: chunked prefill == one-chunk prefill, packed prefill == one
by one, hipGraph decode == eager decode -- bit for bit, on weights drawn on the device (synth.DeviceTree).  The arithmetic itself is
checked against the oracle at these widths with fewer layers in test_gpu_gemma2.py / test_gpu_mixtral.py."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def _engine(cfg, seed=3):
  maxengine = H.maxengine_module()
  eng = maxengine.MaxEngine(cfg)
  params = eng.load_params(params=H.synth.device_tree(cfg, seed))
  return maxengine, eng, params


def _prompt(vocab, n, seed):
  return np.random.Generator(np.random.PCG64(seed)).integers(1, vocab, size=n).astype(np.int32)


def _chunked_vs_whole_and_graph_vs_eager(model, steps=6, **kw):
  import torch
  outs = []
  for use_graph, chunks in ((True, [64]), (False, [32, 32]), (True, [16, 48])):
    cfg = H.make_config(model, per_device_batch_size=2.0, max_prefill_predict_length=64, max_target_length=128, pagedattn_tokens_per_page=32,
                        pagedattn_num_pages=16, use_chunked_prefill=True, prefill_chunk_size=64, ixt_use_hip_graph=use_graph, **kw)
    maxengine, eng, params = _engine(cfg)
    toks = _prompt(cfg.vocab_size, 64, 11)
    done, pre, first = 0, None, None
    for n in chunks:
      ep = None if done == 0 else maxengine.ExistingPrefix(cache=pre["cache"], common_prefix_tokens=torch.from_numpy(toks[:done].copy()))
      pre, res = eng.prefill(params=params, padded_tokens=toks[done:done + n], true_length=n, slot=1, existing_prefix=ep)
      done += n
    ds = eng.insert(pre, eng.init_decode_state(materialize_logits=True), 1)
    seq = [int(res.data.cpu().numpy()[0, 0])]
    for _ in range(steps):
      ds, r = eng.generate(params, ds)
      seq.append(int(r.data.cpu().numpy()[1, 0]))
    k, v = eng.gather_kv(H.sublayers(cfg) - 1, 1, 64 + steps)
    outs.append((seq, pre["logits"].cpu().numpy().reshape(-1).view(np.uint32).copy(), ds["logits"][1].cpu().numpy().reshape(-1).view(np.uint32).copy(),
                 H.bits(k).copy(), H.bits(v).copy()))
    del eng
    torch.cuda.empty_cache()
  for o in outs[1:]:
    assert o[0] == outs[0][0]
    for a, b in zip(o[1:], outs[0][1:]):
      assert np.array_equal(a, b)
  lg = outs[0][2].view(np.float32)
  assert np.isfinite(lg).all() and float(lg.max()) > float(lg.min())       # a live model (a tied head may well echo one token for ever)


def test_gemma2_2b_every_layer_chunked_and_graph_invariance():
  """configs/models/gemma2-2b.yml + the 2048 appended vocabulary rows of BASELINE configs[3], all 13 merged layers (26 sub-layers), window 4096."""
  _chunked_vs_whole_and_graph_vs_eager("gemma2-2b", vocab_size=256128 + 2048)


def test_mixtral_8x7b_widths_four_layers_chunked_and_graph_invariance():
  _chunked_vs_whole_and_graph_vs_eager("mixtral-8x7b", base_num_decoder_layers=4)


def test_gemma3_4b_seven_layers_chunked_and_graph_invariance():
  """gemma3-4b widths (a 4096-row vocabulary, host-side weights) over one whole [5 local, 1 global] pattern and the next local layer; window 32 <
  the 64-token prompt."""
  cfg_kw = dict(base_num_decoder_layers=7, sliding_window_size=32)
  import torch
  outs = []
  cfg0 = H.make_config("gemma3-4b", per_device_batch_size=2.0, max_prefill_predict_length=64, max_target_length=128, pagedattn_tokens_per_page=32,
                       pagedattn_num_pages=16, use_chunked_prefill=True, prefill_chunk_size=64, vocab_size=4096, **cfg_kw)
  w = H.synth.synth_weights(cfg0, 5)
  for use_graph, chunks in ((True, [64]), (False, [32, 32])):
    cfg = H.make_config("gemma3-4b", per_device_batch_size=2.0, max_prefill_predict_length=64, max_target_length=128, pagedattn_tokens_per_page=32,
                        pagedattn_num_pages=16, use_chunked_prefill=True, prefill_chunk_size=64, vocab_size=4096, ixt_use_hip_graph=use_graph, **cfg_kw)
    eng, params = H.make_engine(cfg, w)
    maxengine = H.maxengine_module()
    toks = _prompt(cfg.vocab_size, 64, 12)
    done, pre = 0, None
    for n in chunks:
      ep = None if done == 0 else maxengine.ExistingPrefix(cache=pre["cache"], common_prefix_tokens=torch.from_numpy(toks[:done].copy()))
      pre, res = eng.prefill(params=params, padded_tokens=toks[done:done + n], true_length=n, slot=0, existing_prefix=ep)
      done += n
    ds = eng.insert(pre, eng.init_decode_state(), 0)
    seq = [int(res.data.cpu().numpy()[0, 0])]
    for _ in range(5):
      ds, r = eng.generate(params, ds)
      seq.append(int(r.data.cpu().numpy()[0, 0]))
    outs.append((seq, pre["logits"].cpu().numpy().reshape(-1).view(np.uint32).copy()))
    del eng
    torch.cuda.empty_cache()
  assert outs[0][0] == outs[1][0] and np.array_equal(outs[0][1], outs[1][1])


def _long_prompt_runs(model, total, plans, steps=3, **kw):
  """One `total`-token prompt through the engine's chunked prefill under several (chunk size, hipGraph) plans: first token, last-row logits, the last
  sub-layer's K / V of the whole prompt and `steps` decode steps must agree bit for bit."""
  import torch
  outs = []
  for chunk, use_graph in plans:
    cfg = H.make_config(model, per_device_batch_size=1.0, max_prefill_predict_length=chunk, max_target_length=total + 64, pagedattn_tokens_per_page=32,
                        pagedattn_num_pages=(total + 64) // 32 + 2, use_chunked_prefill=True, prefill_chunk_size=chunk, ixt_use_hip_graph=use_graph, **kw)
    maxengine, eng, params = _engine(cfg)
    toks = torch.from_numpy(_prompt(cfg.vocab_size, total, 21)).to(eng.device)
    done, pre, res = 0, None, None
    while done < total:
      n = min(chunk, total - done)
      ep = None if done == 0 else maxengine.ExistingPrefix(cache=pre["cache"], common_prefix_tokens=toks[:done])
      pre, res = eng.prefill(params=params, padded_tokens=toks[done:done + n], true_length=n, slot=0, existing_prefix=ep)
      done += n
    ds = eng.insert(pre, eng.init_decode_state(materialize_logits=True), 0)
    seq = [int(res.data.cpu().numpy()[0, 0])]
    for _ in range(steps):
      ds, r = eng.generate(params, ds)
      seq.append(int(r.data.cpu().numpy()[0, 0]))
    k, v = eng.gather_kv(H.sublayers(cfg) - 1, 0, total + steps)
    outs.append((seq, pre["logits"].cpu().numpy().reshape(-1).view(np.uint32).copy(), ds["logits"][0].cpu().numpy().reshape(-1).view(np.uint32).copy(),
                 H.bits(k).copy(), H.bits(v).copy()))
    eng.close()
    del eng
    torch.cuda.empty_cache()
  for o in outs[1:]:
    assert o[0] == outs[0][0]
    for a, b in zip(o[1:], outs[0][1:]):
      assert np.array_equal(a, b)
  lg = outs[0][2].view(np.float32)
  assert np.isfinite(lg).all() and float(lg.max()) > float(lg.min())


def test_gemma2_2b_4096_token_chunked_prefill():
  """BASELINE.json configs[3]'s workload shape: Gemma-2 2B + the 2048 appended vocabulary rows, every layer, ONE 4096-token prompt (the local
  sub-layers' window is 4096: the last chunks see the whole prompt).  256-row chunks == 1024-row chunks == one 4096-row pass, hipGraph == eager."""
  _long_prompt_runs("gemma2-2b", 4096, [(256, True), (1024, False), (4096, True)], vocab_size=256128 + 2048)


def test_gemma2_2b_4096_token_chunked_prefill_fp8():
  """The same workload with `quantization: fp8` (IXT-FP8-v1): the activation scales belong to a row and a 512-k block, so a row's result still does
  not depend on the launch it runs in -- 256-row chunks == 1024-row chunks, hipGraph == eager, bit for bit."""
  _long_prompt_runs("gemma2-2b", 4096, [(256, True), (1024, False)], vocab_size=256128 + 2048, quantization="fp8")
