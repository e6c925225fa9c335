"""GPU tests of IXT-FP8-v1 (`quantization: fp8`, BASELINE.json configs[3]; csrc/ixt_fp8.h).  The reference has no fp8 inference behaviour
(quantizations.py:289-298 is training-only AQT, decode.py:127-128 refuses it), so nothing here is bit-compared with it.  What is checked:
  1. the device's e4m3 conversion == the OCP rule restated in oracle/fp8_ref.py, on a dense sweep of inputs;
  2. the fp8 dot (both kernel forms: split-K below 128 rows, the LDS-ring GEMM from 128 rows) == the float64 restatement of the same
     quantised operands within the matrix cores' accumulation noise (|err| <= 2e-5 x sum of |products|; measured: up to 4.4e-6 -- the instruction aligns
     and truncates its products much as the bf16 one does);
  3. a model with fp8 projection kernels against the bf16 oracle on the same weights: the error IXT-FP8-v1 costs, stated as tolerances.
"""
import ctypes

import numpy as np
import pytest

import helpers as H
from oracle import fp8_ref as F8

pytestmark = pytest.mark.gpu


def _st(torch):
  return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _vp(t):
  return ctypes.c_void_p(t.data_ptr())


@pytest.fixture(scope="module")
def ctx8():
  import torch
  cfg = H.make_config(per_device_batch_size=64.0, pagedattn_num_pages=256, max_prefill_predict_length=256, max_target_length=320, quantization="fp8",
                      base_mlp_dim=2048)
  w = H.synth.synth_weights(cfg, 0)
  eng, _ = H.make_engine(cfg, w)
  return {"cfg": cfg, "eng": eng, "torch": torch}


def test_e4m3_conversion_matches_the_ocp_rule(ctx8):
  torch, eng = ctx8["torch"], ctx8["eng"]
  rng = np.random.default_rng(0)
  vals = np.concatenate([
      np.linspace(-500.0, 500.0, 200001), rng.standard_normal(100000) * 3.0, rng.standard_normal(100000) * 0.01,
      np.ldexp(1.0, np.arange(-14, 10)), np.ldexp(1.0625, np.arange(-12, 9)), np.ldexp(1.1875, np.arange(-12, 9)),   # ties: 1.0625 = halfway 1.0 / 1.125
      np.array([0.0, -0.0, 448.0, 449.0, 464.0, 480.0, 1e9, -1e9, 2.0 ** -9, 2.0 ** -10, 1.5 * 2.0 ** -10, 0.75 * 2.0 ** -9])]).astype(np.float32)
  x = torch.from_numpy(vals).cuda()
  out = torch.empty_like(x)
  H.lib.check(eng.lib.ixt_op_fp8_roundtrip(eng._ctx, _vp(x), _vp(out), x.numel(), _st(torch)))
  got = out.cpu().numpy().astype(np.float64)
  want = F8.e4m3_round(vals)
  bad = np.flatnonzero(got != want)
  assert bad.size == 0, (vals[bad[:8]], got[bad[:8]], want[bad[:8]])


# rows: the split-K kernel with 1, 2 and 4 row blocks per block (<= 64 rows share a weight fragment), two launches of row blocks (100), the ring
# GEMM (>= 128 rows) with ragged row groups; K: one chain, several, a short last chain (640, 1152 = 2 x 512 + 128)
@pytest.mark.parametrize("R,K,N", [(1, 256, 512), (8, 2048, 272), (16, 512, 64), (17, 640, 1008), (40, 1152, 256), (64, 2048, 128), (100, 1024, 144),
                                   (128, 1024, 256), (200, 2048, 144), (250, 1152, 256), (256, 640, 80)])
def test_fp8_dot_matches_the_restatement(ctx8, R, K, N):
  torch, eng = ctx8["torch"], ctx8["eng"]
  rng = np.random.default_rng(R * 1000 + K + N)
  x = H.rand_bf16(rng, (R, K))
  w = H.rand_bf16(rng, (K, N), 0.05)
  w[:, 3] = 0                                           # an all-zero column: scale 1, result 0
  if R > 1:
    x[R // 2, :] = 0                                    # an all-zero row: scale 1, result 0
  out = torch.empty((R, N), dtype=torch.float32, device="cuda")
  xd, wd = H.dev_bf16(x), H.dev_bf16(w)                 # (named: a temporary would be freed before the launch reads it)
  H.lib.check(eng.lib.ixt_op_dot_fp8(eng._ctx, _vp(xd), _vp(wd), _vp(out), R, K, N, _st(torch)))
  got = out.cpu().numpy().astype(np.float64)
  want, bound = F8.dot(x, w)
  err = np.abs(got - want)
  assert np.all(err <= 2e-5 * bound + 1e-30), f"max err / bound = {(err / np.maximum(bound, 1e-30)).max():.3e}"
  assert (R == 1 or not got[R // 2].any()) and not got[:, 3].any()
  # and the price of the scheme against the exact product of the bf16 operands: a few per cent of the row's magnitude
  exact = F8.bf16_to_f64(x) @ F8.bf16_to_f64(w)
  rel = np.linalg.norm(got - exact) / np.linalg.norm(exact)
  assert rel < 0.06, rel


def _decode_logits(eng, params, orc, prompts, steps):
  """Teacher-forced on the bf16 oracle's ids: prefill + `steps` decode steps on both; yields (fp8 engine logits, bf16 oracle logits) per step."""
  import torch
  B = len(prompts)
  ds, ods = eng.init_decode_state(materialize_logits=True), orc.init_decode_state()
  pairs = []
  for s, (t, tl) in enumerate(prompts):
    pre, _ = eng.prefill(params=params, padded_tokens=t, true_length=tl, slot=s)
    opre, ores = orc.prefill(t, tl, slot=s, want_logits=True)
    pairs.append((pre["logits"].reshape(-1).cpu().numpy(), opre["logits"].reshape(-1)))
    pre["tokens"][0, 0] = int(ores[0, 0])
    ds, ods = eng.insert(pre, ds, s), orc.insert(opre, ods, s)
  for _ in range(steps):
    ds, _ = eng.generate(params, ds)
    ods, orr = orc.generate(ods, want_logits=True)
    lg = ds["logits"].reshape(B, -1).cpu().numpy()
    for s in range(B):
      pairs.append((lg[s].copy(), np.asarray(ods["logits"])[s].reshape(-1)))
    ds["tokens"][:, 0] = torch.from_numpy(orr[:, 0].astype(np.int32)).to(ds["tokens"].device)     # same context on both sides next step
  return pairs


# model -> (mean relative L2 error of a logits row, its maximum, minimum cosine, share of rows whose bf16 top-1 id is in the fp8 top-3)
# measured (MI355X, these seeds): ixt-tiny 0.055 / 0.070 / 0.9976 / 1.0; gemma2-tiny 0.110 / 0.129 / 0.9917 / 1.0; mixtral-tiny 0.084 / 0.25 / 0.968 / 1.0
# (a mixture's worst rows are those where the quantisation noise of the layers before flips the router's choice of an expert)
BARS = {"ixt-tiny": (0.08, 0.10, 0.995, 0.9), "gemma2-tiny": (0.15, 0.18, 0.985, 0.9), "mixtral-tiny": (0.12, 0.35, 0.95, 0.9)}


@pytest.mark.parametrize("model,kw", [("ixt-tiny", {}), ("gemma2-tiny", {}), ("mixtral-tiny", {})])
def test_fp8_model_against_the_bf16_oracle(model, kw):
  """The fp8 engine and the bf16 oracle on the same weights, same contexts (teacher forcing): logits stay within a stated distance (BARS: measured values with margin)."""
  cfg8 = H.make_config(model, per_device_batch_size=2.0, max_prefill_predict_length=32, max_target_length=64, pagedattn_num_pages=32, quantization="fp8", **kw)
  cfg16 = H.make_config(model, per_device_batch_size=2.0, max_prefill_predict_length=32, max_target_length=64, pagedattn_num_pages=32, **kw)
  w = H.synth.synth_weights(cfg16, 5)
  eng, params = H.make_engine(cfg8, w)
  orc = H.make_oracle(cfg16, w)
  prompts = [H.synth.synth_prompt(cfg16.vocab_size, 32, 40), H.synth.synth_prompt(cfg16.vocab_size, 32, 41)]
  pairs = _decode_logits(eng, params, orc, prompts, 12)
  rels, coss, top3 = [], [], []
  for a, b in pairs:
    a, b = a.astype(np.float64), b.astype(np.float64)
    rels.append(np.linalg.norm(a - b) / np.linalg.norm(b))
    coss.append(float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b))))
    top3.append(int(np.argmax(b)) in np.argsort(a)[-3:])
  print(f"{model}: rel L2 max {max(rels):.4f} mean {np.mean(rels):.4f}; cosine min {min(coss):.5f}; top-1 in fp8 top-3: {np.mean(top3):.3f}")
  assert np.mean(rels) <= BARS[model][0] and max(rels) <= BARS[model][1] and min(coss) >= BARS[model][2] and np.mean(top3) >= BARS[model][3]


def test_fp8_split_k_and_ring_forms_agree_bit_for_bit():
  """A 160-token prompt in one chunk (>= 128 rows: the LDS-ring GEMM on e4m3 fragments) and in five 32-row chunks (the split-K kernel): the same MFMA
  sequence per 512-k chain and the same scales, so first token, logits and K / V are equal bit for bit."""
  import torch
  maxengine = H.maxengine_module()
  outs = []
  for chunk in (160, 32):
    cfg = H.make_config(per_device_batch_size=1.0, max_prefill_predict_length=chunk, max_target_length=192, pagedattn_num_pages=32, use_chunked_prefill=True,
                        prefill_chunk_size=chunk, quantization="fp8")
    w = H.synth.synth_weights(cfg, 9)
    eng, params = H.make_engine(cfg, w)
    toks = H.synth.synth_prompt(cfg.vocab_size, 160, 3)[0]
    done, pre, res = 0, None, None
    while done < 160:
      ep = None if done == 0 else maxengine.ExistingPrefix(cache=pre["cache"], common_prefix_tokens=toks[:done])
      pre, res = eng.prefill(params=params, padded_tokens=toks[done:done + chunk], true_length=chunk, slot=0, existing_prefix=ep)
      done += chunk
    k, v = eng.gather_kv(cfg.num_decoder_layers - 1, 0, 160)
    outs.append((int(res.data.cpu().numpy()[0, 0]), pre["logits"].cpu().numpy().reshape(-1).view(np.uint32).copy(), H.bits(k).copy(), H.bits(v).copy()))
    torch.cuda.synchronize()
  assert outs[0][0] == outs[1][0]
  for a, b in zip(outs[0][1:], outs[1][1:]):
    assert np.array_equal(a, b)
