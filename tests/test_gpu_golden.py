"""GPU parity against the committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py with the
CPU oracle): full-size C2 (1B, 256-token prefill -> 1024 greedy steps), C1 (134M) and the tiny model, plus
size-independent properties at full size (batch rows independent, teacher-forced prefill == decode)."""
import os

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def run_engine(cfg, w, prompts, steps):
  eng, params = H.make_engine(cfg, w)
  pres, firsts = [], []
  for s, (t, tl) in enumerate(prompts):
    p, r = eng.prefill(params=params, padded_tokens=t, true_length=tl, slot=s)
    pres.append(p)
    firsts.append(r.data)
  ds = eng.init_decode_state()
  for s in range(len(prompts)):
    ds = eng.insert(pres[s], ds, s)
  res = []
  for _ in range(steps):
    ds, r = eng.generate(params, ds)
    res.append(r.data)
  B = int(cfg.per_device_batch_size)
  out = np.zeros((steps + 1, B), np.int32)
  for s in range(len(prompts)):
    out[0, s] = int(firsts[s][0, 0])
  import torch
  out[1:] = torch.stack(res)[:, :, 0].cpu().numpy()
  return out, eng, params


def test_golden_tiny():
  g = np.load(os.path.join(H.GOLDEN, "tiny_decode.npz"))
  cfg = H.make_config(per_device_batch_size=2.0)
  w = H.synth.synth_weights(cfg, int(g["weight_seed"]))
  prompts = [(g["prompt_tokens"][i], int(g["true_lengths"][i])) for i in range(2)]
  out, _, _ = run_engine(cfg, w, prompts, int(g["steps"]))
  assert np.array_equal(out, g["tokens"])


def test_golden_c1():
  g = np.load(os.path.join(H.GOLDEN, "c1_decode.npz"))
  cfg = H.make_config(model_name="c1-134m", per_device_batch_size=1.0, max_prefill_predict_length=64, max_target_length=128,
                      pagedattn_tokens_per_page=32, pagedattn_num_pages=8)
  w = H.synth.synth_weights(cfg, int(g["weight_seed"]))
  out, _, _ = run_engine(cfg, w, [(g["prompt_tokens"][0], int(g["true_lengths"][0]))], int(g["steps"]))
  assert np.array_equal(out, g["tokens"])


@pytest.fixture(scope="module")
def c2_weights():
  path = os.path.join(H.GOLDEN, "c2_decode.npz")
  if not os.path.exists(path):
    pytest.skip("c2 golden fixture not generated")
  g = np.load(path)
  cfg = H.make_config(model_name="indextts2-1b", per_device_batch_size=1.0, max_prefill_predict_length=256, max_target_length=1280,
                      pagedattn_tokens_per_page=32, pagedattn_num_pages=41)
  return g, H.synth.synth_weights(cfg, int(g["weight_seed"]))


def _checksum(t):
  """sum of the 32-bit patterns of each row of f32 logits (make_golden.logits_checksum), on the GPU."""
  import torch
  return (t.contiguous().view(torch.int32).to(torch.int64) & 0xffffffff).sum(dim=-1).cpu().numpy().astype(np.uint64)


@pytest.mark.parametrize("name,packed", [("c2_rows_130", False), ("c2_rows_256", False), ("c2_rows_256", True)])
def test_golden_c2_rows(c2_weights, name, packed):
  """C2 at the row counts the benchmark times, against committed oracle output (tests/golden/make_golden.py c2_rows_*): 130 utterances
  = two 128-row GEMM groups, the second ragged; 256 = the bench default.  Ragged prompts (a few long ones: page crossings, hundreds of
  keys), then decode steps of all rows through the kernels the bench runs at these shapes (ring GEMMs with the QKV / residual / SwiGLU /
  sampling epilogues at 2048 x 6144, 2048 x 2048, 2048 x 14336, 7168 x 2048, 2048 x 264192; one-launch decode attention at >= 128 x 16
  pairs).  Token ids and a checksum of every row's f32 logits must match.  packed: the prompts go through prefill_concat /
  insert_partial (one pass over the weights for all of them) instead of one prefill per utterance."""
  import torch
  g = np.load(os.path.join(H.GOLDEN, name + ".npz"))
  lens, steps, T = g["true_lengths"], int(g["steps"]), int(g["max_target_length"])
  B = lens.shape[0]
  total = int(lens.sum())
  cfg = H.make_config(model_name="indextts2-1b", per_device_batch_size=float(B), max_prefill_predict_length=256, max_target_length=T,
                      pagedattn_tokens_per_page=32, pagedattn_num_pages=(2 if packed else 1) * B * (T // 32) + 1,
                      ixt_staging_groups=B if packed else 0, ixt_max_packed_rows=total if packed else 0)
  assert int(g["weight_seed"]) == int(c2_weights[0]["weight_seed"])
  eng, params = H.make_engine(cfg, c2_weights[1])
  ds = eng.init_decode_state(materialize_logits=True)
  tokens = np.zeros((steps + 1, B), np.int32)
  sums = np.zeros((steps + 1, B), np.uint64)
  if packed:
    toks = np.concatenate([g["prompt_tokens"][s][:lens[s]] for s in range(B)]).astype(np.int32)
    pos = np.concatenate([np.arange(lens[s]) for s in range(B)]).astype(np.int32)
    seg = np.concatenate([np.full(lens[s], s + 1) for s in range(B)]).astype(np.int32)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int32)
    cache, pres, results = eng.prefill_concat(params=params, padded_tokens=toks, decoder_positions=pos, decoder_segment_ids=seg, start_pos=starts,
                                              true_lengths=lens, num_prompts=B)
    tokens[0] = pres["tokens"].reshape(B).cpu().numpy()
    sums[0] = _checksum(pres["logits"].reshape(B, -1))
    ds = eng.insert_partial(pres, ds, cache, np.arange(B), start_indices=starts, num_prompts=B, seq_len=256)
  else:
    for s in range(B):
      pre, res = eng.prefill(params=params, padded_tokens=g["prompt_tokens"][s], true_length=int(lens[s]), slot=s)
      tokens[0, s] = int(res.data[0, 0])
      sums[0, s] = _checksum(pre["logits"].reshape(1, -1))[0]
      ds = eng.insert(pre, ds, s)
  for i in range(steps):
    ds, res = eng.generate(params, ds)
    tokens[i + 1] = res.data[:, 0].cpu().numpy()
    sums[i + 1] = _checksum(ds["logits"].reshape(B, -1))
  bad = np.argwhere(tokens != g["tokens"])
  assert bad.size == 0, f"{len(bad)} token ids differ, first (step, row) {bad[:5].tolist()}"
  bad = np.argwhere(sums != g["logits_checksum"])
  assert bad.size == 0, f"{len(bad)} logits checksums differ, first (step, row) {bad[:5].tolist()}"
  del eng
  torch.cuda.synchronize()


def test_c2_packed_4096_rows_equals_two_halves(c2_weights):
  """The bench's packed prefill shape (16 prompts x 256 tokens = 4096 rows in one pass: every layer GEMM -- also O and MLP down with the residual
  epilogue -- in the two-blocks-per-CU ring form) against the same prompts in two passes of 2048 rows (the 64-k chunk form for O / MLP down, whose
  fixtures pin it to the oracle): first tokens, f32 logits and the K/V of two layers must be identical bit for bit."""
  import torch
  n, L = 16, 256
  cfg = H.make_config(model_name="indextts2-1b", per_device_batch_size=float(n), max_prefill_predict_length=L, max_target_length=L + 32,
                      pagedattn_tokens_per_page=32, pagedattn_num_pages=2 * n * 9 + 1, ixt_staging_groups=n, ixt_max_packed_rows=n * L)
  eng, params = H.make_engine(cfg, c2_weights[1])
  prompts = [(H.synth.synth_prompt(cfg.vocab_size, L, 8100 + s)[0], L) for s in range(n)]

  def run(groups):
    firsts, logits, kv = [], [], []
    ds = eng.init_decode_state()
    slot0 = 0
    for grp in groups:
      t, pos, seg, starts, ln = _packed(grp)
      cache, pres, _ = eng.prefill_concat(params=params, padded_tokens=t, decoder_positions=pos, decoder_segment_ids=seg, start_pos=starts, true_lengths=ln,
                                          num_prompts=len(grp))
      firsts.append(pres["tokens"].reshape(-1).cpu().numpy().copy())
      logits.append(pres["logits"].reshape(len(grp), -1).cpu().numpy().view(np.uint32).copy())
      slots = np.arange(slot0, slot0 + len(grp))
      ds = eng.insert_partial(pres, ds, cache, slots, start_indices=starts, num_prompts=len(grp), seq_len=L)
      slot0 += len(grp)
    for s in (0, n - 1):
      for layer in (0, cfg.num_decoder_layers - 1):
        kv.append(tuple(H.bits(x).copy() for x in eng.gather_kv(layer, s, L)))
    for s in range(n):
      eng.release_pages(s)
    return np.concatenate(firsts), np.concatenate(logits), kv
  a = run([prompts])
  b = run([prompts[:8], prompts[8:]])
  assert np.array_equal(a[0], b[0])
  assert np.array_equal(a[1], b[1])
  for (ka, va), (kb, vb) in zip(a[2], b[2]):
    assert np.array_equal(ka, kb) and np.array_equal(va, vb)
  del eng
  torch.cuda.synchronize()


def _packed(prompts):
  lens = np.array([tl for _, tl in prompts], np.int32)
  toks = np.concatenate([np.asarray(t, np.int32)[:tl] for t, tl in prompts])
  pos = np.concatenate([np.arange(tl) for tl in lens]).astype(np.int32)
  seg = np.concatenate([np.full(tl, i + 1) for i, tl in enumerate(lens)]).astype(np.int32)
  starts = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int32)
  return toks, pos, seg, starts, lens


def test_golden_c2_full_decode(c2_weights):
  """BASELINE.json configs[1] at full size: every one of the 1025 token ids equals the oracle's."""
  g, w = c2_weights
  cfg = H.make_config(model_name="indextts2-1b", per_device_batch_size=1.0, max_prefill_predict_length=256, max_target_length=1280,
                      pagedattn_tokens_per_page=32, pagedattn_num_pages=41)
  out, _, _ = run_engine(cfg, w, [(g["prompt_tokens"][0], int(g["true_lengths"][0]))], int(g["steps"]))
  assert np.array_equal(out, g["tokens"]), int(np.argmax(np.any(out != g["tokens"], axis=1)))


@pytest.mark.parametrize("B", [8, 20])
def test_c2_batch_properties(c2_weights, B):
  """Full-size, batch 8 (vector-ALU dot kernel) and batch 20 (matrix-core dot kernel, two 16-row blocks per thread block):
  slot 0 carries the golden utterance and must reproduce it whatever the other slots hold;
  two slots with the same prompt give the same tokens; teacher-forced chunked prefill of prompt+generated tokens
  reproduces the decode path's next token (prefill == decode, tests/model_test.py:115-191 property)."""
  g, w = c2_weights
  cfg = H.make_config(model_name="indextts2-1b", per_device_batch_size=float(B), max_prefill_predict_length=256, max_target_length=1280,
                      pagedattn_tokens_per_page=32, pagedattn_num_pages=B * 40 + 1, use_chunked_prefill=True, prefill_chunk_size=256)
  p0 = (g["prompt_tokens"][0], int(g["true_lengths"][0]))
  prompts = [p0] + [H.synth.synth_prompt(cfg.vocab_size, 256, 2000 + s) for s in range(1, 7)] + [p0]
  steps = 300 if B == 8 else 160
  out, eng, params = run_engine(cfg, w, prompts, steps)
  assert np.array_equal(out[:, 0], g["tokens"][:steps + 1, 0])
  assert np.array_equal(out[:, 7], out[:, 0])
  # teacher forcing through chunked prefill into slot 1 (released first): prompt + first 299 generated tokens
  maxengine = H.importlib.import_module("maxtext-indextts2_amd.maxengine")
  eng.release_pages(1)
  seq = np.concatenate([p0[0][:p0[1]], out[:steps, 0]]).astype(np.int32)
  done, pre = 0, None
  while done < len(seq):
    n = min(256, len(seq) - done)
    chunk = np.zeros((256,), np.int32)
    chunk[:n] = seq[done:done + n]
    ep = None if done == 0 else maxengine.ExistingPrefix(cache=pre["cache"], common_prefix_tokens=seq[:done])
    pre, res = eng.prefill(params=params, padded_tokens=chunk, true_length=n, slot=1, existing_prefix=ep)
    done += n
  assert int(res.data[0, 0]) == int(out[steps, 0])


def test_decode_harness_reproduces_tiny_golden(capsys):
  """The CLI harness with the reference's call order (MaxText/decode.py:85-184 -> maxtext-indextts2_amd/decode.py): the seeded
  synthetic prompt (seed 1234 = slot 0 of the tiny golden), random-init weights of seed 0, 64 generate calls; the printed ids are
  the golden's slot 0, and with audio-only sampling every id maps to an audio code."""
  import re
  g = np.load(os.path.join(H.GOLDEN, "tiny_decode.npz"))
  decode = H.importlib.import_module("maxtext-indextts2_amd.decode")
  argv = ["decode", H.BASE_YML, "model_name=ixt-tiny", "attention=paged", "scan_layers=false", "per_device_batch_size=2",
          "max_prefill_predict_length=32", "max_target_length=96", "pagedattn_num_pages=64", "pagedattn_tokens_per_page=8",
          "ixt_synthetic_seed=0", "autoregressive_decode_assert=" + ",".join(str(int(t)) for t in g["tokens"][:4, 0])]
  decode.main(argv)
  out = capsys.readouterr().out
  ids = [int(t) for t in re.search(r"generated ids: \[(.*?)\]", out).group(1).split(",")]
  assert ids == [int(t) for t in g["tokens"][:65, 0]]
  # audio-only sampling through the harness: ranges from a mapping file in the array form of the reference's JSON
  VE = H.importlib.import_module("maxtext-indextts2_amd.vocab_expansion")
  m = VE.AudioTokenMapping.from_rule(num_audio=600, low_slots=99, low_base=6, high_base=400, padded_rows_end=1024)   # a table that fits the 1024-row tiny vocabulary
  import tempfile
  with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "map.npz")
    np.savez(path, audio_to_embedding=m.audio_to_embedding, embedding_rows=m.embedding_rows, embedding_audio_id=m.embedding_audio_id)
    decode.main(argv[:-1] + ["ixt_audio_only_sampling=true", "audio_token_mapping_path=" + path])
  out = capsys.readouterr().out
  codes = [int(t) for t in re.search(r"audio codes: \[(.*?)\]", out).group(1).split(",")]
  assert len(codes) == 65 and min(codes) >= 0 and max(codes) < 600


def test_expanded_checkpoint_end_to_end(tmp_path):
  """SURVEY 8(f) item 1 through the product path: a flat checkpoint is extended by `expand_embedder` (embedding rows + logits
  columns, padded to the GEMM tile), saved as .npz, loaded by MaxEngine.load_params() via `load_parameters_path`, and decodes
  the same ids as the oracle on the same expanded weights; sampling restricted to the appended rows picks only appended rows."""
  VE = H.importlib.import_module("maxtext-indextts2_amd.vocab_expansion")
  maxengine = H.importlib.import_module("maxtext-indextts2_amd.maxengine")
  base = H.make_config(per_device_batch_size=2.0)
  w = H.synth.synth_weights(base, 9)
  wx = VE.expand_embedder(w, num_tokens=40)
  V2 = wx["params/token_embedder/embedding"].shape[0]
  assert V2 == 1072 and wx["params/decoder/logits_dense/kernel"].shape == (base.emb_dim, V2)
  path = str(tmp_path / "expanded.npz")
  np.savez(path, **wx)
  cfg = H.make_config(per_device_batch_size=2.0, vocab_size=V2, load_parameters_path=path)
  eng = maxengine.MaxEngine(cfg)
  params = eng.load_params()
  orc = H.make_oracle(cfg, wx)
  toks, tl = H.synth.synth_prompt(base.vocab_size, 32, 77)
  new_rows = [(base.vocab_size, base.vocab_size + 40)]
  for ranges in ([], new_rows):
    eng.set_sampling_ranges(ranges)
    orc.set_sampling_ranges(ranges)
    try:
      pre, r0 = eng.prefill(params=params, padded_tokens=toks, true_length=tl, slot=0)
      opre, or0 = orc.prefill(toks, tl, slot=0)
      ds = eng.insert(pre, eng.init_decode_state(), 0)
      ods = orc.insert(opre, orc.init_decode_state(), 0)
      got, want = [int(r0.data[0, 0])], [int(or0[0, 0])]
      for _ in range(16):
        ds, r = eng.generate(params, ds)
        ods, orr = orc.generate(ods)
        got.append(int(r.data[0, 0]))
        want.append(int(orr[0, 0]))
      assert got == want
      if ranges:
        assert all(base.vocab_size <= t < base.vocab_size + 40 for t in got)
      eng.release_pages(0)
      orc.release_pages(0)
    finally:
      orc.set_sampling_ranges([])


@pytest.mark.skipif(os.environ.get("IXT_SOAK") != "1", reason="slow (about 10 minutes of oracle time): set IXT_SOAK=1")
def test_c2_many_rows_soak(c2_weights):
  """Full-size model, 130 utterances (two GEMM row groups, the second one 2 rows): tiled GEMM kernels (full-K, split, sampling
  epilogue) at the real matrix shapes against the oracle, 10 decode steps after ragged short prompts."""
  g, w = c2_weights
  B = 130
  cfg = H.make_config(model_name="indextts2-1b", per_device_batch_size=float(B), max_prefill_predict_length=64, max_target_length=128,
                      pagedattn_tokens_per_page=32, pagedattn_num_pages=B * 4 + 1)
  eng, params = H.make_engine(cfg, w)
  orc = H.make_oracle(cfg, w)
  firsts, pres, opres = [], [], []
  for s in range(B):
    toks, _ = H.synth.synth_prompt(cfg.vocab_size, 64, 5000 + s)
    tl = 8 + (s * 7) % 40
    pre, r = eng.prefill(params=params, padded_tokens=toks, true_length=tl, slot=s)
    opre, orr = orc.prefill(toks, tl, slot=s)
    assert int(r.data[0, 0]) == int(orr[0, 0]), s
    pres.append(pre)
    opres.append(opre)
    if s % 10 == 0:
      print("soak: prefilled slot", s, flush=True)   # (run with -s: a silent GPU job is taken to be hung)
  ds, ods = eng.init_decode_state(), orc.init_decode_state()
  for s in range(B):
    ds = eng.insert(pres[s], ds, s)
    ods = orc.insert(opres[s], ods, s)
  for i in range(10):
    ds, r = eng.generate(params, ds)
    ods, orr = orc.generate(ods)
    assert np.array_equal(r.data.cpu().numpy(), orr), i
    print("soak: decode step", i, "ok", flush=True)
