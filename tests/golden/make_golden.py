#!/usr/bin/env python3
"""Generates the committed golden fixtures (run in the build container, where /root/reference is mounted):

  audio_token_mapping_adjusted.npz   the reference's audio-id <-> embedding-row table (data file
                                     /root/reference/audio_token_mapping_adjusted.json) in array form
  audio_token_mappings_other.npz     its two other mapping files (vocab_expansion/audio_token_mapping{,_complete}.json) in array form
  tiny_decode.npz / c1_decode.npz / c2_decode.npz
                                     greedy token ids produced by the CPU ORACLE (oracle/ixt_oracle.c) on seeded synthetic
                                     weights and prompts.  These pin the oracle against drift and give the GPU tests a
                                     full-size (C2, 1024-step) target without re-running the oracle on the GPU box.

The floating-point fixtures are oracle outputs, NOT reference outputs (the reference needs JAX, which is not installed:
parity unpinned).    c2_rows_130.npz / c2_rows_256.npz  C2 at the row counts the benchmark times (130 = two GEMM row groups, the second ragged; 256 = the
                                     bench default): ragged prompts, a few decode steps, token ids + per-row logits checksums
usage: make_golden.py [mapping] [tiny] [c1] [c2] [c2_rows_130] [c2_rows_256]
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402


def mapping():
  src = "/root/reference/audio_token_mapping_adjusted.json"
  raw = open(src, "rb").read()
  d = json.loads(raw)
  a2e = np.full((8192,), -1, np.int32)
  for k, v in d["audio_to_embedding"].items():
    a2e[int(k)] = int(v)
  rows = np.array(sorted(int(k) for k in d["embedding_to_audio"]), np.int32)
  aud = np.array([int(d["embedding_to_audio"][str(r)]) for r in rows], np.int32)
  np.savez_compressed(os.path.join(HERE, "audio_token_mapping_adjusted.npz"), audio_to_embedding=a2e, embedding_rows=rows, embedding_audio_id=aud,
                      stats=np.array(json.dumps(d["stats"])), sha256=np.array(hashlib.sha256(raw).hexdigest()))
  print("mapping:", d["stats"], hashlib.sha256(raw).hexdigest()[:8])


def mappings_other():
  """The reference's two other mapping data files in array form, with their sha256: vocab_expansion/audio_token_mapping.json
  (tokenizer index -> audio id, before the soft-token shift) and vocab_expansion/audio_token_mapping_complete.json."""
  out = {}
  src = "/root/reference/vocab_expansion/audio_token_mapping.json"
  raw = open(src, "rb").read()
  d = json.loads(raw)
  idx = np.array(sorted(int(k) for k in d["audio_mappings"]), np.int32)
  out["simple_tokenizer_index"] = idx
  out["simple_audio_id"] = np.array([int(d["audio_mappings"][str(i)]) for i in idx], np.int32)
  out["simple_vocab_size"] = np.array(int(d["vocab_size"]))
  out["simple_sha256"] = np.array(hashlib.sha256(raw).hexdigest())
  src = "/root/reference/vocab_expansion/audio_token_mapping_complete.json"
  raw = open(src, "rb").read()
  d = json.loads(raw)
  rows = np.array(sorted(int(k) for k in d["embedding_to_audio"]), np.int32)
  out["complete_embedding_rows"] = rows
  out["complete_embedding_audio_id"] = np.array([int(d["embedding_to_audio"][str(r)]) for r in rows], np.int32)
  a2e = np.full((8192,), -1, np.int32)
  for k, v in d["audio_to_embedding"].items():
    a2e[int(k)] = int(v)
  out["complete_audio_to_embedding"] = a2e
  tix = np.array(sorted(int(k) for k in d["tokenizer_to_audio"]), np.int32)
  out["complete_tokenizer_index"] = tix
  out["complete_tokenizer_audio_id"] = np.array([int(d["tokenizer_to_audio"][str(t)]) for t in tix], np.int32)
  out["complete_detail_original_token_idx"] = np.array([int(d["detailed_mappings"][str(r)]["original_token_idx"]) for r in rows], np.int32)
  out["complete_stats"] = np.array(json.dumps(d["stats"]))
  out["complete_sha256"] = np.array(hashlib.sha256(raw).hexdigest())
  np.savez_compressed(os.path.join(HERE, "audio_token_mappings_other.npz"), **out)
  print("mappings_other:", out["simple_sha256"].item()[:8], out["complete_sha256"].item()[:8], d["stats"])


def decode_fixture(name, cfg, seed_w, prompts, steps, B):
  from oracle import oracle as O
  t0 = time.time()
  w = H.synth.synth_weights(cfg, seed_w)
  print(name, "weights", time.time() - t0, flush=True)
  orc = H.make_oracle(cfg, w)
  firsts, oprefixes = [], []
  for s, (toks, tl) in enumerate(prompts):
    opre, ores = orc.prefill(toks, tl, slot=s)
    firsts.append(int(ores[0, 0]))
    oprefixes.append(opre)
  print(name, "prefill", time.time() - t0, flush=True)
  ods = orc.init_decode_state()
  for s in range(len(prompts)):
    ods = orc.insert(oprefixes[s], ods, s)
  out = np.zeros((steps + 1, B), np.int32)
  out[0, :len(prompts)] = firsts
  for i in range(steps):
    ods, r = orc.generate(ods)
    out[i + 1] = r[:, 0]
    if i % 64 == 0:
      print(name, "step", i, time.time() - t0, flush=True)
  np.savez_compressed(os.path.join(HERE, name + ".npz"), tokens=out, prompt_tokens=np.stack([p[0] for p in prompts]),
                      true_lengths=np.array([p[1] for p in prompts], np.int32), weight_seed=np.array(seed_w), steps=np.array(steps))
  print(name, "done", time.time() - t0, out[:6].tolist(), flush=True)


def tiny():
  cfg = H.make_config(per_device_batch_size=2.0)
  prompts = [H.synth.synth_prompt(cfg.vocab_size, 32, 1234), H.synth.synth_prompt(cfg.vocab_size, 32, 1235)]
  decode_fixture("tiny_decode", cfg, 0, prompts, 64, 2)


def c1():
  cfg = H.make_config(model_name="c1-134m", per_device_batch_size=1.0, max_prefill_predict_length=64, max_target_length=128,
                      pagedattn_tokens_per_page=32, pagedattn_num_pages=8)
  prompts = [H.synth.synth_prompt(cfg.vocab_size, 64, 1234)]
  decode_fixture("c1_decode", cfg, 0, prompts, 64, 1)


def c2():
  cfg = H.make_config(model_name="indextts2-1b", per_device_batch_size=1.0, max_prefill_predict_length=256, max_target_length=1280,
                      pagedattn_tokens_per_page=32, pagedattn_num_pages=41)
  prompts = [H.synth.synth_prompt(cfg.vocab_size, 256, 1234)]
  decode_fixture("c2_decode", cfg, 0, prompts, 1024, 1)


def logits_checksum(lg):
  """Order-independent checksum of one row of f32 logits: sum of the 32-bit patterns (mod 2^64)."""
  return np.ascontiguousarray(lg, np.float32).view(np.uint32).astype(np.uint64).sum(axis=-1)


def rows_lengths(B, long_rows, hi, seed):
  """Ragged prompt lengths: the first rows long (page crossings, many keys), the rest 1..hi tokens."""
  rng = np.random.Generator(np.random.PCG64(seed))
  lens = rng.integers(1, hi + 1, size=B).astype(np.int32)
  lens[:len(long_rows)] = long_rows
  return lens


def c2_rows(name, B, long_rows, hi, steps, seed):
  """C2 at the benchmarked row counts: B utterances (ragged prompts, prefilled as ONE ragged batch of rows by the oracle),
  then `steps` decode steps of all B rows.  Stores token ids and a checksum of every row's f32 logits."""
  T = 320
  cfg = H.make_config(model_name="indextts2-1b", per_device_batch_size=float(B), max_prefill_predict_length=256, max_target_length=T,
                      pagedattn_tokens_per_page=32, pagedattn_num_pages=B * (T // 32) + 1)
  t0 = time.time()
  w = H.synth.synth_weights(cfg, 0)
  print(name, "weights", time.time() - t0, flush=True)
  orc = H.make_oracle(cfg, w)
  lens = rows_lengths(B, long_rows, hi, seed)
  prompts = []
  for s in range(B):
    toks, _ = H.synth.synth_prompt(cfg.vocab_size, 256, 5000 + seed * 1000 + s)
    prompts.append((toks, int(lens[s])))
  pres, firsts, lg = orc.prefill_many(prompts, want_logits=True)
  print(name, "prefill", int(lens.sum()), "rows", time.time() - t0, flush=True)
  tokens = np.zeros((steps + 1, B), np.int32)
  sums = np.zeros((steps + 1, B), np.uint64)
  tokens[0] = firsts
  sums[0] = logits_checksum(lg)
  ods = orc.init_decode_state()
  for s in range(B):
    ods = orc.insert(pres[s], ods, s)
  for i in range(steps):
    ods, r = orc.generate(ods, want_logits=True)
    tokens[i + 1] = r[:, 0]
    sums[i + 1] = logits_checksum(ods["logits"])
    print(name, "step", i, time.time() - t0, flush=True)
  np.savez_compressed(os.path.join(HERE, name + ".npz"), tokens=tokens, logits_checksum=sums, prompt_tokens=np.stack([p[0] for p in prompts]),
                      true_lengths=lens, weight_seed=np.array(0), steps=np.array(steps), max_target_length=np.array(T))
  print(name, "done", time.time() - t0, tokens[:, :6].tolist(), flush=True)


def c2_rows_130():
  c2_rows("c2_rows_130", 130, [250, 131, 97, 64, 33], 12, 4, 1)


def c2_rows_256():
  c2_rows("c2_rows_256", 256, [200, 40, 31], 8, 2, 2)


if __name__ == "__main__":
  which = sys.argv[1:] or ["mapping", "tiny", "c1"]
  for w in which:
    {"mapping": mapping, "mappings_other": mappings_other, "tiny": tiny, "c1": c1, "c2": c2, "c2_rows_130": c2_rows_130, "c2_rows_256": c2_rows_256}[w]()
