"""Shared test plumbing: configs, synthetic weights, oracle + HIP engines built on the same inputs."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

ixt = importlib.import_module("maxtext-indextts2_amd")
pyconfig = importlib.import_module("maxtext-indextts2_amd.pyconfig")
synth = importlib.import_module("maxtext-indextts2_amd.synth")
lib = importlib.import_module("maxtext-indextts2_amd._lib")
BASE_YML = os.path.join(pyconfig.PKG_DIR, "configs", "base.yml")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def make_config(model_name="ixt-tiny", **overrides):
  kw = dict(model_name=model_name, attention="paged", scan_layers=False, per_device_batch_size=1.0,
            max_prefill_predict_length=32, max_target_length=128, pagedattn_num_pages=64, pagedattn_tokens_per_page=8,
            enable_checkpointing=False, run_name="test",
            override_model_config=True)     # the tests resize the named models (base.yml:20: the reference's switch for exactly that)
  kw.update(overrides)
  return pyconfig.initialize(["test", BASE_YML], **kw)


def oracle_config(cfg):
  from oracle import oracle as O
  return O.OracleConfig(
      vocab_size=cfg.vocab_size, emb_dim=cfg.emb_dim, num_query_heads=cfg.num_query_heads, num_kv_heads=cfg.num_kv_heads,
      head_dim=cfg.head_dim, mlp_dim=cfg.mlp_dim, num_decoder_layers=cfg.num_decoder_layers,
      normalization_layer_epsilon=cfg.normalization_layer_epsilon, rope_min_timescale=float(cfg.rope_min_timescale),
      rope_max_timescale=float(cfg.rope_max_timescale), max_target_length=cfg.max_target_length,
      max_prefill_predict_length=cfg.max_prefill_predict_length, pagedattn_num_pages=cfg.pagedattn_num_pages,
      pagedattn_tokens_per_page=cfg.pagedattn_tokens_per_page, pagedattn_max_pages_per_group=cfg.pagedattn_max_pages_per_group,
      max_slots=int(cfg.per_device_batch_size), decoder_block=cfg.decoder_block if cfg.decoder_block in ("gemma2", "gemma3", "mixtral") else "llama2",
      num_experts=int(cfg.num_experts), num_experts_per_tok=int(cfg.num_experts_per_tok),
      attn_logits_soft_cap=float(cfg.attn_logits_soft_cap or 0.0), final_logits_soft_cap=float(cfg.final_logits_soft_cap or 0.0),
      sliding_window_size=int(cfg.sliding_window_size or 0), use_post_attn_norm=bool(cfg.use_post_attn_norm),
      use_post_ffw_norm=bool(cfg.use_post_ffw_norm), logits_via_embedding=bool(cfg.logits_via_embedding),
      normalize_embedding_logits=bool(cfg.normalize_embedding_logits),
      local_rope_max_timescale=float(cfg.local_rope_max_timescale), rope_type=str(cfg.rope_type), rope_use_scale=bool(cfg.rope_use_scale),
      query_pre_attn_scalar=float(maxengine_module().get_query_pre_attn_scalar(cfg)) if cfg.decoder_block == "gemma3" else 0.0)


def maxengine_module():
  return importlib.import_module("maxtext-indextts2_amd.maxengine")


def sublayers(cfg):
  """(attention, MLP) sub-layers = K/V page pools: two per merged [local, global] layer of a Gemma-2 config."""
  return cfg.num_decoder_layers * (2 if cfg.decoder_block == "gemma2" else 1)


def make_oracle(cfg, weights):
  from oracle import oracle as O
  model = O.OracleModel(oracle_config(cfg), weights)
  return O.OracleEngine(model)


def make_engine(cfg, weights):
  maxengine = importlib.import_module("maxtext-indextts2_amd.maxengine")
  eng = maxengine.MaxEngine(cfg)
  params = eng.load_params(params=weights)
  return eng, params


def free_port():
  import socket
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    return s.getsockname()[1]


def bits(t):
  """torch bf16 tensor -> numpy uint16 bit patterns."""
  import torch
  return t.detach().contiguous().view(torch.int16).cpu().numpy().view(np.uint16)


def dev_bf16(a):
  """numpy uint16 bit patterns -> torch bf16 tensor on cuda."""
  import torch
  return torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).cuda().view(torch.bfloat16)


def rand_bf16(rng, shape, scale=1.0):
  return synth.f32_to_bf16_bits((rng.standard_normal(int(np.prod(shape)), dtype=np.float32) * np.float32(scale))).reshape(shape)
