"""Append token rows to a checkpoint (reference: vocab_expansion/expand_embedder.py:36-90).

The reference restores an Orbax checkpoint, concatenates `num_tokens` rows of truncated-normal(+-2 sigma) * 0.02 drawn from
jax.random.PRNGKey(67) to `token_embedder/embedding` (:50-63) and saves it again; it leaves `decoder/logits_dense/kernel`
alone, which breaks an untied head (the logits kernel keeps the old vocabulary).  This build works on the flat
{MaxText tree path: array} form its loader reads (.safetensors / .npz), extends BOTH tables when the head is untied, and
pads the new vocabulary to a multiple of 16 columns (the GEMM tile).  JAX's threefry stream cannot be reproduced here (JAX is
not installed), so the new rows come from numpy's PCG64 with the same distribution and the same seed number: a documented
deviation in the VALUES of randomly initialised rows, not in the layout.
"""
from __future__ import annotations

import os
from typing import Dict

import numpy as np

EMBEDDING = "params/token_embedder/embedding"
LOGITS = "params/decoder/logits_dense/kernel"
_ALT = {"token_embedder/embedding": EMBEDDING, "decoder/logits_dense/kernel": LOGITS}


def _trunc_normal(rng: np.random.Generator, shape, std: float) -> np.ndarray:
  """Truncated normal on [-2, 2] standard deviations, scaled by std (expand_embedder.py:11-14), by rejection."""
  n = int(np.prod(shape))
  out = np.empty((n,), np.float64)
  filled = 0
  while filled < n:
    x = rng.standard_normal(max(1024, int((n - filled) * 1.1)))
    x = x[np.abs(x) <= 2.0]
    k = min(n - filled, x.size)
    out[filled:filled + k] = x[:k]
    filled += k
  return (out * std).astype(np.float32).reshape(shape)


def _f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
  u = np.ascontiguousarray(x, np.float32).view(np.uint32)
  return ((u + (0x7FFF + ((u >> 16) & 1))) >> 16).astype(np.uint16)


def _like(new_f32: np.ndarray, ref: np.ndarray) -> np.ndarray:
  """New values in the storage type of the table they extend: uint16 = bf16 bit patterns, float32 stays float32."""
  if ref.dtype == np.uint16:
    return _f32_to_bf16_bits(new_f32)
  if ref.dtype == np.float32:
    return new_f32
  raise ValueError(f"unsupported table dtype {ref.dtype} (float32 or uint16 bf16 bit patterns)")


def expand_embedder(params: Dict[str, np.ndarray], num_tokens: int = 2048, std: float = 0.02, seed: int = 67,
                    pad_to_multiple: int = 16) -> Dict[str, np.ndarray]:
  """Returns a new parameter dict with `num_tokens` rows appended to the embedding (plus zero rows up to a multiple of
  `pad_to_multiple`), and the same number of columns appended to the untied logits kernel when it is present.
  Every other entry is passed through untouched."""
  if num_tokens <= 0:
    raise ValueError("num_tokens must be positive")
  out = dict(params)
  key = EMBEDDING if EMBEDDING in params else next((k for k in params if _ALT.get(k) == EMBEDDING), None)
  if key is None:
    raise KeyError("token_embedder/embedding not in the checkpoint")
  emb = np.asarray(params[key])
  old_vocab, dim = emb.shape
  new_vocab = old_vocab + num_tokens
  padded = (new_vocab + pad_to_multiple - 1) // pad_to_multiple * pad_to_multiple
  rng = np.random.Generator(np.random.PCG64(seed))
  rows = _trunc_normal(rng, (num_tokens, dim), std)
  rows = np.concatenate([rows, np.zeros((padded - new_vocab, dim), np.float32)], axis=0)
  out[key] = np.concatenate([emb, _like(rows, emb)], axis=0)
  lkey = LOGITS if LOGITS in params else next((k for k in params if _ALT.get(k) == LOGITS), None)
  if lkey is not None:
    w = np.asarray(params[lkey])
    if w.shape != (dim, old_vocab):
      raise ValueError(f"logits kernel {w.shape} does not match embedding {emb.shape}")
    cols = _trunc_normal(rng, (dim, num_tokens), std)
    cols = np.concatenate([cols, np.zeros((dim, padded - new_vocab), np.float32)], axis=1)
    out[lkey] = np.concatenate([w, _like(cols, w)], axis=1)
  return out


def _load(path: str) -> Dict[str, np.ndarray]:
  if path.endswith(".safetensors"):
    from safetensors.numpy import load_file  # pylint: disable=import-outside-toplevel
    return dict(load_file(path))
  if path.endswith(".npz"):
    with np.load(path) as z:
      return {k: z[k] for k in z.files}
  raise ValueError(f"{path}: expected .safetensors or .npz")


def _save(path: str, params: Dict[str, np.ndarray]) -> None:
  tmp = path + ".tmp"
  if path.endswith(".safetensors"):
    from safetensors.numpy import save_file  # pylint: disable=import-outside-toplevel
    save_file({k: np.ascontiguousarray(v) for k, v in params.items()}, tmp)
  elif path.endswith(".npz"):
    with open(tmp, "wb") as f:
      np.savez(f, **params)
  else:
    raise ValueError(f"{path}: expected .safetensors or .npz")
  os.replace(tmp, path)


def expand_checkpoint_file(in_path: str, out_path: str, num_tokens: int = 2048, std: float = 0.02, seed: int = 67) -> Dict[str, tuple]:
  """File form of expand_embedder (the reference's `add_embeddings` CLI, expand_embedder.py:92-104).  Returns the new shapes."""
  new = expand_embedder(_load(in_path), num_tokens=num_tokens, std=std, seed=seed)
  _save(out_path, new)
  return {k: tuple(v.shape) for k, v in new.items() if k.endswith("embedding") or k.endswith("logits_dense/kernel")}
