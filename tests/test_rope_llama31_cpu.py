"""rope_type "llama3.1" in the oracle against the cases the reference's tests hold for LLaMARotaryEmbedding (tests/llama_test.py:105-160) and against
the complex-number formulation the reference tests its default RoPE with (tests/llama_test.py:36-103)."""
import ctypes

import numpy as np

import helpers as H
from oracle import oracle as O


def _rotate(x_bits, pos, sn, cs):
  """x_bits uint16 [rows][heads][hd] -> orc_rope_llama31 per row (in place on a copy)."""
  out = x_bits.copy()
  lib = O.lib()
  lib.orc_rope_llama31.argtypes = [O._u16p, ctypes.c_int, ctypes.c_int, O._f32p, O._f32p]
  for r in range(out.shape[0]):
    row = np.ascontiguousarray(out[r])
    lib.orc_rope_llama31(O._p(row, O._u16p), out.shape[1], out.shape[2], O._p(np.ascontiguousarray(sn[pos[r]]), O._f32p), O._p(np.ascontiguousarray(cs[pos[r]]), O._f32p))
    out[r] = row
  return out


def _inputs(seed=0, seq=8, heads=4, hd=128):
  rng = np.random.default_rng(seed)
  x = rng.normal(1, 0.5, (seq, heads, hd)).astype(np.float32)           # llama_test.py:86-88
  return O.f32_to_bf16(x).reshape(seq, heads, hd), np.arange(seq)


def test_llama_rope_single_rotation_formula():
  """llama_test.py:132-160: outputs == inputs * cos + inputs_shifted * sin * sign with the pair (2i, 2i + 1) sharing a timescale, no scaling."""
  xb, pos = _inputs(1)
  sn, cs = O.rope_tables_llama31(128, 8, use_scale=False)
  got = O.bf16_to_f32(_rotate(xb, pos, sn, cs))
  x = O.bf16_to_f32(xb)
  left = np.concatenate([x[..., 1:], x[..., :1]], -1)
  right = np.concatenate([x[..., -1:], x[..., :-1]], -1)
  shifted = np.where(np.arange(128) % 2 == 1, right, left)
  frac = np.repeat(2 * np.arange(64) / 128, 2)
  timescale = 1.0 * (10000.0 / 1.0) ** frac
  ang = pos[:, None, None] / timescale[None, None, :]
  want = x * np.cos(ang) + shifted * np.sin(ang) * np.tile([-1.0, 1.0], 64)
  assert np.allclose(got, want, rtol=1e-2, atol=1e-2)                    # llama_test.py:159 (bf16 outputs against an f64 formula)


def test_llama_rope_equals_the_complex_rotation():
  """llama_test.py:36-103's reference implementation (complex multiply of neighbouring pairs) -- for the interleaved form no permutation is needed."""
  xb, pos = _inputs(2)
  sn, cs = O.rope_tables_llama31(128, 8, use_scale=False)
  got = O.bf16_to_f32(_rotate(xb, pos, sn, cs))
  x = O.bf16_to_f32(xb).astype(np.float64)
  freqs = 1.0 / (10000.0 ** (np.arange(0, 128, 2) / 128))
  cis = np.exp(1j * np.outer(pos, freqs))[:, None, :]
  z = (x[..., 0::2] + 1j * x[..., 1::2]) * cis
  want = np.stack([z.real, z.imag], -1).reshape(x.shape)
  assert np.allclose(got, want, rtol=1e-1, atol=1e-2)                    # llama_test.py:102-103 tolerances (rtol 1e-1)


def test_llama_rope_with_scaling_differs():
  """llama_test.py:105-130: use_scale changes the result (positions where the scaled low frequencies matter)."""
  xb, _ = _inputs(3, seq=8)
  pos = np.arange(8) * 400 + 100
  a = _rotate(xb, pos, *O.rope_tables_llama31(128, 4096, use_scale=True))
  b = _rotate(xb, pos, *O.rope_tables_llama31(128, 4096, use_scale=False))
  assert not np.allclose(O.bf16_to_f32(a), O.bf16_to_f32(b), rtol=1e-3, atol=1e-2)
  ts_scaled = 1.0 / (1.0 / (10000.0 ** (np.arange(0, 128, 2) / 128)))
  # the three regimes of the rule: high frequencies untouched, low frequencies stretched by 8
  sn_s, _ = O.rope_tables_llama31(128, 2, use_scale=True)
  sn_u, _ = O.rope_tables_llama31(128, 2, use_scale=False)
  assert np.array_equal(sn_s[1, :20], sn_u[1, :20]) and np.allclose(np.arcsin(sn_s[1, -1]) * 8, np.arcsin(sn_u[1, -1]), rtol=1e-4)
  del ts_scaled


def test_rotation_keeps_the_norm_and_scaling_commutes():
  """llama_test.py:109-128 (test_scaling_rope) as a property: rotating after or before scaling by head_dim ** -0.5 agrees (rtol 1e-1)."""
  xb, pos = _inputs(4)
  sn, cs = O.rope_tables_llama31(128, 8, use_scale=True)
  x = O.bf16_to_f32(xb)
  a = O.bf16_to_f32(_rotate(xb, pos, sn, cs)) * 128 ** -0.5
  b = O.bf16_to_f32(_rotate(O.f32_to_bf16((x * 128 ** -0.5).astype(np.float32)).reshape(xb.shape), pos, sn, cs))
  assert np.allclose(a, b, rtol=1e-1, atol=2e-3)                        # (bf16 inputs on both sides here: the scaled copy rounds once more)
  assert np.allclose(np.linalg.norm(O.bf16_to_f32(_rotate(xb, pos, sn, cs)), axis=-1), np.linalg.norm(x, axis=-1), rtol=1e-2)
