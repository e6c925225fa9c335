"""CPU restatement of IXT-FP8-v1 (test infrastructure, NOT product code): the arithmetic csrc/ixt_fp8.h defines for the projection kernels
with `quantization: fp8`, in numpy with float64 accumulation.

PARITY UNPINNED by construction: the reference has no fp8 inference path (layers/quantizations.py:289-298 installs AQT fp8 for training,
decode.py:127-128 refuses it), so there is nothing to pin to; this file states the scheme independently of the kernels and
tests/test_gpu_fp8.py bounds (1) kernels vs this restatement (accumulation-order noise only) and (2) the fp8 model vs the bf16 oracle.

e4m3 is the OCP format (MI355X_MICROARCH / cdna_hip_programming.md "FP8 -- OCP, not FNUZ"): 1 sign, 4 exponent (bias 7), 3 mantissa bits,
subnormals of 2^-9, largest finite 448, no infinities; conversions round to nearest even and saturate.
Only tests/ may import this module."""
from __future__ import annotations

import numpy as np

E4M3_MAX = 448.0
CHAIN = 512


def e4m3_round(x: np.ndarray) -> np.ndarray:
  """f32/f64 values -> the nearest e4m3 value (round half to even, saturating at +-448), as float64."""
  x = np.asarray(x, np.float64)
  s = np.sign(x)
  a = np.minimum(np.abs(x), E4M3_MAX)
  e = np.floor(np.log2(np.maximum(a, 2.0 ** -20)))
  e = np.maximum(e, -6.0)                      # below 2^-6 the spacing stays 2^-9 (subnormals)
  q = 2.0 ** (e - 3.0)                         # spacing of representable values in [2^e, 2^(e+1))
  r = np.round(a / q) * q                      # numpy rounds half to even
  return s * np.minimum(r, E4M3_MAX)


def bf16_to_f64(bits: np.ndarray) -> np.ndarray:
  return (np.asarray(bits, np.uint16).astype(np.uint32) << 16).view(np.float32).astype(np.float64)


def quant_weights(w: np.ndarray):
  """w f64 [K, N] (bf16 values) -> (q [K, N] e4m3 values, s [N] f32 scales): s_n = amax_k |w| / 448 in f32, q = e4m3(f32(w) / s)."""
  amax = np.abs(w).max(axis=0).astype(np.float32)
  s = np.where(amax > 0, amax / np.float32(448.0), np.float32(1.0)).astype(np.float32)
  q = e4m3_round((w.astype(np.float32) / s[None, :]).astype(np.float32))
  return q, s


def quant_rows(x: np.ndarray):
  """x f64 [R, K] (bf16 values) -> (q [R, K], a [R, chains] f32): one scale per row and 512-k block."""
  R, K = x.shape
  chains = (K + CHAIN - 1) // CHAIN
  q = np.zeros_like(x)
  a = np.ones((R, chains), np.float32)
  for c in range(chains):
    blk = x[:, c * CHAIN:(c + 1) * CHAIN].astype(np.float32)
    amax = np.abs(blk).max(axis=1)
    a[:, c] = np.where(amax > 0, amax / np.float32(448.0), np.float32(1.0))
    q[:, c * CHAIN:(c + 1) * CHAIN] = e4m3_round((blk / a[:, c:c + 1]).astype(np.float32))
  return q, a


def dot(x_bits: np.ndarray, w_bits: np.ndarray):
  """x bf16 bits [R, K], w bf16 bits [K, N] -> (t [R, N] float64 = sum_c (xq_c . wq_c) a_rc s_n, bound [R, N] = sum of |products| in the
  same units: the scale the kernels' f32 accumulation error is measured against)."""
  x, w = bf16_to_f64(x_bits), bf16_to_f64(w_bits)
  wq, s = quant_weights(w)
  xq, a = quant_rows(x)
  R, K = x.shape
  t = np.zeros((R, w.shape[1]))
  bound = np.zeros_like(t)
  for c in range((K + CHAIN - 1) // CHAIN):
    sl = slice(c * CHAIN, (c + 1) * CHAIN)
    scale = a[:, c:c + 1].astype(np.float64) * s[None, :].astype(np.float64)
    t += (xq[:, sl] @ wq[sl]) * scale
    bound += (np.abs(xq[:, sl]) @ np.abs(wq[sl])) * scale
  return t, bound
