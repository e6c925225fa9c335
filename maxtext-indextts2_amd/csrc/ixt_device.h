// ixt_device.h -- scalar numerics shared by every kernel ("IXT-MFMA-v2").
// Compiled with -ffp-contract=off: a*b+c never fuses; every fused multiply-add is an explicit fmaf.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;

#define IXT_MK 32      // k per matrix instruction (v_mfma_f32_16x16x32_bf16)
#define IXT_CHAIN 512  // k per accumulator chain of the canonical dot = the unit of a split-K slice
#define IXT_KALIGN 128 // feature dimensions must be multiples of this
#define IXT_SB 4     // logical pages per attention super-block

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// round-to-nearest-even; NaN stays NaN.  gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32: RNE, subnormals kept under this
// build's -fno-gpu-flush-denormals-to-zero); tests/test_gpu_ops.py::test_bf16_conversion_exhaustive compares it with the integer
// formula (u + 0x7fff + ((u >> 16) & 1)) >> 16 over all 2^32 inputs (NaN payloads aside).
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ bf16_t f2bf_int(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x0040u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }

// two bf16 packed in one dword: low half = element 0
__device__ __forceinline__ float bflo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bfhi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ uint32_t pack2(bf16_t lo, bf16_t hi) { return (uint32_t)lo | ((uint32_t)hi << 16); }

// exp from IEEE-only operations; argument clamped to [-80, 80] (no subnormals, no overflow).
__device__ __forceinline__ float ixt_exp(float x) {
  x = fminf(fmaxf(x, -80.0f), 80.0f);
  float n = __builtin_rintf(x * 1.44269504088896340736f);
  float r = __builtin_fmaf(n, -0.693145751953125f, x);
  r = __builtin_fmaf(n, -1.42860682030941723212e-6f, r);
  float p = 1.98412698412698412698e-4f;
  p = __builtin_fmaf(p, r, 1.38888888888888888889e-3f);
  p = __builtin_fmaf(p, r, 8.33333333333333333333e-3f);
  p = __builtin_fmaf(p, r, 4.16666666666666666667e-2f);
  p = __builtin_fmaf(p, r, 1.66666666666666666667e-1f);
  p = __builtin_fmaf(p, r, 0.5f);
  p = __builtin_fmaf(p, r, 1.0f);
  p = __builtin_fmaf(p, r, 1.0f);
  int e = (int)n;
  float s = __uint_as_float((uint32_t)(e + 127) << 23);
  return p * s;
}

// natural logarithm from IEEE-only operations (positive normal arguments): x = 2^e f, f in [sqrt(1/2), sqrt(2)); z = (f - 1) / (f + 1);
// log f = 2 z (1 + z^2/3 + ... + z^10/11) by Horner fma; result = fma(e, ln2_hi, fma(e, ln2_lo, log f)).
__device__ __forceinline__ float ixt_log(float x) {
  const uint32_t u = __float_as_uint(x);
  int e = (int)(u >> 23) - 127;
  float f = __uint_as_float((u & 0x007fffffu) | 0x3f800000u);
  if (f > 1.41421356237309504880f) { f = f * 0.5f; e = e + 1; }
  const float z = (f - 1.0f) / (f + 1.0f);
  const float w = z * z;
  float p = 9.09090909090909090909e-2f;
  p = __builtin_fmaf(p, w, 1.11111111111111111111e-1f);
  p = __builtin_fmaf(p, w, 1.42857142857142857143e-1f);
  p = __builtin_fmaf(p, w, 0.2f);
  p = __builtin_fmaf(p, w, 3.33333333333333333333e-1f);
  p = __builtin_fmaf(p, w, 1.0f);
  const float lf = (z + z) * p;
  const float fe = (float)e;
  return __builtin_fmaf(fe, 0.693145751953125f, __builtin_fmaf(fe, 1.42860682030941723212e-6f, lf));
}

// MlpBlock gate: bf16(bf16(a * bf16(sigmoid(a))) * b)
__device__ __forceinline__ bf16_t ixt_swiglu(float a, float b) {
  float sg = rbf(1.0f / (1.0f + ixt_exp(-a)));
  float u = rbf(a * sg);
  return f2bf(u * b);
}

// tanh from IEEE-only operations.  |x| < 1/16: odd series x (1 - x^2/3 + 2 x^4/15 - 17 x^6/315) by Horner fma (next term < 2e-13 relative);
// |x| >= 10: 1 (1 - tanh < 2^-25); else (1 - e) / (1 + e), e = ixt_exp(-2 |x|).  Sign copied from x.
__device__ __forceinline__ float ixt_tanh(float x) {
  const float a = fabsf(x);
  float t;
  if (a < 0.0625f) {
    const float w = a * a;
    float p = -5.39682539682539682540e-2f;
    p = __builtin_fmaf(p, w, 1.33333333333333333333e-1f);
    p = __builtin_fmaf(p, w, -3.33333333333333333333e-1f);
    p = __builtin_fmaf(p, w, 1.0f);
    t = a * p;
  } else if (a >= 10.0f) {
    t = 1.0f;
  } else {
    const float e = ixt_exp(-2.0f * a);
    t = (1.0f - e) / (1.0f + e);
  }
  return copysignf(t, x);
}

// Gemma MlpBlock gate (mlp_activations ["gelu", "linear"], linears.py:449-464; jax.nn.gelu approximate=True on bf16 values, every
// primitive rounded to bf16): cdf = 0.5 (1 + tanh(c (a + 0.044715 a^3))), bf16(bf16(a * cdf) * b)
__device__ __forceinline__ bf16_t ixt_geglu(float a, float b) {
  const float x3 = rbf(rbf(a * a) * a);
  const float u = rbf(a + rbf(0.044677734375f * x3));       // bf16(0.044715)
  const float th = rbf(ixt_tanh(rbf(0.796875f * u)));       // bf16(sqrt(2 / pi))
  const float cdf = rbf(0.5f * rbf(1.0f + th));
  const float g = rbf(a * cdf);
  return f2bf(g * b);
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// agent-scope (L1-bypassing) accesses for state one wave re-reads after writing it
__device__ __forceinline__ int ldg_i32(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stg_i32(int32_t* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
