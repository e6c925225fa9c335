"""The work counts of MaxText/maxtext_utils.py that the measurements here are priced in: training TFLOP per step and
device (`calculate_tflops_training_per_device`, maxtext_utils.py:475-587 with the FFN helper :297-309 and the Gemma-2 / Gemma-3 attention
variants :156-224) and prefill TFLOP (`calculate_prefill_tflops_per_device`, :590-614).  Same names, arguments and return tuples; pinned by the
reference's own golden cases (tests/flop_calculation_test.py:99-208 -> tests/test_flop_counts_cpu.py)."""
from __future__ import annotations

from typing import Any, Tuple


def _tokens(config: Any) -> float:
  return float(config.per_device_batch_size) * config.max_target_length


def calculate_ffn_mamtul_tflops_per_device(config: Any, mlp_dim: int) -> float:
  """maxtext_utils.py:297-309 (FLOPs, despite the name): the up projections (one per activation) and the down projection of one layer."""
  up = 2.0 * _tokens(config) * mlp_dim * config.emb_dim * len(config.mlp_activations)
  down = 2.0 * _tokens(config) * mlp_dim * config.emb_dim
  return up + down


def _attention_tflops(config: Any) -> float:
  """Causal attention of all layers, forward + backward (x 3), in TFLOP: full attention per layer, except that Gemma-2 pairs a global with a
  local (windowed) sub-layer per layer (:156-181) and Gemma-3 has one global layer in six (:184-224)."""
  B, S, H, D, L = float(config.per_device_batch_size), config.max_target_length, config.num_query_heads, config.head_dim, config.num_decoder_layers
  full = 4.0 * B * S * S * H * D
  block = config.decoder_block
  if block in ("gemma2", "gemma3"):
    local = 4.0 * B * S * min(config.sliding_window_size, S) * H * D
    if block == "gemma2":
      noncausal = (full + local) * L
    else:
      n_global = L // 6
      noncausal = n_global * full + (L - n_global) * local
  else:
    noncausal = full * L
  return noncausal / 2.0 * 3.0 / 1e12


def calculate_tflops_training_per_device(config: Any, log: bool = True) -> Tuple[float, float, float]:
  """maxtext_utils.py:475-587 for the blocks built here (llama2 / default, mixtral, gemma2, gemma3): (total, learnable-weight, attention)
  TFLOP per step and device; x 3 for forward + backward is inside."""
  T = _tokens(config)
  if config.num_experts > 1:                      # dropless routed mixture: the gate plus num_experts_per_tok expert MLPs per token
    ffn = 2.0 * T * config.emb_dim * config.num_experts + calculate_ffn_mamtul_tflops_per_device(config, config.mlp_dim) * config.num_experts_per_tok
  else:
    ffn = calculate_ffn_mamtul_tflops_per_device(config, config.mlp_dim)
  qkv = 2.0 * T * config.emb_dim * (config.num_query_heads + 2 * config.num_kv_heads) * config.head_dim
  proj = 2.0 * T * config.emb_dim * config.num_query_heads * config.head_dim
  embedding = 2.0 * T * config.emb_dim * config.vocab_size
  sublayers = config.num_decoder_layers * (2 if config.decoder_block == "gemma2" else 1)       # :176-179
  accum = getattr(config, "gradient_accumulation_steps", 1)
  learnable = ((ffn + qkv + proj) * sublayers + embedding) * 3.0 / 1e12 * accum
  attention = _attention_tflops(config) * accum
  total = learnable + attention
  if log:
    print(f"Per train step:\n Total TFLOPs: {total:.2f} \n split as {100 * learnable / total:.2f}% learnable weight flops and "
          f"{100 * attention / total:.2f}% attention flops")
  return total, learnable, attention


def calculate_prefill_tflops_per_device(num_model_parameters: float, prefill_length: int, config: Any, log: bool = True, device_count: int = 1):
  """maxtext_utils.py:590-614 (`jax.device_count()` is the `device_count` argument here: one process per GPU)."""
  learnable = 2.0 * num_model_parameters * prefill_length / device_count / 1e12
  causal = 4.0 * config.num_query_heads * config.num_decoder_layers * config.head_dim * prefill_length ** 2 / device_count / 1e12 / 2.0
  total = learnable + causal
  if log:
    print(f"Per prefill step per device: \n\tTotal TFLOPs: {total:.2f} \n\t\tLearnable weight TFLOPs: {learnable:.2f} \n\t\tCausal attention TFLOPs: {causal:.2f}")
  return total, learnable, causal
