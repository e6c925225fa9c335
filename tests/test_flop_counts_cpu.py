"""The work counts the measurements are priced in (maxtext-indextts2_amd/maxtext_utils.py) against the golden cases the reference's own tests
hold (tests/flop_calculation_test.py:99-208: 6 x tokens x published parameter count + causal attention, within 5 %), and against the figures
SURVEY.md section 8(d) derives for this build's training and prefill rows."""
import importlib
from types import SimpleNamespace

import helpers as H  # noqa: F401  (sys.path)

mu = importlib.import_module("maxtext-indextts2_amd.maxtext_utils")


def _cfg(**kw):
  base = dict(decoder_block="llama2", num_experts=1, num_experts_per_tok=1, mlp_activations=("silu", "linear"), gradient_accumulation_steps=1,
              sliding_window_size=0)
  base.update(kw)
  return SimpleNamespace(**base)


def _golden(B, S, layers, heads, head_dim, params):
  attention = 2 * 3 * layers * B * S ** 2 * heads * head_dim / 1e12          # flop_calculation_test.py:46-73
  return 6 * B * S * params / 1e12 + attention


def _close(a, b, rel=5e-2):
  return abs(a - b) / max(abs(a), abs(b)) <= rel


def test_llama2_7b_reference_case():
  c = _cfg(per_device_batch_size=12, max_target_length=2048, emb_dim=4096, mlp_dim=11008, num_query_heads=32, num_kv_heads=32, num_decoder_layers=32,
           head_dim=128, vocab_size=32000)
  total, learnable, attention = mu.calculate_tflops_training_per_device(c, log=False)
  assert _close(total, _golden(12, 2048, 32, 32, 128, 6.74e9)) and abs(total - learnable - attention) < 1e-9


def test_llama3_8b_reference_case():
  c = _cfg(per_device_batch_size=4, max_target_length=2048, emb_dim=4096, mlp_dim=14336, num_query_heads=32, num_kv_heads=8, num_decoder_layers=32,
           head_dim=128, vocab_size=128256)
  assert _close(mu.calculate_tflops_training_per_device(c, log=False)[0], _golden(4, 2048, 32, 32, 128, 7.50e9))


def test_mixtral_8x7b_reference_case():
  c = _cfg(decoder_block="mixtral", num_experts=8, num_experts_per_tok=2, per_device_batch_size=4, max_target_length=8192, emb_dim=4096, mlp_dim=14336,
           num_query_heads=32, num_kv_heads=8, num_decoder_layers=32, head_dim=128, vocab_size=32000)
  assert _close(mu.calculate_tflops_training_per_device(c, log=False)[0], _golden(4, 8192, 32, 32, 128, 12.9e9))


def test_c3_training_row_of_the_survey():
  """C2 dims, 8 x 2048 tokens per GPU -> learnable-weight 148.8 TFLOP + attention 6.6 TFLOP = 155.4 TFLOP per step."""
  c = _cfg(per_device_batch_size=8, max_target_length=2048, emb_dim=2048, mlp_dim=7168, num_query_heads=16, num_kv_heads=16, num_decoder_layers=16,
           head_dim=128, vocab_size=264192)
  total, learnable, attention = mu.calculate_tflops_training_per_device(c, log=False)
  assert abs(learnable - 148.8) < 0.1 and abs(attention - 6.6) < 0.05 and abs(total - 155.4) < 0.1


def test_gemma_attention_variants_and_prefill_count():
  g2 = _cfg(decoder_block="gemma2", sliding_window_size=4096, mlp_activations=("gelu", "linear"), per_device_batch_size=1, max_target_length=8192, emb_dim=2304,
            mlp_dim=9216, num_query_heads=8, num_kv_heads=4, num_decoder_layers=13, head_dim=256, vocab_size=256128)
  _, learnable, attention = mu.calculate_tflops_training_per_device(g2, log=False)
  full = 4 * 8192 ** 2 * 8 * 256
  assert abs(attention - (full + full / 2) * 13 / 2 * 3 / 1e12) < 1e-9          # a global and a half-length local sub-layer per layer
  g2.sliding_window_size = 8192
  assert abs(mu.calculate_tflops_training_per_device(g2, log=False)[2] - 2 * full * 13 / 2 * 3 / 1e12) < 1e-9
  g3 = _cfg(decoder_block="gemma3", sliding_window_size=1024, mlp_activations=("gelu", "linear"), per_device_batch_size=1, max_target_length=8192, emb_dim=2560,
            mlp_dim=10240, num_query_heads=8, num_kv_heads=4, num_decoder_layers=34, head_dim=256, vocab_size=262144)
  att3 = mu.calculate_tflops_training_per_device(g3, log=False)[2]
  assert abs(att3 - (5 * full + 29 * full / 8) / 2 * 3 / 1e12) < 1e-9           # 34 // 6 = 5 global layers, 29 with a 1024-key window
  # prefill: 2 x 256 x 973.2 M + causal attention 4.3e9 at C2
  c2 = _cfg(num_query_heads=16, num_decoder_layers=16, head_dim=128)
  total, lw, att = mu.calculate_prefill_tflops_per_device(973.2e6, 256, c2, log=False)
  assert abs(lw - 2 * 256 * 973.2e6 / 1e12) < 1e-12 and abs(att - 4.3e9 / 1e12) < 1e-4 and abs(total - lw - att) < 1e-15
