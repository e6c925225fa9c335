"""Cases the REFERENCE's own tests hold for code on or next to the hot path, replayed against this build's restatements (CPU).

  tests/max_utils_test.py:34-37      l2norm_pytree({"a": [0,2,0], "b": [0,3,6]}) == 7
  tests/max_utils_test.py:65-83      T5X cross entropy with one-hot targets == integer-label softmax cross entropy (rtol 1e-5, atol 1e-8)
  tests/maxtext_utils_test.py:53-76  gradient clipping changes every gradient; fp8 statistics are neither clipped nor counted
  tests/pyconfig_test.py:24-39       model keys replace config keys; empty string stays empty
  tests/pyconfig_test.py:41-79       logical_axis_rules are merged rule by rule
  tests/pyconfig_test.py:81-128      two configs do not share state, a config is read-only; override_model_config lets the CLI win
"""
import numpy as np
import pytest
import torch

import helpers as H
from oracle import train_oracle as TO

pyconfig = H.pyconfig


# ------------------------------------------------------------------ max_utils_test.py
def test_l2norm_pytree():
  assert np.isclose(TO.l2norm_pytree({"a": np.array([0, 2, 0]), "b": np.array([0, 3, 6])}), 7, rtol=1e-5, atol=1e-8)


def test_t5x_cross_entropy_equals_integer_label_cross_entropy():
  """The reference draws targets in [1, 10) and uniform [0, 1) logits of shape (48, 2048, 4096); the same distributions at (8, 256, 4096)
  (the identity is per position).  Checked for the numpy restatement, for the expression the training oracle differentiates, and -- the VJP --
  against autograd of the integer-label form."""
  rng = np.random.default_rng(0)
  targets = rng.integers(1, 10, size=(8, 256))
  logits = rng.random((8, 256, 4096), dtype=np.float32)
  one_hot = np.eye(4096, dtype=np.float32)[targets]
  t5x, z = TO.cross_entropy_with_logits(logits, one_hot, 0.0)
  lt = torch.from_numpy(logits).requires_grad_(True)
  ref = torch.nn.functional.cross_entropy(lt.reshape(-1, 4096), torch.from_numpy(targets).reshape(-1), reduction="none").reshape(8, 256)
  assert np.allclose(t5x, ref.detach().numpy(), rtol=1e-5, atol=1e-8) and not z.any()
  # what oracle/train_oracle.py's loss() writes: logsumexp - gather
  lse = torch.logsumexp(lt, -1) - torch.gather(lt, -1, torch.from_numpy(targets)[..., None])[..., 0]
  assert np.allclose(lse.detach().numpy(), t5x, rtol=1e-5, atol=1e-8)
  # custom VJP (max_utils.py:624-651) == autograd of the plain form when z_loss = 0
  g = rng.random((8, 256), dtype=np.float32)
  ref.backward(torch.from_numpy(g))
  assert np.allclose(TO.cross_entropy_vjp(logits, one_hot, 0.0, g), lt.grad.numpy(), rtol=1e-4, atol=1e-7)


def test_z_loss_term_and_its_gradient():
  """max_utils.py:588-590, :646: loss += z * log_z^2; the VJP's factor (1 + 2 z log_z) is that term's derivative (finite differences)."""
  rng = np.random.default_rng(1)
  logits = rng.standard_normal((3, 5, 17)).astype(np.float64)
  one_hot = np.eye(17)[rng.integers(0, 17, size=(3, 5))]
  zc = 1e-2
  loss, zt = TO.cross_entropy_with_logits(logits, one_hot, zc)
  plain, _ = TO.cross_entropy_with_logits(logits, one_hot, 0.0)
  log_z = np.log(np.exp(logits).sum(-1))
  assert np.allclose(loss - plain, zc * log_z ** 2) and np.allclose(zt, zc * log_z ** 2)
  ana = TO.cross_entropy_vjp(logits, one_hot, zc, np.ones((3, 5)))
  eps = 1e-6
  for idx in [(0, 0, 0), (1, 2, 5), (2, 4, 16)]:
    lp, lm = logits.copy(), logits.copy()
    lp[idx] += eps
    lm[idx] -= eps
    num = (TO.cross_entropy_with_logits(lp, one_hot, zc)[0].sum() - TO.cross_entropy_with_logits(lm, one_hot, zc)[0].sum()) / (2 * eps)
    assert abs(num - ana[idx]) < 1e-6


# ------------------------------------------------------------------ maxtext_utils_test.py
def test_grad_clipping_with_no_fp8_stats():
  raw = {"params": np.array([3.0, -4.0]), "wi_0": np.array([5.0, -6.0])}
  clipped = TO.apply_gradient_clipping(raw, 1.0)
  for k, v in raw.items():
    assert not np.array_equal(v, clipped[k])
  assert np.isclose(TO.l2norm_pytree(clipped), 1.0, rtol=1e-6)          # optax.clip_by_global_norm: the clipped tree has norm == threshold


def test_fp8_stats_not_clipped_but_others_are():
  raw = {"params": {"wi_0": np.array([5.0, -6.0]), "wi_1": np.array([7.0, -8.0])}}
  expected = TO.apply_gradient_clipping(raw, 1.0)
  stats = {"amax_history_wi_0": np.array([3.0, -4.0]), "scale_wi_0": np.array([13.2, -4.4])}
  raw["_overwrite_with_gradient"] = stats
  clipped = TO.apply_gradient_clipping(raw, 1.0)
  for k in stats:
    assert np.array_equal(clipped["_overwrite_with_gradient"][k], stats[k])
  for k in ("wi_0", "wi_1"):
    assert np.array_equal(clipped["params"][k], expected["params"][k])


def test_small_gradients_are_left_alone():
  raw = {"w": np.array([0.3, -0.4], np.float32)}
  assert np.array_equal(TO.apply_gradient_clipping(raw, 1.0)["w"], raw["w"])


# ------------------------------------------------------------------ pyconfig_test.py
def test_basic_override():
  raw = {"megablox": None, "foo": ["bar", "baz"]}
  pyconfig.validate_and_update_keys(raw, {"foo": ["x", "y"]}, config_name="config")
  assert raw == {"megablox": None, "foo": ["x", "y"]}


def test_model_key_must_exist_with_the_same_type():
  with pytest.raises(ValueError):
    pyconfig.validate_and_update_keys({"foo": 1}, {"bar": 2}, config_name="config")
  with pytest.raises(ValueError):
    pyconfig.validate_and_update_keys({"foo": 1}, {"foo": "text"}, config_name="config")


def test_empty_string_parse_as_empty_string():
  config = pyconfig.initialize(["train.py", H.BASE_YML], quantization="")
  assert config.quantization is None or config.quantization == ""


def test_logical_axis_override():
  raw = {"megablox": None, "foo": ["bar", "baz"], "logical_axis_rules": [["activation", ["data", "fsdp"]], ["norm", "tensor"]]}
  pyconfig.validate_and_update_keys(raw, {"logical_axis_rules": [["activation", ["data", "fsdp_transpose"]], ["norm", "fsdp"]]}, config_name="config")
  assert raw == {"megablox": None, "foo": ["bar", "baz"], "logical_axis_rules": [("activation", ["data", "fsdp_transpose"]), ("norm", "fsdp")]}


def test_logical_axis_partial_override():
  raw = {"megablox": None, "foo": ["bar", "baz"], "logical_axis_rules": [["activation", ["data", "fsdp"]], ["norm", "tensor"]]}
  pyconfig.validate_and_update_keys(raw, {"logical_axis_rules": [["norm", "fsdp"]]}, config_name="config")
  assert raw == {"megablox": None, "foo": ["bar", "baz"], "logical_axis_rules": [("activation", ("data", "fsdp")), ("norm", "fsdp")]}


def test_multiple_unmodifiable_configs():
  """pyconfig_test.py:81-114 with keys this build's base.yml carries (the parallelism keys of the original are outside the path)."""
  common = dict(per_device_batch_size=1.0, run_name="test", enable_checkpointing=False, base_num_decoder_layers=2, attention="paged",
                max_target_length=16, base_emb_dim=256, base_num_query_heads=2, base_num_kv_heads=2, max_prefill_predict_length=4)
  config_train = pyconfig.initialize(["train.py", H.BASE_YML], pagedattn_num_pages=8, **common)
  config_inference = pyconfig.initialize(["decode.py", H.BASE_YML], pagedattn_num_pages=16, **common)
  assert config_train.pagedattn_num_pages != config_inference.pagedattn_num_pages
  with pytest.raises(ValueError):
    config_inference.pagedattn_num_pages = 4


def test_overriding_model():
  """pyconfig_test.py:116-128: with override_model_config the CLI value wins and the model file fills the rest; without it, a key given in
  both places is an error (pyconfig.py:368-374, :510-511)."""
  config = pyconfig.initialize(["train.py", H.BASE_YML], model_name="gemma2-2b", override_model_config=True, base_emb_dim=1024)   # 2304 in gemma2-2b
  assert config.base_emb_dim == 1024
  assert config.base_mlp_dim == 9216
  with pytest.raises(ValueError, match="overwritten from both the model"):
    pyconfig.initialize(["train.py", H.BASE_YML], model_name="gemma2-2b", base_emb_dim=1024)


def test_model_files_next_to_the_base_config_win(tmp_path):
  """pyconfig.py:686-692: configs/models/<name>.yml is looked up beside the yaml that was passed first, then in the package."""
  base = tmp_path / "base.yml"
  base.write_text(f"base_config: {H.BASE_YML}\n")
  (tmp_path / "models").mkdir()
  (tmp_path / "models" / "ixt-tiny.yml").write_text("base_emb_dim: 384\n")
  assert pyconfig.initialize(["decode.py", str(base)], model_name="ixt-tiny").emb_dim == 384
  assert pyconfig.initialize(["decode.py", H.BASE_YML], model_name="ixt-tiny").emb_dim == 256
