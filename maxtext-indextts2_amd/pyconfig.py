"""Config shim with the reference's pyconfig surface (no JAX, no omegaconf).

Mirrors /root/reference/MaxText/pyconfig.py for the hot-path keys:
  * `initialize(argv, **kwargs)` with argv = [prog, yaml, "k=v", ...]          (pyconfig.py:1170)
  * yaml inheritance through `base_config:`                                     (:475-497)
  * every CLI / kwarg / env key must already exist in the yaml                  (:437-439)
  * env override `M_<KEY>`; CLI + env for one key is an error                   (:48-49, :414-421, :441-443)
  * values are typed like the yaml default                                      (:462-471)
  * `model_name` pulls configs/models/<name>.yml                                (:682-703)
  * derived keys (emb_dim, num_query_heads, ..., global_batch_size_to_load,
    pagedattn_max_pages_per_group)                                              (:551-644, :1044-1086)
  * the result is read-only                                                     (:1163-1164)
One process drives one GPU here, so "number of devices" in the batch-size derivation is 1.
"""
from __future__ import annotations

import math
import os
from typing import Any, Dict, List

import yaml

_MAX_PREFIX = "M_"
_BASE_CONFIG_ATTR = "base_config"
PKG_DIR = os.path.dirname(os.path.abspath(__file__))

VALID_ATTENTION = ("autoselected", "dot_product", "flash", "cudnn_flash_te", "cudnn_flash_jax", "paged")


def yaml_key_to_env_key(s: str) -> str:
  return _MAX_PREFIX + s.upper()


def _string_to_bool(s: str) -> bool:
  if s.lower() == "true":
    return True
  if s.lower() == "false":
    return False
  raise ValueError(f"Can't convert {s} to bool")


_yaml_types_to_parser = {str: str, int: int, float: float, bool: _string_to_bool}


def _parse_cli_value(text: str) -> Any:
  """omegaconf.from_cli types values with yaml rules; yaml.safe_load is the same grammar."""
  try:
    return yaml.safe_load(text) if text != "" else None
  except yaml.YAMLError:
    return text


def get_individual_scales(scale):
  """pyconfig.py:1044-1064."""
  log_2_scale = math.floor(math.log2(scale))
  if 2**log_2_scale != scale:
    raise ValueError("Global parameter scale should be a power of 2.")
  base_scale, rem = divmod(log_2_scale, 3)
  num_head_scale = base_scale + int(rem > 0)
  return base_scale + int(rem > 1), num_head_scale, num_head_scale, base_scale


def calculate_global_batch_sizes(per_device_batch_size, expansion_factor_real_data, num_devices, gradient_accumulation_steps):
  """pyconfig.py:1067-1086."""
  if per_device_batch_size < 1.0:
    micro_to_load = num_devices * expansion_factor_real_data if expansion_factor_real_data != -1 else num_devices
  else:
    if expansion_factor_real_data != -1:
      micro_to_load = int(num_devices * per_device_batch_size * expansion_factor_real_data)
    else:
      micro_to_load = int(num_devices * per_device_batch_size)
  micro_to_train = int(num_devices * per_device_batch_size)
  return int(micro_to_load * gradient_accumulation_steps), int(micro_to_train * gradient_accumulation_steps), micro_to_train


class _HyperParameters:

  def _load_config(self, config_name: str) -> Dict[str, Any]:
    with open(config_name, "r", encoding="utf-8") as f:
      raw = yaml.safe_load(f) or {}
    if _BASE_CONFIG_ATTR in raw:
      parent = raw[_BASE_CONFIG_ATTR]
      if not os.path.isabs(parent):
        cand = os.path.join(os.path.dirname(config_name), parent)
        if not os.path.isfile(cand):
          cand = os.path.join(PKG_DIR, "configs", parent)
        parent = cand
      base = self._load_config(parent)
      base.update(raw)
      return base
    return raw

  def __init__(self, argv: List[str], **kwargs):
    raw_yaml = self._load_config(argv[1])
    for env in os.environ:
      if env.startswith(_MAX_PREFIX):
        key = env[len(_MAX_PREFIX):].lower()
        if key not in raw_yaml:
          raise ValueError(f"We received env `{env}` but it doesn't match a key, so it is assumed a mistake.")
        if not env[len(_MAX_PREFIX):].isupper():
          raise ValueError(f"We received env `{env}` but it isn't all uppercase.")

    cmd: Dict[str, Any] = {}
    for item in argv[2:]:
      if "=" not in item:
        raise ValueError(f"CLI override `{item}` is not of the form key=value")
      k, v = item.split("=", 1)
      cmd[k] = _parse_cli_value(v)
    cmd.update(kwargs)

    raw_keys: Dict[str, Any] = {}
    for k in cmd:
      if k not in raw_yaml:
        raise ValueError(f"Key {k} was passed at the command line but isn't in config.")
    for k, default in raw_yaml.items():
      env_key = yaml_key_to_env_key(k)
      if k in cmd and env_key in os.environ:
        raise ValueError(f"You are passing overrides by both CLI and ENV for `{k}`. This isn't allowed.")
      if k not in cmd and env_key not in os.environ:
        raw_keys[k] = default
        continue
      proposal = cmd[k] if k in cmd else os.environ.get(env_key)
      if (not isinstance(proposal, type(default))) and (type(default) not in _yaml_types_to_parser):
        raise ValueError(f"For key '{k}', type {type(default)} not in {list(_yaml_types_to_parser)}, can't pass at the CLI or ENV")
      if proposal is None:
        raw_keys[k] = None
      elif isinstance(proposal, type(default)):
        raw_keys[k] = proposal  # take the raw data, no type conversion
      else:
        try:
          raw_keys[k] = _yaml_types_to_parser[type(default)](str(proposal) if isinstance(default, bool) else proposal)
        except ValueError as e:
          raise ValueError(f"Couldn't parse value from CLI or ENV '{proposal}' for key '{k}'") from e

    # model_name -> configs/models/<name>.yml, CLI/env wins over the model file (pyconfig.py:682-703)
    updated = set(cmd) | {k for k in raw_yaml if yaml_key_to_env_key(k) in os.environ}
    name = raw_keys["model_name"]
    if name != "default":
      path = os.path.join(PKG_DIR, "configs", "models", f"{name}.yml")
      if not os.path.isfile(path):
        raise ValueError(f"Invalid model name {name}: no {path}")
      with open(path, "r", encoding="utf-8") as f:
        model_vars = yaml.safe_load(f) or {}
      for k, v in model_vars.items():
        if k not in raw_yaml:
          raise ValueError(f"Key {k} in model config {name} isn't in config.")
        if k not in updated:
          raw_keys[k] = v

    self.keys = raw_keys
    self.user_init(raw_keys)

  @staticmethod
  def user_init(raw_keys):
    """pyconfig.py:551-644, restricted to the hot-path keys."""
    if raw_keys["run_name"] == "" or raw_keys["run_name"] is None:
      raw_keys["run_name"] = os.environ.get("JOBSET_NAME") or f'{raw_keys["model_name"]}_run'
    if raw_keys["attn_logits_soft_cap"] == 0.0:
      raw_keys["attn_logits_soft_cap"] = None
    if raw_keys["final_logits_soft_cap"] == 0.0:
      raw_keys["final_logits_soft_cap"] = None
    emb_scale, num_head_scale, mlp_dim_scale, layer_scale = get_individual_scales(raw_keys["global_parameter_scale"])
    raw_keys["emb_dim"] = 2**emb_scale * raw_keys["base_emb_dim"]
    raw_keys["num_query_heads"] = 2**num_head_scale * raw_keys["base_num_query_heads"]
    raw_keys["num_kv_heads"] = 2**num_head_scale * raw_keys["base_num_kv_heads"]
    raw_keys["mlp_dim"] = 2**mlp_dim_scale * raw_keys["base_mlp_dim"]
    raw_keys["num_decoder_layers"] = 2**layer_scale * raw_keys["base_num_decoder_layers"]
    requested_gbs = raw_keys["global_batch_size_to_load"]
    (raw_keys["global_batch_size_to_load"], raw_keys["global_batch_size_to_train_on"],
     raw_keys["micro_batch_size_to_train_on"]) = calculate_global_batch_sizes(
         raw_keys["per_device_batch_size"], raw_keys["expansion_factor_real_data"], 1, raw_keys["gradient_accumulation_steps"])
    if requested_gbs is not None and requested_gbs > 0:
      raw_keys["global_batch_size_to_load"] = requested_gbs
    if raw_keys["pagedattn_max_pages_per_group"] <= 0:
      raw_keys["pagedattn_max_pages_per_group"] = (
          raw_keys["max_target_length"] + raw_keys["pagedattn_tokens_per_page"] - 1) // raw_keys["pagedattn_tokens_per_page"]
    if raw_keys["attention"] not in VALID_ATTENTION:
      raise ValueError(f"Invalid attention {raw_keys['attention']}; valid: {VALID_ATTENTION}")
    if raw_keys["decoder_block"] not in ("llama2", "default"):
      raise ValueError(f"decoder_block {raw_keys['decoder_block']} is outside this build's hot path (llama2 only)")
    if isinstance(raw_keys["mlp_activations"], list):
      raw_keys["mlp_activations"] = tuple(raw_keys["mlp_activations"])


class HyperParameters:
  """Read-only view (pyconfig.py:1150-1167)."""

  def __init__(self, config):
    object.__setattr__(self, "_config", config)

  def __getattr__(self, attr):
    try:
      return object.__getattribute__(self, "_config").keys[attr]
    except KeyError as exc:
      raise AttributeError(f"'{self.__class__.__name__}' object has no attribute '{attr}'") from exc

  def __setattr__(self, attr, value):
    raise ValueError("Reinitialization of config is not allowed")

  def get_keys(self):
    return self._config.keys


def initialize(argv, **kwargs):
  return HyperParameters(_HyperParameters(list(argv), **kwargs))
