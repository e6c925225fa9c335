"""Config shim with the reference's pyconfig surface (no JAX, no omegaconf).

Behaviour mirrored from /root/reference/MaxText/pyconfig.py for the hot-path keys (the code is this build's own):
  * `initialize(argv, **kwargs)` with argv = [prog, yaml, "k=v", ...]                     (pyconfig.py:1170)
  * yaml inheritance through `base_config:`                                              (:475-497)
  * every CLI / kwarg / env key must already exist in the yaml                           (:437-439)
  * env override `M_<KEY>`; CLI + env for one key is an error                            (:48-49, :414-421, :441-443)
  * a value given as text is parsed to the type of the yaml default                      (:462-471)
  * `model_name` pulls configs/models/<name>.yml; a key set both there and at the CLI /
    environment is an error unless `override_model_config` is true (then the CLI wins)   (:368-374, :510-511, :682-703)
  * `validate_and_update_keys`: a model key must exist with the same type; the
    `logical_axis_rules` list is merged rule by rule                                     (:990-1041)
  * derived keys (emb_dim, num_query_heads, ..., global_batch_size_to_load,
    pagedattn_max_pages_per_group)                                                       (:551-644, :1044-1086)
  * the result is read-only                                                              (:1163-1164)
One process drives one GPU here, so "number of devices" in the batch-size derivation is 1.
"""
from __future__ import annotations

import math
import os
import types
from typing import Any, Dict, List, Mapping, Tuple

import yaml

_MAX_PREFIX = "M_"
PKG_DIR = os.path.dirname(os.path.abspath(__file__))

VALID_ATTENTION = ("autoselected", "dot_product", "flash", "cudnn_flash_te", "cudnn_flash_jax", "paged")


def yaml_key_to_env_key(s: str) -> str:
  return _MAX_PREFIX + s.upper()


# ------------------------------------------------------------------------------------------------ sources
def _read_yaml_chain(path: str) -> Dict[str, Any]:
  """A yaml file on top of the file its `base_config:` names (looked up next to it, then in this package's configs/)."""
  with open(path, "r", encoding="utf-8") as f:
    here = yaml.safe_load(f) or {}
  parent = here.pop("base_config", None)
  if parent is None:
    return here
  if not os.path.isabs(parent):
    beside = os.path.join(os.path.dirname(path), parent)
    parent = beside if os.path.isfile(beside) else os.path.join(PKG_DIR, "configs", parent)
  merged = _read_yaml_chain(parent)
  merged.update(here)
  return merged


def _collect_overrides(argv_tail: List[str], kwargs: Mapping[str, Any], known: Mapping[str, Any]) -> Dict[str, Tuple[Any, str]]:
  """{key: (proposal, where it came from)} from "k=v" arguments, keyword arguments and M_<KEY> environment variables."""
  found: Dict[str, Tuple[Any, str]] = {}
  for item in argv_tail:
    key, sep, text = item.partition("=")
    if not sep:
      raise ValueError(f"CLI override `{item}` is not of the form key=value")
    try:
      value = yaml.safe_load(text) if text != "" else None      # the grammar omegaconf.from_cli applies
    except yaml.YAMLError:
      value = text
    found[key] = (value, "cli")
  for key, value in kwargs.items():
    found[key] = (value, "cli")
  for key in found:
    if key not in known:
      raise ValueError(f"Key {key} was passed at the command line but isn't in config.")
  for name, text in os.environ.items():
    if not name.startswith(_MAX_PREFIX):
      continue
    bare = name[len(_MAX_PREFIX):]
    key = bare.lower()
    if key not in known:
      raise ValueError(f"We received env `{name}` but it doesn't match a key, so it is assumed a mistake.")
    if not bare.isupper():
      raise ValueError(f"We received env `{name}` but it isn't all uppercase.")
    if key in found:
      raise ValueError(f"You are passing overrides by both CLI and ENV for `{key}`. This isn't allowed.")
    found[key] = (text, "env")
  return found


def _as_type_of(default: Any, proposal: Any, key: str) -> Any:
  """The proposal, typed like the yaml default: taken as is when it already has that type, parsed when it arrived as something else."""
  if proposal is None or isinstance(proposal, type(default)):
    return proposal
  if isinstance(default, bool):
    text = str(proposal).lower()
    if text in ("true", "false"):
      return text == "true"
    raise ValueError(f"Couldn't parse value from CLI or ENV '{proposal}' for key '{key}'")
  for kind in (int, float, str):
    if isinstance(default, kind):
      try:
        return kind(proposal)
      except ValueError as e:
        raise ValueError(f"Couldn't parse value from CLI or ENV '{proposal}' for key '{key}'") from e
  raise ValueError(f"For key '{key}', type {type(default)} not in [str, int, float, bool], can't pass at the CLI or ENV")


# ------------------------------------------------------------------------------------------------ model keys
def _freeze(x):
  return tuple(_freeze(v) for v in x) if isinstance(x, list) else x


def create_new_logical_axis_rules(old_logical_axis_rules, new_logical_axis_rules):
  """pyconfig.py:1001-1016: rules of the model file replace the rules of the same logical axis (moved to the end, mesh axes as given);
  untouched rules keep their place with their mesh axes frozen to tuples; a rule for an axis the base does not know is dropped."""
  known = {rule[0] for rule in old_logical_axis_rules}
  taken = [(axis, mesh) for axis, mesh in new_logical_axis_rules if axis in known]
  replaced = {axis for axis, _ in taken}
  return [(axis, _freeze(mesh)) for axis, mesh in old_logical_axis_rules if axis not in replaced] + taken


def validate_and_update_keys(raw_keys, model_keys, config_name: str):
  """pyconfig.py:1019-1041: every key of a model file must exist in the config with a value of the same type; then it replaces it."""
  for key, value in model_keys.items():
    if key not in raw_keys:
      raise ValueError(f"Key {key} does not exist in config {config_name}.")
    if not isinstance(raw_keys[key], type(value)):
      raise ValueError(f"Type of key:{key} does not match with {type(value)}")
    raw_keys[key] = create_new_logical_axis_rules(raw_keys[key], value) if key == "logical_axis_rules" else value
  return raw_keys


# ------------------------------------------------------------------------------------------------ derived keys
def get_individual_scales(scale):
  """pyconfig.py:1044-1064: a power-of-two parameter scale spread over (emb, heads, mlp, layers), three doublings per round:
  heads and mlp first, then emb, then layers."""
  exponent = math.floor(math.log2(scale))
  if 2**exponent != scale:
    raise ValueError("Global parameter scale should be a power of 2.")
  rounds, extra = divmod(exponent, 3)
  heads_and_mlp = rounds + (1 if extra >= 1 else 0)
  return rounds + (1 if extra >= 2 else 0), heads_and_mlp, heads_and_mlp, rounds


def calculate_global_batch_sizes(per_device_batch_size, expansion_factor_real_data, num_devices, gradient_accumulation_steps):
  """pyconfig.py:1067-1086 -> (global batch to load, global batch to train on, micro batch to train on)."""
  expand = 1 if expansion_factor_real_data == -1 else expansion_factor_real_data
  per_device_to_load = 1 if per_device_batch_size < 1.0 else per_device_batch_size
  micro_to_load = int(num_devices * per_device_to_load * expand)
  micro_to_train = int(num_devices * per_device_batch_size)
  return micro_to_load * int(gradient_accumulation_steps), micro_to_train * int(gradient_accumulation_steps), micro_to_train


def _derive(keys: Dict[str, Any]) -> None:
  """pyconfig.py:551-644, restricted to the hot-path keys."""
  if not keys["run_name"]:
    keys["run_name"] = os.environ.get("JOBSET_NAME") or f'{keys["model_name"]}_run'
  for cap in ("attn_logits_soft_cap", "final_logits_soft_cap"):
    if keys[cap] == 0.0:
      keys[cap] = None
  emb, heads, mlp, layers = get_individual_scales(keys["global_parameter_scale"])
  keys["emb_dim"] = keys["base_emb_dim"] << emb
  keys["num_query_heads"] = keys["base_num_query_heads"] << heads
  keys["num_kv_heads"] = keys["base_num_kv_heads"] << heads
  keys["mlp_dim"] = keys["base_mlp_dim"] << mlp
  keys["num_decoder_layers"] = keys["base_num_decoder_layers"] << layers
  asked = keys["global_batch_size_to_load"]
  to_load, to_train, micro = calculate_global_batch_sizes(keys["per_device_batch_size"], keys["expansion_factor_real_data"], 1,
                                                          keys["gradient_accumulation_steps"])
  keys["global_batch_size_to_load"] = asked if (asked is not None and asked > 0) else to_load
  keys["global_batch_size_to_train_on"], keys["micro_batch_size_to_train_on"] = to_train, micro
  if keys["pagedattn_max_pages_per_group"] <= 0:
    keys["pagedattn_max_pages_per_group"] = -(-keys["max_target_length"] // keys["pagedattn_tokens_per_page"])
  if keys["attention"] not in VALID_ATTENTION:
    raise ValueError(f"Invalid attention {keys['attention']}; valid: {VALID_ATTENTION}")
  if keys["decoder_block"] not in ("llama2", "default", "gemma2", "gemma3", "mixtral"):
    raise ValueError(f"decoder_block {keys['decoder_block']} is outside this build's hot path (llama2, gemma2, gemma3 and mixtral are built)")
  if keys["decoder_block"] == "mixtral":
    if not 1 <= keys["num_experts_per_tok"] <= min(keys["num_experts"], 8) or keys["num_experts"] > 32:
      raise ValueError("mixtral: 1 <= num_experts_per_tok <= min(num_experts, 8) and num_experts <= 32")
  elif keys["num_experts"] != 1 or keys["num_experts_per_tok"] != 1:
    raise ValueError("num_experts > 1 is built for decoder_block mixtral only")
  if keys["decoder_block"] not in ("gemma2", "gemma3") and (keys["attn_logits_soft_cap"] or keys["final_logits_soft_cap"] or keys["sliding_window_size"] or
                                            keys["use_post_attn_norm"] or keys["use_post_ffw_norm"] or keys["logits_via_embedding"]):
    raise ValueError("soft caps, sliding_window_size, post norms and logits_via_embedding are built for decoder_block gemma2 / gemma3 only")
  if isinstance(keys["mlp_activations"], list):
    keys["mlp_activations"] = tuple(keys["mlp_activations"])


def _resolve(argv: List[str], kwargs: Mapping[str, Any]) -> Dict[str, Any]:
  defaults = _read_yaml_chain(argv[1])
  overrides = _collect_overrides(list(argv[2:]), kwargs, defaults)
  keys = {k: (_as_type_of(d, overrides[k][0], k) if k in overrides else d) for k, d in defaults.items()}
  name = keys["model_name"]
  if name != "default":                                     # pyconfig.py:682-703
    path = os.path.join(os.path.dirname(os.path.abspath(argv[1])), "models", f"{name}.yml")
    if not os.path.isfile(path):
      path = os.path.join(PKG_DIR, "configs", "models", f"{name}.yml")
    if not os.path.isfile(path):
      raise ValueError(f"Invalid model name {name}: no {path}")
    with open(path, "r", encoding="utf-8") as f:
      model_keys = yaml.safe_load(f) or {}
    if keys["override_model_config"]:                       # debugging / testing: what was given explicitly wins over the model file
      model_keys = {k: v for k, v in model_keys.items() if k not in overrides}
    else:
      twice = [k for k in overrides if k in model_keys]
      if twice:
        raise ValueError(f"Keys {twice} are overwritten from both the model and the environment/command line. This isn't allowed.")
    validate_and_update_keys(keys, model_keys, config_name=argv[1])
  _derive(keys)
  return keys


# ------------------------------------------------------------------------------------------------ the read-only result
class HyperParameters:
  """Attribute view of the resolved keys; assignment raises (pyconfig.py:1163-1164)."""

  __slots__ = ("_keys",)

  def __init__(self, keys: Mapping[str, Any]):
    object.__setattr__(self, "_keys", types.MappingProxyType(dict(keys)))

  def __getattr__(self, name):
    keys = object.__getattribute__(self, "_keys")
    if name in keys:
      return keys[name]
    raise AttributeError(f"'{type(self).__name__}' object has no attribute '{name}'")

  def __setattr__(self, name, value):
    raise ValueError("Reinitialization of config is not allowed")

  def get_keys(self):
    return dict(object.__getattribute__(self, "_keys"))


def initialize(argv, **kwargs):
  return HyperParameters(_resolve(list(argv), kwargs))
