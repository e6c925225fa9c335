"""CLI harness with the call order of MaxText/decode.py:85-184 over the MI355X engine:

  python -c "import sys, importlib; importlib.import_module('maxtext-indextts2_amd.decode').main(sys.argv)" \\
      maxtext-indextts2_amd/configs/base.yml model_name=indextts2-1b per_device_batch_size=1 \\
      max_prefill_predict_length=256 max_target_length=1280 attention=paged scan_layers=false

pyconfig.initialize(argv) -> MaxEngine -> load_params -> prefill (per stream) -> init_decode_state -> insert -> generate loop over
range(max_prefill_predict_length, max_target_length) -> tokens of each stream.  The engine consumes and produces token ids
(the reference tokenizes `prompt` with a sentencepiece / HF tokenizer that is not part of this path): the prompt ids come from
`ixt_prompt_tokens` (comma-separated ids) or, when that is empty, from the seeded synthetic prompt the tests and the bench use
(text-like ids followed by audio-code rows).  `ixt_audio_only_sampling=true` restricts greedy sampling to the audio rows of
`audio_token_mapping_path` and prints the audio codes next to the rows.
"""
from __future__ import annotations

import importlib
import sys
from typing import Sequence

import numpy as np

_NUM_STREAMS = 1   # decode.py:41


def _batch_first_result_token(first_tokens, batch_size: int):
  """decode.py:44-82: the per-stream first tokens as one batch-sized ResultTokens (streams beyond _NUM_STREAMS are invalid)."""
  engine_api = importlib.import_module(__package__ + ".engine_api")
  import torch  # pylint: disable=import-outside-toplevel
  data = torch.zeros((batch_size, 3), dtype=torch.int32, device=first_tokens[0].data.device)
  for i, ft in enumerate(first_tokens):
    data[i] = ft.data[0]
  return engine_api.ResultTokens(data=data, tokens_idx=(0, 1), valid_idx=(1, 2), length_idx=(2, 3), samples_per_slot=1)


def main(argv: Sequence[str]) -> None:
  pyconfig = importlib.import_module(__package__ + ".pyconfig")
  maxengine = importlib.import_module(__package__ + ".maxengine")
  synth = importlib.import_module(__package__ + ".synth")
  config = pyconfig.initialize(argv)
  assert config.load_full_state_path == "", "Decode doesn't operate on full states! Convert to parameter checkpoint first."   # decode.py:187-190
  engine = maxengine.MaxEngine(config)
  params = engine.load_params()
  mapping = None
  if config.ixt_audio_only_sampling:
    ve = importlib.import_module(__package__ + ".vocab_expansion")
    path = config.audio_token_mapping_path
    mapping = ve.AudioTokenMapping.from_npz(path) if path.endswith(".npz") else ve.AudioTokenMapping.from_json(path)
    engine.set_audio_only_sampling(mapping)

  prefill_length = config.max_prefill_predict_length
  if config.ixt_prompt_tokens:
    ids = [int(t) for t in str(config.ixt_prompt_tokens).replace(" ", "").split(",") if t]
    true_length = len(ids)
    assert true_length <= prefill_length, f"Input token length {true_length} is longer than {config.max_prefill_predict_length=}"   # decode.py:124-126
    tokens = np.zeros((prefill_length,), np.int32)
    tokens[:true_length] = ids
  else:
    tokens, true_length = synth.synth_prompt(config.vocab_size, prefill_length, 1234)
  batch_size = int(config.per_device_batch_size)
  assert 0 < _NUM_STREAMS <= batch_size, f"The number of streams {_NUM_STREAMS} must be > 0 and <= batch size {batch_size}"

  prefill_result_list, first_token_list, sampled_tokens_list = [], [], []
  for i in range(_NUM_STREAMS):   # Prefill, decode.py:139-151
    prefill_result, first_token = engine.prefill(params=params, padded_tokens=tokens, true_length=true_length, slot=i)
    prefill_result_list.append(prefill_result)
    first_token_list.append(first_token)
  decode_state = engine.init_decode_state()   # Insert, decode.py:153-157
  for i in range(_NUM_STREAMS):
    decode_state = engine.insert(prefill_result_list[i], decode_state, slot=i)
  steps = range(config.max_prefill_predict_length, config.max_target_length)   # Generate, decode.py:160-173
  sampled_tokens_list.append(_batch_first_result_token(first_token_list, batch_size))
  for _ in steps:
    decode_state, sampled_tokens = engine.generate(params, decode_state)
    sampled_tokens_list.append(sampled_tokens)

  for i in range(_NUM_STREAMS):   # Get results, decode.py:175-179
    results = [int(t.get_result_at_slot(i).tokens.item()) for t in sampled_tokens_list]
    print(f"Input ids ({true_length}) -> {len(results)} generated ids: {results}")
    if mapping is not None:
      print(f"audio codes: {mapping.to_audio(np.array(results)).tolist()}")
  if config.autoregressive_decode_assert:
    want = [int(t) for t in str(config.autoregressive_decode_assert).replace(" ", "").split(",") if t]
    assert results[:len(want)] == want, f"generated ids mismatch {results[:len(want)]=}, {config.autoregressive_decode_assert=}"   # decode.py:181-183


if __name__ == "__main__":
  main(sys.argv)
