"""Vocabulary expansion for audio codes (reference: vocab_expansion/*, audio_token_mapping_adjusted.json).

  AudioTokenMapping   the reference's audio-id <-> embedding-row table (integer-exact)
  expand_embedder     append rows to `token_embedder/embedding` AND, when the head is untied, columns to
                      `decoder/logits_dense/kernel` of a flat {MaxText tree path: array} checkpoint (.safetensors / .npz)
"""
from .audio_mapping import AudioTokenMapping  # noqa: F401
from .expand_embedder import expand_embedder, expand_checkpoint_file  # noqa: F401
