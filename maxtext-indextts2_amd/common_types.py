"""Names shared with the reference (/root/reference/MaxText/common_types.py:62-74)."""
MODEL_MODE_AUTOREGRESSIVE = "autoregressive"
MODEL_MODE_PREFILL = "prefill"
MODEL_MODE_TRAIN = "train"
DECODING_ACTIVE_SEQUENCE_INDICATOR = 1
