"""The reference's layer surface (MaxText/layers/*), by name, over the op-level C ABI of libixt.so.

Each class keeps the reference's constructor / call argument names for the arguments this path uses; arguments that only steer XLA
sharding, quantisation or initialisation are accepted and ignored.  Parameters are torch CUDA tensors (bf16); every arithmetic step is an
`ixt_op_*` call, so a model assembled from these modules produces the same bits as the engine's fused kernels (tests/test_gpu_layers.py)."""
from . import attentions, decoders, embeddings, linears, llama2, models, normalizations  # noqa: F401
