"""MI355X-native KGE hot path behind the KGEkeras surface (models / utils of the reference).

    from kge_keras_amd.models import DistMult, TransE, ComplEx
    from kge_keras_amd.utils import generate_negative, oversample_data, validate, loss_function_lookup

All arithmetic runs in hand-written HIP kernels (csrc/, built in-tree into libkge_hip.so) reached
through the C ABI of include/kge_hip.h; PyTorch-ROCm only owns device memory and streams.  There is no
CPU fallback: importing works anywhere (the library also holds the host-side MT19937 samplers), but any
scoring / training / ranking call raises if no MI355X is present.
"""
from . import _lib  # noqa: F401  (fails loudly if libkge_hip.so is not built)
from . import models, utils, harness, sampler, dist  # noqa: F401

__all__ = ["models", "utils", "harness", "sampler", "dist"]
__version__ = "0.1.0"
