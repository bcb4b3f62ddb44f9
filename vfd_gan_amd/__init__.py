"""vfd_gan_amd — MI355X-native video-GAN training step behind the reference's own Python surface.

Layout (mirrors umaionigiri/vfd_gan's module names for the hot path only):
  csrc/                HIP kernels + C ABI (include/vfdgan_hip.h) -> libvfdgan_hip.so
  _lib.py              ctypes binding (fails loudly when the library is missing; no CPU fallback)
  functional.py, nn.py autograd glue and torch.nn-compatible layers over the C ABI
  optim.py             flat-arena Adam (one launch per optimiser step)
  dist.py              RCCL gradient reducer (one process per GPU)
  lib/, models/, trainer.py   host-side mirror of the reference's lib/, models/, trainer.py
"""
from .functional import (ClTensor, from_cl, get_compute_dtype, invalidate_weight_cache, set_compute_dtype,  # noqa: F401
                         to_cl)

__version__ = "0.1.0"
