"""rdmi -- MI355X-native Reflected-Diffusion hot path (NCSN++ score network + reflected PC sampler).

Python host mirror of the reference's closure/registry API (Reflected-Diffusion/{sampling,sde_lib,cube,
losses}.py and models/{utils,ncsnpp,ema}.py); all arithmetic runs in librdmi.so (hand-written HIP for
gfx950) through the C ABI in include/rdmi.h.  PyTorch is used for device memory, streams and
torch.distributed only.
"""
from . import _native  # noqa: F401

__all__ = ['cube', 'sde_lib', 'sampling', 'losses', 'models']
