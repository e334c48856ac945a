"""Drop-in for Reflected-Diffusion/models/ema.py -> rdmi.models.ema"""
from rdmi.models import ema as _impl
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith('__')})
