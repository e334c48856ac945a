"""Drop-in for Reflected-Diffusion/models/ncsnpp.py -> rdmi.models.ncsnpp"""
from rdmi.models import ncsnpp as _impl
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith('__')})
