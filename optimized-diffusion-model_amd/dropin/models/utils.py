"""Drop-in for Reflected-Diffusion/models/utils.py -> rdmi.models.utils"""
from rdmi.models import utils as _impl
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith('__')})
