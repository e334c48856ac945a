"""Drop-in module: put optimized-diffusion-model_amd/dropin (and its parent) on sys.path in place of the reference's
Reflected-Diffusion/ directory and `import cube` resolves to the MI355X-native implementation (rdmi.cube)."""
from rdmi.cube import *  # noqa: F401,F403
from rdmi import cube as _impl
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith('__')})
