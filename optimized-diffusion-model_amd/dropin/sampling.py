"""Drop-in module: put optimized-diffusion-model_amd/dropin (and its parent) on sys.path in place of the reference's
Reflected-Diffusion/ directory and `import sampling` resolves to the MI355X-native implementation (rdmi.sampling)."""
from rdmi.sampling import *  # noqa: F401,F403
from rdmi import sampling as _impl
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith('__')})
