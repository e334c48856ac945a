"""Drop-in module: `import utils` resolves to the MI355X-native counterpart (rdmi.utils) of the reference's
Reflected-Diffusion/utils.py for the parts on or next to the hot path (see that module's docstring)."""
from rdmi.utils import *  # noqa: F401,F403
from rdmi import utils as _impl
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith('__')})
