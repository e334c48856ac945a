"""Drop-in module: `import datasets` resolves to the MI355X-native counterpart (rdmi.datasets) of the reference's
Reflected-Diffusion/datasets.py for the parts on or next to the hot path (see that module's docstring)."""
from rdmi.datasets import *  # noqa: F401,F403
from rdmi import datasets as _impl
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith('__')})
