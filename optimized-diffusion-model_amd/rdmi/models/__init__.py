"""Importing the package registers the score models (as the reference's models/__init__.py does)."""
from . import utils, ema, ncsnpp  # noqa: F401
