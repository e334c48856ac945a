"""Drop-in `models` package (registers 'ncsnpp' on import, like the reference's models/__init__.py)."""
from rdmi.models import utils, ema, ncsnpp  # noqa: F401
