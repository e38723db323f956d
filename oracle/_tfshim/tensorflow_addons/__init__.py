"""tensorflow_addons stand-in (TEST INFRASTRUCTURE): the one layer leaf_audio/frontend.py uses."""
from . import layers   # noqa: F401
