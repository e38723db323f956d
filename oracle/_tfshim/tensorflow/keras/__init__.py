from ._impl import Model, Sequential   # noqa: F401
from . import activations, backend, constraints, initializers, layers, models, regularizers   # noqa: F401
