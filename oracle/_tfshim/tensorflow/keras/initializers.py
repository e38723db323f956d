from ._impl import Initializer, Zeros, Ones, Constant, GlorotUniform, RandomUniform, Identity, _serialize as serialize   # noqa: F401
from ._impl import get_initializer as get   # noqa: F401
