from ._impl import Regularizer, _passthrough_get as get, _serialize as serialize   # noqa: F401
