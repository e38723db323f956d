from ._impl import Constraint, UnitNorm, _passthrough_get as get, _serialize as serialize   # noqa: F401
