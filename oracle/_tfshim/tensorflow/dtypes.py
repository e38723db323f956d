from ._core import DType, as_dtype, cast   # noqa: F401
