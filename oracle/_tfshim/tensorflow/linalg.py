from ._core import band_part, matmul, einsum, tensordot   # noqa: F401
