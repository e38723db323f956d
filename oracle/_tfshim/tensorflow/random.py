from ._core import set_seed, random_uniform as uniform, random_normal as normal   # noqa: F401
