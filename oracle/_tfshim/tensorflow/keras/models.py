from ._impl import Model, Sequential   # noqa: F401
