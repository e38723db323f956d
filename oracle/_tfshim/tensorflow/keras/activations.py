from .._core import swish, relu, sigmoid, tanh, softmax   # noqa: F401
from ._impl import linear, get_activation as get   # noqa: F401
