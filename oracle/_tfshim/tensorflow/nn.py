from ._core import (softmax, log_softmax, sigmoid, relu, leaky_relu, swish, bias_add, moments, batch_normalization,   # noqa: F401
                    ctc_greedy_decoder, tanh)
from ._core import nn_conv1d as conv1d, nn_conv2d as conv2d, nn_depthwise_conv2d as depthwise_conv2d   # noqa: F401
silu = swish
