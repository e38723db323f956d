import numpy as np
from tensorflow.keras._impl import Layer
from tensorflow._core import Tensor, convert_to_tensor, _raw


class InstanceNormalization(Layer):
    """tfa.layers.InstanceNormalization = GroupNormalization with one group per channel (tfa/layers/normalizations.py):
    mean and biased variance over every axis except batch and `axis`, (x - mean) * rsqrt(var + eps) * gamma + beta"""
    def __init__(self, axis=-1, epsilon=1e-3, center=True, scale=True, beta_initializer="zeros", gamma_initializer="ones", **kwargs):
        super().__init__(**kwargs)
        self.axis, self.epsilon, self.center, self.scale = axis, epsilon, center, scale

    def build(self, input_shape):
        n = int(input_shape[self.axis])
        self.gamma = self.add_weight("gamma", shape=(n,), initializer="ones") if self.scale else None
        self.beta = self.add_weight("beta", shape=(n,), initializer="zeros") if self.center else None
        self.built = True

    def call(self, inputs):
        a = convert_to_tensor(inputs)._a
        ax = self.axis % a.ndim
        red = tuple(i for i in range(1, a.ndim) if i != ax)
        mean = a.mean(axis=red, keepdims=True)
        var = a.var(axis=red, keepdims=True)
        bshape = [1] * a.ndim
        bshape[ax] = a.shape[ax]
        y = (a - mean) / np.sqrt(var + np.asarray(self.epsilon, a.dtype))
        if self.gamma is not None:
            y = y * _raw(self.gamma).reshape(bshape)
        if self.beta is not None:
            y = y + _raw(self.beta).reshape(bshape)
        return Tensor(y.astype(a.dtype, copy=False))
