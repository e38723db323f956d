"""Keras layers / Model bookkeeping for the stand-in (TEST INFRASTRUCTURE, see ../../README.md).

Restated from Keras' public behaviour (keras 2.8, the version the reference asks for, README.md:221):
* auto-naming: `to_snake_case(class name)` + a process-wide counter per base name, zero based (`dense`, `dense_1`, ...);
* variables are named <name-scope path at build time>/<weight name>:0, scopes nest in eager mode, a layer's scope is
  entered by `__call__`; variables made in `__init__` carry whatever scope the constructor ran in;
* `weights` = trainable then non-trainable, each depth-first in attribute order; `.trainable = False` on a layer moves
  every variable below it to the non-trainable list;
* `training` / `mask` keyword arguments are dropped when the layer's `call` does not take them;
* layer arithmetic: Dense = tensordot + bias, Conv*/SeparableConv1D with TF 'SAME' / Keras 'causal' padding,
  LayerNormalization (axis -1, eps 1e-3, biased variance), BatchNormalization inference form (eps 1e-3),
  MultiHeadAttention (query scaled by 1/sqrt(key_dim) after projection, mask adder -1e9 * (1 - mask) before the softmax).
"""
import inspect
import math
import re

import numpy as np

from .. import _core as C
from .._core import Tensor, Variable, TensorShape, convert_to_tensor, _raw

_UID = {}


def reset_uids():
    _UID.clear()


def to_snake_case(name):
    inter = re.sub("(.)([A-Z][a-z0-9]+)", r"\1_\2", name)
    insecure = re.sub("([a-z])([A-Z])", r"\1_\2", inter).lower()
    return insecure if insecure[0] != "_" else "private" + insecure


def unique_object_name(base):
    n = _UID.get(base, 0)
    _UID[base] = n + 1
    return base if n == 0 else "%s_%d" % (base, n)


# ---------------------------------------------------------------------------------------------------------------------
# initializers / regularizers / constraints / activations
# ---------------------------------------------------------------------------------------------------------------------
class Initializer:
    def __call__(self, shape, dtype=None, **kw):
        raise NotImplementedError

    def get_config(self):
        return {}


class Zeros(Initializer):
    def __call__(self, shape, dtype=None, **kw):
        return C.zeros(shape, dtype or C.float32)


class Ones(Initializer):
    def __call__(self, shape, dtype=None, **kw):
        return C.ones(shape, dtype or C.float32)


class Constant(Initializer):
    def __init__(self, value=0):
        self.value = value

    def __call__(self, shape, dtype=None, **kw):
        dt = C._npdt(dtype or C.float32)
        return Tensor(np.broadcast_to(np.asarray(_raw(self.value), dt), tuple(shape)).copy())


class GlorotUniform(Initializer):
    """values never survive (every test overwrites them); the draw is there so that `_build()` runs on finite numbers"""
    _rng = np.random.default_rng(7)

    def __init__(self, seed=None):
        pass

    def __call__(self, shape, dtype=None, **kw):
        shape = tuple(int(s) for s in shape)
        if len(shape) < 1:
            fan_in = fan_out = 1
        elif len(shape) == 1:
            fan_in = fan_out = shape[0]
        else:
            rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
            fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
        lim = math.sqrt(6.0 / max(1, fan_in + fan_out))
        return Tensor(self._rng.uniform(-lim, lim, shape).astype(C._npdt(dtype or C.float32)))


class RandomUniform(GlorotUniform):
    def __init__(self, minval=-0.05, maxval=0.05, seed=None):
        self.lo, self.hi = minval, maxval

    def __call__(self, shape, dtype=None, **kw):
        return Tensor(self._rng.uniform(self.lo, self.hi, tuple(shape)).astype(C._npdt(dtype or C.float32)))


class Identity(Initializer):
    def __init__(self, gain=1.0):
        self.gain = gain

    def __call__(self, shape, dtype=None, **kw):
        return Tensor((self.gain * np.eye(*shape)).astype(C._npdt(dtype or C.float32)))


_INITS = {"zeros": Zeros, "ones": Ones, "glorot_uniform": GlorotUniform, "uniform": RandomUniform, "random_uniform": RandomUniform}


def get_initializer(x):
    if x is None:
        return None
    if isinstance(x, str):
        return _INITS[x]()
    if isinstance(x, type):
        return x()
    return x


def _passthrough_get(x):
    return x


def _serialize(x):
    return None if x is None else getattr(x, "__class__", type(x)).__name__


class Regularizer:
    pass


class Constraint:
    def __call__(self, w):
        return w


class UnitNorm(Constraint):
    def __init__(self, axis=0):
        self.axis = axis

    def __call__(self, w):
        a = _raw(w)
        return Tensor(a / (1e-7 + np.sqrt(np.sum(np.square(a), axis=self.axis, keepdims=True))))


def linear(x):
    return x


_ACTS = {"relu": C.relu, "sigmoid": C.sigmoid, "linear": linear, None: None, "swish": C.swish, "tanh": C.tanh,
         "softmax": C.softmax}


def get_activation(x):
    if callable(x):
        return x
    return _ACTS[x]


# ---------------------------------------------------------------------------------------------------------------------
# Layer / Model
# ---------------------------------------------------------------------------------------------------------------------
class InputSpec:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def _shapes_of(x):
    if isinstance(x, Tensor):
        return x.shape
    if isinstance(x, (list, tuple)):
        return [_shapes_of(v) for v in x]
    if isinstance(x, np.ndarray):
        return TensorShape(x.shape)
    return None


def _accepts(fn, name):
    try:
        ps = inspect.signature(fn).parameters
    except (TypeError, ValueError):
        return True
    return name in ps or any(p.kind == inspect.Parameter.VAR_KEYWORD for p in ps.values())


class Layer:
    def __init__(self, trainable=True, name=None, dtype=None, dynamic=False, **kwargs):
        bad = set(kwargs) - {"input_shape", "input_dim", "batch_input_shape", "batch_size", "weights", "activity_regularizer",
                             "autocast", "implementation"}
        if bad:
            raise TypeError("Keyword argument not understood: %s" % sorted(bad))
        self._name = name if name else unique_object_name(to_snake_case(self.__class__.__name__))
        self.trainable = trainable
        self.built = False
        self._added_weights = []
        self._dtype = dtype
        self.input_spec = None
        self.supports_masking = False

    name = property(lambda self: self._name)
    dtype = property(lambda self: C.float32)

    # -- variables ------------------------------------------------------------------------------------------------------
    def add_weight(self, name=None, shape=None, dtype=None, initializer=None, regularizer=None, trainable=None,
                   constraint=None, **kw):
        shape = () if shape is None else tuple(int(s) for s in shape)
        init = get_initializer(initializer) or (GlorotUniform() if True else None)
        v = Variable(init(shape, dtype=dtype or C.float32), trainable=True if trainable is None else trainable, name=name,
                     constraint=constraint)
        if tuple(v.shape) != shape:
            raise ValueError("initializer for %s returned shape %s, wanted %s" % (name, tuple(v.shape), shape))
        self._added_weights.append(v)
        return v

    add_variable = add_weight

    def _walk(self, seen, trainable, out_t, out_n):
        if id(self) in seen:
            return
        seen.add(id(self))
        trainable = trainable and bool(self.trainable)

        def visit(o):
            if isinstance(o, Variable):
                if id(o) not in seen:
                    seen.add(id(o))
                    (out_t if (trainable and o.trainable) else out_n).append(o)
            elif isinstance(o, Layer):
                nested.append(o)
            elif isinstance(o, (list, tuple)):
                for v in o:
                    visit(v)
            elif isinstance(o, dict):
                for v in o.values():
                    visit(v)

        nested = []
        for v in self._added_weights:
            visit(v)
        for k, o in list(self.__dict__.items()):
            if k != "_added_weights":
                visit(o)
        for l in nested:
            l._walk(seen, trainable, out_t, out_n)

    def _collect(self):
        t, n = [], []
        self._walk(set(), True, t, n)
        return t, n

    weights = property(lambda self: sum(self._collect(), []))
    variables = weights
    trainable_weights = property(lambda self: self._collect()[0])
    non_trainable_weights = property(lambda self: self._collect()[1])
    trainable_variables = trainable_weights
    non_trainable_variables = non_trainable_weights

    def get_weights(self):
        return [v.numpy() for v in self.weights]

    def set_weights(self, ws):
        vs = self.weights
        assert len(vs) == len(ws)
        for v, w in zip(vs, ws):
            v.assign(w)

    def count_params(self):
        return int(sum(v._a.size for v in self.weights))

    def _sublayers(self):
        out, seen = [], set()

        def visit(o):
            if isinstance(o, Layer):
                if id(o) not in seen:
                    seen.add(id(o))
                    out.append(o)
            elif isinstance(o, (list, tuple)):
                for v in o:
                    visit(v)
            elif isinstance(o, dict):
                for v in o.values():
                    visit(v)
        for k, o in self.__dict__.items():
            visit(o)
        return out

    # -- calling --------------------------------------------------------------------------------------------------------
    def build(self, input_shape):
        self.built = True

    def call(self, inputs, *a, **k):
        return inputs

    def __call__(self, *args, **kwargs):
        with C.name_scope(self._name):
            if not self.built:
                self.build(_shapes_of(args[0]) if args else None)
                self.built = True
            for kw in ("training", "mask"):
                if kw in kwargs and not _accepts(self.call, kw):
                    kwargs.pop(kw)
            return self.call(*args, **kwargs)

    def add_loss(self, *a, **k):
        return None

    def add_update(self, *a, **k):
        return None

    def add_metric(self, *a, **k):
        return None

    def get_config(self):
        return {"name": self._name, "trainable": self.trainable}

    def compute_output_shape(self, input_shape):
        return input_shape


class Model(Layer):
    def __init__(self, *args, **kwargs):
        kwargs.pop("inputs", None), kwargs.pop("outputs", None)
        super().__init__(**kwargs)

    layers = property(lambda self: self._sublayers())

    def summary(self, *a, **k):
        print("Model %s: %d parameters in %d variables" % (self.name, self.count_params(), len(self.weights)))

    def compile(self, *a, **k):
        return None

    # TF-format checkpoints: variables keyed by their attribute path in the object graph (what tf.train.Checkpoint does for a
    # subclassed model: breadth-first over attribute dependencies, shortest path wins, lists indexed by position)
    def _object_graph(self):
        keys, seen = {}, set()
        queue = [("", self)]
        while queue:
            nxt = []
            for path, o in queue:
                if isinstance(o, Variable):
                    if id(o) not in seen:
                        seen.add(id(o))
                        keys[path + "/.ATTRIBUTES/VARIABLE_VALUE"] = o
                    continue
                if id(o) in seen:
                    continue
                seen.add(id(o))
                if isinstance(o, Layer):
                    items = [(k, v) for k, v in o.__dict__.items() if k != "_added_weights"]
                elif isinstance(o, (list, tuple)):
                    items = [(str(i), v) for i, v in enumerate(o)]
                elif isinstance(o, dict):
                    items = [(str(k), v) for k, v in o.items()]
                else:
                    continue
                for k, v in items:
                    if isinstance(v, (Variable, Layer)) or (isinstance(v, (list, tuple, dict)) and _has_trackable(v)):
                        nxt.append(((path + "/" if path else "") + k, v))
            queue = nxt
        return keys

    def save_weights(self, filepath, overwrite=True, save_format=None, options=None):
        keys = self._object_graph()
        np.savez(filepath + ".shim-ckpt.npz", **{k.replace("/", "|"): v.numpy() for k, v in keys.items()})

    def load_weights(self, filepath, by_name=False, **kw):
        d = np.load(filepath + ".shim-ckpt.npz")
        for k, v in self._object_graph().items():
            v.assign(d[k.replace("/", "|")])


def _has_trackable(o):
    if isinstance(o, (Variable, Layer)):
        return True
    if isinstance(o, (list, tuple)):
        return any(_has_trackable(v) for v in o)
    if isinstance(o, dict):
        return any(_has_trackable(v) for v in o.values())
    return False


class Sequential(Model):
    def __init__(self, layers=None, name=None):
        super().__init__(name=name)
        self._layers = list(layers or [])

    def add(self, layer):
        self._layers.append(layer)

    def call(self, inputs, training=None, mask=None):
        x = inputs
        for l in self._layers:
            x = l(x, training=training) if _accepts(l.call, "training") else l(x)
        return x


# ---------------------------------------------------------------------------------------------------------------------
# layers
# ---------------------------------------------------------------------------------------------------------------------
class Dense(Layer):
    def __init__(self, units, activation=None, use_bias=True, kernel_initializer="glorot_uniform", bias_initializer="zeros",
                 kernel_regularizer=None, bias_regularizer=None, activity_regularizer=None, kernel_constraint=None,
                 bias_constraint=None, **kwargs):
        super().__init__(**kwargs)
        self.units, self.activation, self.use_bias = int(units), get_activation(activation), use_bias
        self.kernel_initializer, self.bias_initializer = get_initializer(kernel_initializer), get_initializer(bias_initializer)

    def build(self, input_shape):
        self.kernel = self.add_weight("kernel", shape=[int(input_shape[-1]), self.units], initializer=self.kernel_initializer)
        self.bias = self.add_weight("bias", shape=[self.units], initializer=self.bias_initializer) if self.use_bias else None
        self.built = True

    def call(self, inputs, training=None):
        y = C.tensordot(inputs, self.kernel, [[convert_to_tensor(inputs).ndim - 1], [0]])
        if self.bias is not None:
            y = y + self.bias
        return self.activation(y) if self.activation is not None else y


def _tup(v, n):
    return (int(v),) * n if isinstance(v, (int, np.integer)) else tuple(int(x) for x in v)


class _Conv(Layer):
    RANK = 1

    def __init__(self, filters, kernel_size, strides=1, padding="valid", data_format=None, dilation_rate=1, groups=1,
                 activation=None, use_bias=True, kernel_initializer="glorot_uniform", bias_initializer="zeros",
                 kernel_regularizer=None, bias_regularizer=None, activity_regularizer=None, kernel_constraint=None,
                 bias_constraint=None, **kwargs):
        super().__init__(**kwargs)
        n = self.RANK
        self.filters, self.kernel_size, self.strides = int(filters), _tup(kernel_size, n), _tup(strides, n)
        self.padding, self.dilation_rate, self.groups = padding.lower(), _tup(dilation_rate, n), groups
        self.activation, self.use_bias = get_activation(activation), use_bias
        self.kernel_initializer, self.bias_initializer = get_initializer(kernel_initializer), get_initializer(bias_initializer)
        self.kernel_regularizer = kernel_regularizer
        assert data_format in (None, "channels_last") and groups == 1

    def build(self, input_shape):
        cin = int(input_shape[-1])
        self.kernel = self.add_weight("kernel", shape=self.kernel_size + (cin, self.filters), initializer=self.kernel_initializer)
        self.bias = self.add_weight("bias", shape=(self.filters,), initializer=self.bias_initializer) if self.use_bias else None
        self.built = True

    def _compute_causal_padding(self, inputs=None):
        left = self.dilation_rate[0] * (self.kernel_size[0] - 1)
        return [[0, 0], [left, 0], [0, 0]]

    def _conv(self, x, kernel):
        if self.RANK == 1:
            pad = self.padding
            if pad == "causal":
                x, pad = C.pad(x, self._compute_causal_padding()), "valid"
            return C.nn_conv1d(x, kernel, stride=self.strides[0], padding=pad.upper(), dilations=self.dilation_rate[0])
        return C.nn_conv2d(x, kernel, strides=self.strides, padding=self.padding.upper(), dilations=self.dilation_rate)

    def call(self, inputs, training=None):
        y = self._conv(inputs, self.kernel)
        if self.bias is not None:
            y = y + self.bias
        return self.activation(y) if self.activation is not None else y


class Conv1D(_Conv):
    RANK = 1


class Conv2D(_Conv):
    RANK = 2


class SeparableConv1D(Layer):
    def __init__(self, filters, kernel_size, strides=1, padding="valid", data_format=None, dilation_rate=1, depth_multiplier=1,
                 activation=None, use_bias=True, depthwise_initializer="glorot_uniform", pointwise_initializer="glorot_uniform",
                 bias_initializer="zeros", **kwargs):
        kwargs = {k: v for k, v in kwargs.items() if not k.endswith(("_regularizer", "_constraint"))}
        super().__init__(**kwargs)
        self.filters, self.kernel_size, self.strides = int(filters), _tup(kernel_size, 1), _tup(strides, 1)
        self.padding, self.dilation_rate, self.depth_multiplier = padding.lower(), _tup(dilation_rate, 1), depth_multiplier
        self.activation, self.use_bias = get_activation(activation), use_bias

    def build(self, input_shape):
        cin = int(input_shape[-1])
        self.depthwise_kernel = self.add_weight("depthwise_kernel", shape=self.kernel_size + (cin, self.depth_multiplier))
        self.pointwise_kernel = self.add_weight("pointwise_kernel", shape=(1, cin * self.depth_multiplier, self.filters))
        self.bias = self.add_weight("bias", shape=(self.filters,), initializer="zeros") if self.use_bias else None
        self.built = True

    def call(self, inputs, training=None):
        # keras/layers/convolutional.py SeparableConv1D.call: causal = left pad dilation * (k - 1) then 'valid'; the 1-D
        # problem is run as tf.nn.separable_conv2d on [B, 1, T, C]
        x, pad = inputs, self.padding
        if pad == "causal":
            x, pad = C.pad(x, [[0, 0], [self.dilation_rate[0] * (self.kernel_size[0] - 1), 0], [0, 0]]), "valid"
        xa, dk = C._pair(x, self.depthwise_kernel)
        y = C.depthwise_conv2d_nhwc(xa[:, None], dk[None], (1, self.strides[0]), pad.upper(), (1, self.dilation_rate[0]))
        y = C.conv2d_nhwc(y, _raw(self.pointwise_kernel)[None].astype(y.dtype), (1, 1), "VALID")[:, 0]
        y = Tensor(y)
        if self.bias is not None:
            y = y + self.bias
        return self.activation(y) if self.activation is not None else y


class LayerNormalization(Layer):
    def __init__(self, axis=-1, epsilon=1e-3, center=True, scale=True, beta_initializer="zeros", gamma_initializer="ones", **kwargs):
        super().__init__(**kwargs)
        self.axis, self.epsilon, self.center, self.scale = axis, epsilon, center, scale

    def build(self, input_shape):
        n = int(input_shape[self.axis])
        self.gamma = self.add_weight("gamma", shape=[n], initializer="ones") if self.scale else None
        self.beta = self.add_weight("beta", shape=[n], initializer="zeros") if self.center else None
        self.built = True

    def call(self, inputs):
        a = convert_to_tensor(inputs)._a
        mean = a.mean(axis=self.axis, keepdims=True)
        var = np.mean(np.square(a - mean), axis=self.axis, keepdims=True)
        y = (a - mean) / np.sqrt(var + np.asarray(self.epsilon, a.dtype))
        if self.gamma is not None:
            y = y * _raw(self.gamma)
        if self.beta is not None:
            y = y + _raw(self.beta)
        return Tensor(y.astype(a.dtype, copy=False))


class BatchNormalization(Layer):
    def __init__(self, axis=-1, momentum=0.99, epsilon=1e-3, center=True, scale=True, **kwargs):
        kwargs = {k: v for k, v in kwargs.items() if not k.endswith(("_initializer", "_regularizer", "_constraint"))}
        super().__init__(**kwargs)
        self.axis, self.momentum, self.epsilon, self.center, self.scale = axis, momentum, epsilon, center, scale

    def build(self, input_shape):
        n = int(input_shape[self.axis])
        self.gamma = self.add_weight("gamma", shape=[n], initializer="ones") if self.scale else None
        self.beta = self.add_weight("beta", shape=[n], initializer="zeros") if self.center else None
        self.moving_mean = self.add_weight("moving_mean", shape=[n], initializer="zeros", trainable=False)
        self.moving_variance = self.add_weight("moving_variance", shape=[n], initializer="ones", trainable=False)
        self.built = True

    def call(self, inputs, training=None):
        if training:
            raise NotImplementedError("the stand-in runs BatchNormalization in inference form only")
        assert self.axis in (-1, convert_to_tensor(inputs).ndim - 1)
        return C.batch_normalization(inputs, self.moving_mean, self.moving_variance, self.beta, self.gamma, self.epsilon)


class Dropout(Layer):
    def __init__(self, rate, noise_shape=None, seed=None, **kwargs):
        super().__init__(**kwargs)
        self.rate = rate

    def call(self, inputs, training=None):
        if training and self.rate:
            raise NotImplementedError("the stand-in runs forward passes with dropout off")
        return inputs


class Add(Layer):
    def call(self, inputs):
        out = inputs[0]
        for v in inputs[1:]:
            out = out + v
        return out


class Activation(Layer):
    def __init__(self, activation, **kwargs):
        super().__init__(**kwargs)
        self.activation = get_activation(activation)

    def call(self, inputs):
        return self.activation(inputs)


class LeakyReLU(Layer):
    def __init__(self, alpha=0.3, **kwargs):
        super().__init__(**kwargs)
        self.alpha = alpha

    def call(self, inputs):
        return C.leaky_relu(inputs, self.alpha)


class ReLU(Layer):
    def call(self, inputs):
        return C.relu(inputs)


class Softmax(Layer):
    def __init__(self, axis=-1, **kwargs):
        super().__init__(**kwargs)
        self.axis = axis

    def call(self, inputs, mask=None):
        if mask is not None:
            a = convert_to_tensor(inputs)
            inputs = a + (1.0 - C.cast(mask, a.dtype)) * (-1e9)
        return C.softmax(inputs, axis=self.axis)


class Embedding(Layer):
    def __init__(self, input_dim, output_dim, embeddings_initializer="uniform", mask_zero=False, input_length=None, **kwargs):
        kwargs = {k: v for k, v in kwargs.items() if not k.endswith(("_regularizer", "_constraint"))}
        super().__init__(**kwargs)
        self.input_dim, self.output_dim = int(input_dim), int(output_dim)

    def build(self, input_shape):
        self.embeddings = self.add_weight("embeddings", shape=(self.input_dim, self.output_dim), initializer="uniform")
        self.built = True

    def call(self, inputs):
        return C.gather(self.embeddings, C.cast(inputs, C.int32))


class Reshape(Layer):
    def __init__(self, target_shape, **kwargs):
        super().__init__(**kwargs)
        self.target_shape = tuple(target_shape)

    def call(self, inputs):
        a = convert_to_tensor(inputs)._a
        return Tensor(a.reshape((a.shape[0],) + self.target_shape))


class _Pool1D(Layer):
    def __init__(self, pool_size=2, strides=None, padding="valid", data_format="channels_last", **kwargs):
        super().__init__(**kwargs)
        self.pool_size = int(pool_size if isinstance(pool_size, int) else pool_size[0])
        self.strides = self.pool_size if strides is None else int(strides if isinstance(strides, int) else strides[0])
        self.padding = padding.lower()

    def _win(self, inputs):
        a = convert_to_tensor(inputs)._a
        assert self.padding == "valid", "only 'valid' pooling is on the reference's path"
        return C._windows(a[:, None], 1, self.pool_size, 1, self.strides, "VALID")[:, 0]      # [B, T', C, 1, k]


class AveragePooling1D(_Pool1D):
    def call(self, inputs):
        return Tensor(self._win(inputs).mean(axis=(-1, -2)))


class MaxPool1D(_Pool1D):
    def call(self, inputs):
        return Tensor(self._win(inputs).max(axis=(-1, -2)))


MaxPooling1D = MaxPool1D


class EinsumDense(Layer):
    """keras.layers.experimental.EinsumDense as MultiHeadAttention builds it"""
    def __init__(self, equation, output_shape, bias_axes=None, **kwargs):
        super().__init__(**kwargs)
        self.equation, self.out_shape, self.bias_axes = equation, output_shape, bias_axes

    def build_with(self, kernel_shape, bias_shape):
        with C.name_scope(self._name):
            self.kernel = self.add_weight("kernel", shape=kernel_shape)
            self.bias = self.add_weight("bias", shape=bias_shape, initializer="zeros") if bias_shape is not None else None
        self.built = True

    def call(self, inputs):
        y = C.einsum(self.equation, inputs, self.kernel)
        return y + self.bias if self.bias is not None else y


class MultiHeadAttention(Layer):
    """tf.keras.layers.MultiHeadAttention for 3-D query / value (keras/layers/multi_head_attention.py): sublayers `query`,
    `key`, `value` (kernels [dim, heads, key_dim], bias [heads, key_dim]) and `attention_output` (kernel [heads, key_dim, dim],
    bias [dim]), reachable as `_query_dense` / `_key_dense` / `_value_dense` / `_output_dense`; the query is multiplied by
    1 / sqrt(key_dim) after its projection; `attention_mask` [B, T, S] gains a head axis and enters as
    (1 - mask) * -1e9 added to the scores (the Softmax layer's masked form); dropout is off at inference."""

    def __init__(self, num_heads, key_dim, value_dim=None, dropout=0.0, use_bias=True, output_shape=None, attention_axes=None, **kwargs):
        kwargs = {k: v for k, v in kwargs.items() if not k.endswith(("_initializer", "_regularizer", "_constraint"))}
        super().__init__(**kwargs)
        self._num_heads, self._key_dim, self._value_dim = int(num_heads), int(key_dim), int(value_dim or key_dim)
        self._use_bias, self._dropout = use_bias, dropout
        assert output_shape is None and attention_axes is None
        self._built_from_signature = False

    def _build_from_signature(self, query, value, key=None):
        dq, dv = int(query.shape[-1]), int(value.shape[-1])
        dk = dv if key is None else int(key.shape[-1])
        H, K, V = self._num_heads, self._key_dim, self._value_dim
        self._query_dense = EinsumDense("abc,cde->abde", None, name="query")
        self._key_dense = EinsumDense("abc,cde->abde", None, name="key")
        self._value_dense = EinsumDense("abc,cde->abde", None, name="value")
        self._output_dense = EinsumDense("abcd,cde->abe", None, name="attention_output")
        self._query_dense.build_with((dq, H, K), (H, K) if self._use_bias else None)
        self._key_dense.build_with((dk, H, K), (H, K) if self._use_bias else None)
        self._value_dense.build_with((dv, H, V), (H, V) if self._use_bias else None)
        self._output_dense.build_with((H, V, dq), (dq,) if self._use_bias else None)
        self._softmax = Softmax(axis=-1)
        self._dropout_layer = Dropout(self._dropout)
        self._built_from_signature = True

    def __call__(self, query, value, key=None, attention_mask=None, return_attention_scores=False, training=None, **kw):
        with C.name_scope(self._name):
            query, value = convert_to_tensor(query), convert_to_tensor(value)
            if not self._built_from_signature:
                self._build_from_signature(query, value, key)
                self.built = True
            key = value if key is None else convert_to_tensor(key)
            q = self._query_dense(query)                                   # [B, T, H, K]
            k = self._key_dense(key)                                       # [B, S, H, K]
            v = self._value_dense(value)                                   # [B, S, H, V]
            q = C.multiply(q, 1.0 / math.sqrt(float(self._key_dim)))
            scores = C.einsum("aecd,abcd->acbe", k, q)                     # [B, H, T, S]
            if attention_mask is not None:
                m = convert_to_tensor(attention_mask)
                m = C.expand_dims(m, -3)
                scores = self._softmax(scores, mask=m)
            else:
                scores = self._softmax(scores)
            scores_d = self._dropout_layer(scores, training=training)
            out = C.einsum("acbe,aecd->abcd", scores_d, v)                 # [B, T, H, V]
            out = self._output_dense(out)
            return (out, scores) if return_attention_scores else out


class SimpleRNN(Layer):
    def __init__(self, *a, **k):
        raise NotImplementedError("SimpleRNN (the fixed-smoother PCEN) is not on the reference's default path")


class LSTM(SimpleRNN):
    pass


class GRU(SimpleRNN):
    pass


# ---------------------------------------------------------------------------------------------------------------------
# keras.backend
# ---------------------------------------------------------------------------------------------------------------------
def k_variable(value, dtype=None, name=None, constraint=None):
    # keras.backend.variable: tf.Variable(value, dtype).  The VALUE is rounded to the declared dtype (float32 for K.floatx())
    # in both precisions of the stand-in; the wide mode only carries it in float64 afterwards.
    v = np.asarray(_raw(value))
    if dtype is not None:
        v = v.astype(C.as_dtype(dtype)._narrow)
    return Variable(v, name=name, constraint=constraint, dtype=dtype)


def k_conv2d(x, kernel, strides=(1, 1), padding="valid", data_format=None, dilation_rate=(1, 1)):
    assert data_format in (None, "channels_last")
    return C.nn_conv2d(x, kernel, strides=strides, padding=padding.upper(), dilations=dilation_rate)


def k_dot(x, y):
    a, b = C._pair(x, y)
    if a.ndim > 2 or b.ndim > 2:                       # keras.backend.dot: contracts the last axis of x with the second-to-last of y
        return Tensor(np.tensordot(a, b, axes=[[a.ndim - 1], [max(b.ndim - 2, 0)]]))
    return Tensor(a @ b)


def k_max(x, axis=None, keepdims=False):
    return C.reduce_max(x, axis=axis, keepdims=keepdims)


def k_ctc_decode(y_pred, input_length, greedy=True, beam_width=100, top_paths=1):
    """keras.backend.ctc_decode, greedy: log(transpose(y_pred) + eps) -> tf.nn.ctc_greedy_decoder(merge_repeated=True) ->
    dense, padded with -1 to the longest decoded sequence of the batch"""
    if not greedy:
        raise NotImplementedError
    p = convert_to_tensor(y_pred)._a
    logp = np.log(np.transpose(p, (1, 0, 2)) + 1e-7)
    seqs, score = C.ctc_greedy_decoder(logp, _raw(input_length))
    width = max((len(s) for s in seqs), default=0)
    dense = -np.ones((len(seqs), width), np.int64)
    for i, s in enumerate(seqs):
        dense[i, :len(s)] = s
    return [Tensor(dense)], score
