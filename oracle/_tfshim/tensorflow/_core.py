"""Eager NumPy tensors and the `tf.*` functions the reference's model code calls.  TEST INFRASTRUCTURE (see ../README.md):
nothing under oracle/ is imported by the product.

What is restated here is TensorFlow's PUBLIC API semantics (shapes, broadcasting, padding rules, argument meaning), so that
the reference's own Python under /root/reference runs unmodified on top of it; every function names the TF symbol it
stands in for.  Arithmetic is NumPy's.  Two precisions:

* narrow (default): tf.float32 is np.float32 -- what a TensorFlow CPU run computes in, up to summation order;
* wide (TFSHIM_WIDE=1 or set_wide(True)): tf.float32 / tf.complex64 are carried as float64 / complex128 while every
  constant the reference rounds to `K.floatx()` stays rounded to float32 -- the reference's composition in (near) exact
  arithmetic, which is what the fp64 oracle must reproduce to 1e-9 rather than to 1e-3.
"""
import math
import os

import numpy as np

_STATE = {"wide": os.environ.get("TFSHIM_WIDE", "0") == "1"}


def set_wide(flag):
    _STATE["wide"] = bool(flag)


def is_wide():
    return _STATE["wide"]


# ---------------------------------------------------------------------------------------------------------------------
# dtypes
# ---------------------------------------------------------------------------------------------------------------------
class DType:
    def __init__(self, name, narrow, wide=None):
        self.name, self._narrow, self._wide = name, np.dtype(narrow), np.dtype(wide if wide is not None else narrow)

    @property
    def as_numpy_dtype(self):
        return (self._wide if _STATE["wide"] else self._narrow).type

    @property
    def np(self):
        return self._wide if _STATE["wide"] else self._narrow

    is_floating = property(lambda self: self._narrow.kind == "f")
    is_complex = property(lambda self: self._narrow.kind == "c")
    is_integer = property(lambda self: self._narrow.kind in "iu")
    is_bool = property(lambda self: self._narrow.kind == "b")

    def __eq__(self, other):
        try:
            return self.name == as_dtype(other).name
        except Exception:
            return False

    def __ne__(self, other):
        return not self == other

    def __hash__(self):
        return hash(self.name)

    def __repr__(self):
        return "tf." + self.name


float16 = DType("float16", np.float16)
float32 = DType("float32", np.float32, np.float64)
float64 = DType("float64", np.float64)
complex64 = DType("complex64", np.complex64, np.complex128)
complex128 = DType("complex128", np.complex128)
int8 = DType("int8", np.int8)
uint8 = DType("uint8", np.uint8)
int16 = DType("int16", np.int16)
int32 = DType("int32", np.int32)
int64 = DType("int64", np.int64)
bool_ = DType("bool", np.bool_)
string = DType("string", np.object_)
_ALL = [float16, float32, float64, complex64, complex128, int8, uint8, int16, int32, int64, bool_, string]
_BY_NAME = {d.name: d for d in _ALL}


def as_dtype(x):
    if isinstance(x, DType):
        return x
    if isinstance(x, str):
        return _BY_NAME[x]
    dt = np.dtype(x)
    if _STATE["wide"]:                 # the carried type of a float32 tensor is float64: report the TF-visible type
        if dt == np.float64:
            return float32
        if dt == np.complex128:
            return complex64
    for d in _ALL:
        if d._narrow == dt:
            return d
    raise TypeError("no tf dtype for %r" % (x,))


def _npdt(dtype):
    return None if dtype is None else as_dtype(dtype).np


def _floatx():
    return float32.np


def _complexx():
    return complex64.np


# ---------------------------------------------------------------------------------------------------------------------
# TensorShape / TensorSpec
# ---------------------------------------------------------------------------------------------------------------------
class TensorShape(tuple):
    def __new__(cls, dims=()):
        if dims is None:
            dims = ()
        return super().__new__(cls, tuple(None if d is None else int(d) for d in dims))

    def as_list(self):
        return list(self)

    ndims = property(lambda self: len(self))
    rank = property(lambda self: len(self))

    def __getitem__(self, i):
        r = tuple.__getitem__(self, i)
        return TensorShape(r) if isinstance(i, slice) else r

    def __add__(self, other):
        return TensorShape(tuple(self) + tuple(other))

    def __radd__(self, other):
        return TensorShape(tuple(other) + tuple(self))

    def num_elements(self):
        return int(np.prod(self)) if all(d is not None for d in self) else None

    def is_fully_defined(self):
        return all(d is not None for d in self)


class TensorSpec:
    def __init__(self, shape=None, dtype=float32, name=None):
        self.shape, self.dtype, self.name = TensorShape(shape), as_dtype(dtype), name


# ---------------------------------------------------------------------------------------------------------------------
# Tensor / Variable
# ---------------------------------------------------------------------------------------------------------------------
def _is_t(x):
    return isinstance(x, Tensor)


def _raw(x):
    """anything -> np.ndarray (no dtype policy)"""
    if isinstance(x, Tensor):
        return x._a
    if isinstance(x, (list, tuple)):
        if any(isinstance(v, (Tensor, list, tuple)) for v in x):
            return np.asarray([_raw(v) for v in x])
        return np.asarray(x)
    return np.asarray(x)


def _default(a):
    """dtype policy for values that enter without a dtype (tf.constant / convert_to_tensor of Python and NumPy data)"""
    if a.dtype.kind == "f":
        return a.astype(_floatx(), copy=False)
    if a.dtype.kind == "c":
        return a.astype(_complexx(), copy=False)
    return a


def _pyvalue_dtype(value, a):
    # Python ints -> int32 (tf.constant(3).dtype == int32); NumPy integer arrays keep their width
    if a.dtype.kind in "iu" and not isinstance(value, (np.ndarray, np.generic)):
        src = value
        while isinstance(src, (list, tuple)) and len(src):
            src = src[0]
        if not isinstance(src, (np.ndarray, np.generic, Tensor)):
            return a.astype(np.int32)
    return a


def convert_to_tensor(value, dtype=None, name=None, dtype_hint=None):
    if isinstance(value, Tensor):
        if dtype is not None and value._a.dtype != _npdt(dtype):
            return Tensor(value._a.astype(_npdt(dtype)))
        return value.value() if isinstance(value, Variable) else value
    a = _raw(value)
    if dtype is not None:
        return Tensor(a.astype(_npdt(dtype)))
    return Tensor(_pyvalue_dtype(value, _default(a)))


def _coerce(x, dt):
    """a Python / NumPy operand meeting a tensor of dtype dt takes the tensor's dtype (TF converts constants that way)"""
    x = np.asarray(x)
    if x.dtype == dt:
        return x
    if dt.kind == "c":
        return x.astype(dt)
    if dt.kind == "f":
        return x.astype(dt) if x.dtype.kind in "iubf" else x
    if dt.kind in "iu":
        return x.astype(dt) if x.dtype.kind in "iub" else _default(x)
    return x


def _pair(a, b):
    ta, tb = isinstance(a, Tensor), isinstance(b, Tensor)
    ua, ub = _raw(a), _raw(b)
    if ta and not tb:
        ub = _coerce(ub, ua.dtype)
    elif tb and not ta:
        ua = _coerce(ua, ub.dtype)
    elif not ta and not tb:
        ua, ub = _default(ua), _default(ub)
    return ua, ub


def _idx(k):
    if isinstance(k, Tensor):
        a = k._a
        return a.item() if a.ndim == 0 else a
    if isinstance(k, slice):
        return slice(_idx(k.start), _idx(k.stop), _idx(k.step))
    if isinstance(k, tuple):
        return tuple(_idx(v) for v in k)
    if isinstance(k, list):
        return [_idx(v) for v in k]
    return k


class Tensor:
    __array_priority__ = 1000

    def __init__(self, a):
        self._a = a if isinstance(a, np.ndarray) else np.asarray(a)

    # -- views ---------------------------------------------------------------------------------------------------------
    shape = property(lambda self: TensorShape(self._a.shape))
    dtype = property(lambda self: as_dtype(self._a.dtype))
    ndim = property(lambda self: self._a.ndim)

    def get_shape(self):
        return self.shape

    def numpy(self):
        a = self._a
        return a.item() if a.dtype == object and a.ndim == 0 else (a.copy() if a.ndim else a[()])

    def __array__(self, dtype=None, copy=None):
        return self._a if dtype is None else self._a.astype(dtype)

    def __repr__(self):
        return "<shim tf.Tensor shape=%s dtype=%s>" % (tuple(self._a.shape), self.dtype.name)

    def __len__(self):
        return self._a.shape[0]

    def __iter__(self):
        return (Tensor(v) for v in self._a)

    def __bool__(self):
        return bool(self._a)

    def __int__(self):
        return int(self._a)

    def __index__(self):
        return int(self._a)

    def __float__(self):
        return float(self._a)

    def __hash__(self):
        return id(self)

    def __getitem__(self, k):
        return Tensor(self._a[_idx(k)])

    def set_shape(self, shape):
        return None

    # -- arithmetic -----------------------------------------------------------------------------------------------------
    def _b(self, other, f, r=False):
        a, b = _pair(other, self) if r else _pair(self, other)
        return Tensor(np.asarray(f(a, b)))

    __add__ = lambda s, o: s._b(o, np.add)
    __radd__ = lambda s, o: s._b(o, np.add, True)
    __sub__ = lambda s, o: s._b(o, np.subtract)
    __rsub__ = lambda s, o: s._b(o, np.subtract, True)
    __mul__ = lambda s, o: s._b(o, np.multiply)
    __rmul__ = lambda s, o: s._b(o, np.multiply, True)
    __truediv__ = lambda s, o: s._b(o, _truediv)
    __rtruediv__ = lambda s, o: s._b(o, _truediv, True)
    __floordiv__ = lambda s, o: s._b(o, np.floor_divide)
    __rfloordiv__ = lambda s, o: s._b(o, np.floor_divide, True)
    __mod__ = lambda s, o: s._b(o, np.mod)
    __pow__ = lambda s, o: s._b(o, np.power)
    __rpow__ = lambda s, o: s._b(o, np.power, True)
    __matmul__ = lambda s, o: s._b(o, np.matmul)
    __lt__ = lambda s, o: s._b(o, np.less)
    __le__ = lambda s, o: s._b(o, np.less_equal)
    __gt__ = lambda s, o: s._b(o, np.greater)
    __ge__ = lambda s, o: s._b(o, np.greater_equal)
    __eq__ = lambda s, o: s._b(o, np.equal) if o is not None else False
    __ne__ = lambda s, o: s._b(o, np.not_equal) if o is not None else True
    __and__ = lambda s, o: s._b(o, np.logical_and)
    __or__ = lambda s, o: s._b(o, np.logical_or)
    __invert__ = lambda s: Tensor(np.logical_not(s._a))
    __neg__ = lambda s: Tensor(-s._a)
    __abs__ = lambda s: Tensor(np.abs(s._a))


def _truediv(a, b):
    if a.dtype.kind in "iub" and b.dtype.kind in "iub":          # tf.truediv on ints -> float64 in TF; floatx is enough here
        return np.true_divide(a, b).astype(np.float64)
    return np.true_divide(a, b)


_NAME_SCOPE = []


class name_scope:
    def __init__(self, name, *a, **k):
        self.name = name

    def __enter__(self):
        _NAME_SCOPE.append(self.name)
        return "/".join(_NAME_SCOPE) + "/"

    def __exit__(self, *exc):
        _NAME_SCOPE.pop()
        return False


def current_scope():
    return "/".join(n for n in _NAME_SCOPE if n)


class Variable(Tensor):
    """tf.Variable: a named, assignable tensor.  Eager TF does not uniquify variable names; the name is the name-scope path
    at creation + the given name + ':0'."""

    def __init__(self, initial_value, trainable=True, name=None, dtype=None, constraint=None, shape=None, **kw):
        if callable(initial_value):
            initial_value = initial_value()
        super().__init__(np.array(convert_to_tensor(initial_value, dtype=dtype)._a))
        scope = current_scope()
        self.name = (scope + "/" if scope else "") + (name or "Variable") + ":0"
        self.trainable, self.constraint = bool(trainable), constraint

    def assign(self, value, **kw):
        v = _raw(value)
        if tuple(v.shape) != tuple(self._a.shape):
            raise ValueError("assign to %s: shape %s != %s" % (self.name, tuple(v.shape), tuple(self._a.shape)))
        self._a = np.array(v).astype(self._a.dtype)
        return self

    def assign_add(self, d, **kw):
        return self.assign(self._a + _raw(d))

    def value(self):
        return Tensor(self._a)

    def read_value(self):
        return Tensor(self._a)

    def __repr__(self):
        return "<shim tf.Variable %s shape=%s>" % (self.name, tuple(self._a.shape))


newaxis = None


# ---------------------------------------------------------------------------------------------------------------------
# creation / shape
# ---------------------------------------------------------------------------------------------------------------------
def constant(value, dtype=None, shape=None, name=None):
    t = convert_to_tensor(value, dtype=dtype)
    if shape is not None:
        t = Tensor(np.broadcast_to(t._a, _shape_arg(shape)).copy() if t._a.size == 1 else t._a.reshape(_shape_arg(shape)))
    return t


def _shape_arg(shape):
    if isinstance(shape, Tensor):
        return tuple(int(v) for v in np.atleast_1d(shape._a))
    if isinstance(shape, (int, np.integer)):
        return (int(shape),)
    return tuple(int(_raw(v)) for v in shape)


def zeros(shape, dtype=float32, name=None):
    return Tensor(np.zeros(_shape_arg(shape), _npdt(dtype)))


def ones(shape, dtype=float32, name=None):
    return Tensor(np.ones(_shape_arg(shape), _npdt(dtype)))


def fill(dims, value):
    v = convert_to_tensor(value)._a
    return Tensor(np.full(_shape_arg(dims), v, v.dtype))


def zeros_like(x, dtype=None):
    a = convert_to_tensor(x)._a
    return Tensor(np.zeros(a.shape, _npdt(dtype) or a.dtype))


def ones_like(x, dtype=None):
    a = convert_to_tensor(x)._a
    return Tensor(np.ones(a.shape, _npdt(dtype) or a.dtype))


def range_(start, limit=None, delta=1, dtype=None, name=None):
    s, l, d = _raw(start), (None if limit is None else _raw(limit)), _raw(delta)
    a = np.arange(s, None if l is None else l, d) if l is not None else np.arange(s)
    if dtype is not None:
        a = a.astype(_npdt(dtype))
    elif a.dtype.kind in "iu":
        a = a.astype(np.int32)
    else:
        a = _default(a)
    return Tensor(a)


def shape(x, out_type=int32):
    return Tensor(np.asarray(convert_to_tensor(x)._a.shape, _npdt(out_type)))


def size(x):
    return Tensor(np.asarray(convert_to_tensor(x)._a.size, np.int32))


def rank(x):
    return Tensor(np.asarray(convert_to_tensor(x)._a.ndim, np.int32))


def reshape(x, shape, name=None):
    return Tensor(convert_to_tensor(x)._a.reshape(_shape_arg(shape)))


def transpose(x, perm=None, name=None):
    a = convert_to_tensor(x)._a
    return Tensor(np.transpose(a, None if perm is None else [int(p) for p in _raw(perm)]))


def expand_dims(x, axis, name=None):
    a = convert_to_tensor(x)._a
    for ax in (axis if isinstance(axis, (list, tuple)) else [axis]):
        a = np.expand_dims(a, int(_raw(ax)))
    return Tensor(a)


def squeeze(x, axis=None, name=None):
    a = convert_to_tensor(x)._a
    if axis is None:
        return Tensor(np.squeeze(a))
    return Tensor(np.squeeze(a, tuple(int(v) for v in (axis if isinstance(axis, (list, tuple)) else [axis]))))


def concat(values, axis, name=None):
    arrs = [convert_to_tensor(v)._a for v in values]
    dt = next((a.dtype for a, v in zip(arrs, values) if isinstance(v, Tensor)), None)
    if dt is not None:
        arrs = [_coerce(a, dt) for a in arrs]
    return Tensor(np.concatenate(arrs, int(_raw(axis))))


def stack(values, axis=0, name=None):
    if isinstance(values, Tensor):
        return Tensor(values._a.copy())
    arrs = [convert_to_tensor(v)._a for v in values]
    return Tensor(np.stack(arrs, int(axis)))


def unstack(x, num=None, axis=0):
    a = convert_to_tensor(x)._a
    return [Tensor(v) for v in np.moveaxis(a, axis, 0)]


def split(value, num_or_size_splits, axis=0, num=None, name=None):
    a = convert_to_tensor(value)._a
    if isinstance(num_or_size_splits, (int, np.integer)):
        return [Tensor(v) for v in np.split(a, int(num_or_size_splits), int(axis))]
    sizes = [int(v) for v in _raw(num_or_size_splits)]
    return [Tensor(v) for v in np.split(a, np.cumsum(sizes)[:-1], int(axis))]


def pad(x, paddings, mode="CONSTANT", constant_values=0, name=None):
    a = convert_to_tensor(x)._a
    p = [(int(_raw(lo)), int(_raw(hi))) for lo, hi in (_raw(paddings).tolist() if isinstance(paddings, Tensor) else paddings)]
    mode = mode.upper()
    if mode == "CONSTANT":
        return Tensor(np.pad(a, p, mode="constant", constant_values=_coerce(_raw(constant_values), a.dtype)))
    return Tensor(np.pad(a, p, mode={"REFLECT": "reflect", "SYMMETRIC": "symmetric"}[mode]))


def tile(x, multiples):
    return Tensor(np.tile(convert_to_tensor(x)._a, _shape_arg(multiples)))


def repeat(x, repeats, axis=None, name=None):
    a = convert_to_tensor(x)._a
    r = _raw(repeats)
    return Tensor(np.repeat(a, r.item() if r.ndim == 0 else r, axis))


def roll(x, shift, axis):
    return Tensor(np.roll(convert_to_tensor(x)._a, _idx(shift) if isinstance(shift, Tensor) else shift, axis))


def reverse(x, axis):
    return Tensor(np.flip(convert_to_tensor(x)._a, tuple(int(v) for v in _raw(axis))))


def gather(params, indices, axis=None, batch_dims=0, name=None):
    a = convert_to_tensor(params)._a
    return Tensor(np.take(a, _raw(indices), axis=0 if axis is None else int(axis)))


def dynamic_stitch(indices, data):
    idx = [np.asarray(_raw(i)) for i in indices]
    dat = [_raw(d) for d in data]
    n = max(int(i.max()) for i in idx if i.size) + 1
    out = np.zeros((n,) + dat[0].shape[idx[0].ndim:], dat[0].dtype)
    for i, d in zip(idx, dat):
        out[i] = d
    return Tensor(out)


def cast(x, dtype, name=None):
    a = convert_to_tensor(x)._a
    dt = _npdt(dtype)
    if a.dtype.kind == "c" and dt.kind != "c":
        a = a.real
    return Tensor(a.astype(dt))


def identity(x, name=None):
    return Tensor(convert_to_tensor(x)._a.copy())


def stop_gradient(x):
    return convert_to_tensor(x)


def sequence_mask(lengths, maxlen=None, dtype=bool_):
    ln = _raw(lengths)
    m = int(ln.max()) if maxlen is None else int(_raw(maxlen))
    return Tensor((np.arange(m) < ln[..., None]).astype(_npdt(dtype)))


# ---------------------------------------------------------------------------------------------------------------------
# element-wise / reductions
# ---------------------------------------------------------------------------------------------------------------------
def _un(f):
    def g(x, name=None):
        return Tensor(np.asarray(f(convert_to_tensor(x)._a)))
    return g


def _bi(f):
    def g(x, y, name=None):
        a, b = _pair(x, y)
        return Tensor(np.asarray(f(a, b)))
    return g


sqrt, exp, log, sin, cos, tanh, abs_, square, sign, floor, ceil, round_ = (
    _un(np.sqrt), _un(np.exp), _un(np.log), _un(np.sin), _un(np.cos), _un(np.tanh), _un(np.abs), _un(np.square), _un(np.sign),
    _un(np.floor), _un(np.ceil), _un(np.rint))
rsqrt = _un(lambda a: 1.0 / np.sqrt(a))
negative = _un(np.negative)
real, imag, conj = _un(np.real), _un(np.imag), _un(np.conj)
is_nan, is_inf = _un(np.isnan), _un(np.isinf)
logical_not = _un(np.logical_not)
add, subtract, multiply, maximum, minimum, pow_ = _bi(np.add), _bi(np.subtract), _bi(np.multiply), _bi(np.maximum), _bi(np.minimum), _bi(np.power)
divide = _bi(_truediv)
floordiv = _bi(np.floor_divide)
less, less_equal, greater, greater_equal, equal, not_equal = (_bi(np.less), _bi(np.less_equal), _bi(np.greater),
                                                              _bi(np.greater_equal), _bi(np.equal), _bi(np.not_equal))
logical_and, logical_or = _bi(np.logical_and), _bi(np.logical_or)


def sigmoid(x, name=None):
    a = convert_to_tensor(x)._a
    return Tensor(1.0 / (1.0 + np.exp(-a)))


def _axis(axis):
    if axis is None:
        return None
    a = _raw(axis)
    return int(a) if a.ndim == 0 else tuple(int(v) for v in a)


def _red(f):
    def g(x, axis=None, keepdims=False, name=None):
        a = _raw(x) if not isinstance(x, Tensor) else x._a
        if not isinstance(x, Tensor):
            a = _pyvalue_dtype(x, _default(a))
        return Tensor(np.asarray(f(a, axis=_axis(axis), keepdims=keepdims)))
    return g


reduce_sum, reduce_max, reduce_min, reduce_mean, reduce_prod = _red(np.sum), _red(np.max), _red(np.min), _red(np.mean), _red(np.prod)
reduce_any, reduce_all = _red(np.any), _red(np.all)


def argmax(x, axis=None, output_type=int64, name=None):
    a = convert_to_tensor(x)._a
    return Tensor(np.argmax(a, 0 if axis is None else int(axis)).astype(_npdt(output_type)))


def argsort(x, axis=-1, direction="ASCENDING", stable=False):
    a = convert_to_tensor(x)._a
    i = np.argsort(a if direction == "ASCENDING" else -a, axis=axis, kind="stable")
    return Tensor(i.astype(np.int32))


def where(condition, x=None, y=None, name=None):
    c = _raw(condition)
    if x is None and y is None:
        return Tensor(np.argwhere(c).astype(np.int64))
    a, b = _pair(x, y)
    if not isinstance(x, Tensor) and not isinstance(y, Tensor):
        a, b = _pyvalue_dtype(x, a), _pyvalue_dtype(y, b)
    return Tensor(np.where(c, a, b))


def clip_by_value(x, clip_value_min, clip_value_max, name=None):
    a = convert_to_tensor(x)._a
    return Tensor(np.clip(a, _coerce(_raw(clip_value_min), a.dtype), _coerce(_raw(clip_value_max), a.dtype)))


def _einsum_eq(eq):
    return eq.replace(" ", "")


def einsum(equation, *inputs, **kw):
    arrs = [convert_to_tensor(v)._a for v in inputs]
    return Tensor(np.einsum(_einsum_eq(equation), *arrs))


def tensordot(a, b, axes, name=None):
    ua, ub = _pair(a, b)
    return Tensor(np.asarray(np.tensordot(ua, ub, axes=axes if isinstance(axes, int) else [list(v) if isinstance(v, (list, tuple)) else v for v in axes])))


def matmul(a, b, transpose_a=False, transpose_b=False, name=None):
    ua, ub = _pair(a, b)
    if transpose_a:
        ua = np.swapaxes(ua, -1, -2)
    if transpose_b:
        ub = np.swapaxes(ub, -1, -2)
    return Tensor(np.matmul(ua, ub))


def band_part(x, num_lower, num_upper, name=None):
    a = convert_to_tensor(x)._a
    m, n = a.shape[-2:]
    i, j = np.arange(m)[:, None], np.arange(n)[None, :]
    keep = ((num_lower < 0) | (i - j <= num_lower)) & ((num_upper < 0) | (j - i <= num_upper))
    return Tensor(np.where(keep, a, np.zeros((), a.dtype)))


def complex_(re, im):
    a, b = _pair(re, im)
    return Tensor((a + 1j * b).astype(_complexx()))


# ---------------------------------------------------------------------------------------------------------------------
# control flow (eager)
# ---------------------------------------------------------------------------------------------------------------------
def while_loop(cond, body, loop_vars, shape_invariants=None, **kw):
    v = list(loop_vars)
    while bool(cond(*v)):
        v = list(body(*v))
    return v


def scan(fn, elems, initializer=None, **kw):
    e = convert_to_tensor(elems)
    acc = convert_to_tensor(initializer) if initializer is not None else e[0]
    out = []
    for i in range(0 if initializer is not None else 1, e._a.shape[0]):
        acc = convert_to_tensor(fn(acc, e[i]))
        out.append(acc._a)
    return Tensor(np.stack(out, 0))


def cond(pred, true_fn, false_fn, name=None):
    return true_fn() if bool(_raw(pred)) else false_fn()


def function(func=None, input_signature=None, **kw):
    """tf.function: eager here -- the traced function is the Python function"""
    if func is None:
        return lambda f: f
    return func


def print_(*args, **kw):
    kw.pop("output_stream", None)
    print(*[a.numpy() if isinstance(a, Tensor) else a for a in args], **{k: v for k, v in kw.items() if k in ("sep", "end")})


def numpy_function(func, inp, Tout, name=None):
    return func(*[_raw(i) for i in inp])


class GradientTape:
    def __enter__(self):
        raise NotImplementedError("the stand-in runs forward passes only")

    def __exit__(self, *a):
        return False


# ---------------------------------------------------------------------------------------------------------------------
# convolution (tf.nn.conv2d / depthwise_conv2d semantics: NHWC input, HWIO filters, TF 'SAME' = out ceil(n / s),
# pad_total = max((out - 1) s + (k - 1) d + 1 - n, 0), the smaller half in front)
# ---------------------------------------------------------------------------------------------------------------------
def same_pads(n, k, s, d=1):
    out = -(-n // s)
    total = max((out - 1) * s + (k - 1) * d + 1 - n, 0)
    return total // 2, total - total // 2


def _windows(x, kh, kw, sh, sw, padding, dh=1, dw=1):
    """x [B,H,W,C] -> windows [B,H',W',C,kh,kw] for the given stride / dilation / padding"""
    B, H, W, C = x.shape
    if isinstance(padding, str):
        padding = padding.upper()
        if padding == "SAME":
            ph, pw = same_pads(H, kh, sh, dh), same_pads(W, kw, sw, dw)
        elif padding == "VALID":
            ph = pw = (0, 0)
        else:
            raise ValueError(padding)
    else:
        ph, pw = padding
    if ph != (0, 0) or pw != (0, 0):
        x = np.pad(x, ((0, 0), ph, pw, (0, 0)))
    eh, ew = (kh - 1) * dh + 1, (kw - 1) * dw + 1
    if x.shape[1] < eh or x.shape[2] < ew:
        return np.zeros((B, 0 if x.shape[1] < eh else (x.shape[1] - eh) // sh + 1, 0 if x.shape[2] < ew else (x.shape[2] - ew) // sw + 1, C, kh, kw), x.dtype)
    win = np.lib.stride_tricks.sliding_window_view(x, (eh, ew), axis=(1, 2))      # [B, H-eh+1, W-ew+1, C, eh, ew]
    return win[:, ::sh, ::sw, :, ::dh, ::dw]


def conv2d_nhwc(x, w, strides=(1, 1), padding="VALID", dilations=(1, 1)):
    kh, kw, ci, co = w.shape
    win = _windows(x, kh, kw, strides[0], strides[1], padding, dilations[0], dilations[1])
    return np.einsum("bhwcij,ijco->bhwo", win, w, optimize=True)


def depthwise_conv2d_nhwc(x, w, strides=(1, 1), padding="VALID", dilations=(1, 1)):
    kh, kw, ci, mult = w.shape
    win = _windows(x, kh, kw, strides[0], strides[1], padding, dilations[0], dilations[1])
    out = np.einsum("bhwcij,ijcm->bhwcm", win, w, optimize=True)
    return out.reshape(out.shape[:3] + (ci * mult,))


def _stride2(s):
    if isinstance(s, (int, np.integer)):
        return int(s), int(s)
    s = [int(v) for v in s]
    return (s[1], s[2]) if len(s) == 4 else (s[0], s[1])


def nn_conv2d(input, filters, strides=1, padding="VALID", data_format="NHWC", dilations=None, name=None):
    assert data_format in ("NHWC", "channels_last", None)
    x, w = _pair(input, filters)
    return Tensor(conv2d_nhwc(x, w, _stride2(strides), padding, _stride2(dilations or 1)))


def nn_depthwise_conv2d(input, filter, strides, padding, data_format=None, dilations=None, name=None):
    x, w = _pair(input, filter)
    return Tensor(depthwise_conv2d_nhwc(x, w, _stride2(strides), padding, _stride2(dilations or 1)))


def nn_conv1d(input, filters, stride=1, padding="VALID", data_format="NWC", dilations=None, name=None):
    assert data_format in ("NWC", None)
    x, w = _pair(input, filters)
    s = stride if isinstance(stride, (int, np.integer)) else [int(v) for v in stride][-2 if len(stride) == 3 else 0]
    d = 1 if dilations is None else (dilations if isinstance(dilations, int) else list(dilations)[0])
    y = conv2d_nhwc(x[:, None], w[None], (1, int(s)), padding if isinstance(padding, str) else ((0, 0), padding), (1, int(d)))
    return Tensor(y[:, 0])


def bias_add(value, bias, data_format=None, name=None):
    a, b = _pair(value, bias)
    return Tensor(a + b)


def softmax(logits, axis=-1, name=None):
    a = convert_to_tensor(logits)._a
    e = np.exp(a - a.max(axis=axis, keepdims=True))
    return Tensor(e / e.sum(axis=axis, keepdims=True))


def log_softmax(logits, axis=-1, name=None):
    a = convert_to_tensor(logits)._a
    z = a - a.max(axis=axis, keepdims=True)
    return Tensor(z - np.log(np.exp(z).sum(axis=axis, keepdims=True)))


def relu(x, name=None):
    a = convert_to_tensor(x)._a
    return Tensor(np.maximum(a, np.zeros((), a.dtype)))


def leaky_relu(x, alpha=0.2, name=None):
    a = convert_to_tensor(x)._a
    return Tensor(np.where(a >= 0, a, a * np.asarray(alpha, a.dtype) if a.dtype.kind == "f" else a * alpha))


def swish(x, beta=1.0):
    a = convert_to_tensor(x)._a
    return Tensor(a / (1.0 + np.exp(-a)))


def moments(x, axes, keepdims=False, name=None):
    a = convert_to_tensor(x)._a
    ax = _axis(axes)
    return Tensor(a.mean(axis=ax, keepdims=keepdims)), Tensor(a.var(axis=ax, keepdims=keepdims))


def batch_normalization(x, mean, variance, offset, scale, variance_epsilon, name=None):
    a = convert_to_tensor(x)._a
    inv = 1.0 / np.sqrt(_raw(variance) + variance_epsilon)
    if scale is not None:
        inv = inv * _raw(scale)
    y = a * inv + ((_raw(offset) if offset is not None else 0.0) - _raw(mean) * inv)
    return Tensor(y.astype(a.dtype, copy=False))


def ctc_greedy_decoder(inputs, sequence_length, merge_repeated=True, blank_index=None):
    """tf.nn.ctc_greedy_decoder on time-major [T, B, V] scores: per frame the FIRST maximal class, repeats merged, blanks
    dropped (TF: ctc_decoder_ops.cc CTCGreedyDecoderOp, `max_coeff` + merge_repeated_; blank default = last class).
    Returns per-utterance id lists and the negated sum of the per-frame maxima."""
    a = _raw(inputs)
    T, B, V = a.shape
    blank = V - 1 if blank_index is None else (blank_index if blank_index >= 0 else V + blank_index)
    lens = _raw(sequence_length)
    out, score = [], np.zeros((B, 1), a.dtype)
    for b in range(B):
        seq, prev = [], -1
        for t in range(int(lens[b])):
            k = int(np.argmax(a[t, b]))
            score[b, 0] += -a[t, b, k]
            if k != blank and not (merge_repeated and k == prev):
                seq.append(k)
            prev = k
        out.append(seq)
    return out, Tensor(score)


# ---------------------------------------------------------------------------------------------------------------------
# tf.signal
# ---------------------------------------------------------------------------------------------------------------------
def _fft_length(fft_length, n):
    if fft_length is None:
        return n
    v = _raw(fft_length)
    return int(v.reshape(-1)[0])


def rfft(x, fft_length=None, name=None):
    a = convert_to_tensor(x)._a
    return Tensor(np.fft.rfft(a, n=_fft_length(fft_length, a.shape[-1]), axis=-1).astype(_complexx()))


def irfft(x, fft_length=None, name=None):
    a = convert_to_tensor(x)._a
    n = _fft_length(fft_length, 2 * (a.shape[-1] - 1))
    return Tensor(np.fft.irfft(a, n=n, axis=-1).astype(_floatx()))


def fft(x, name=None):
    return Tensor(np.fft.fft(convert_to_tensor(x)._a, axis=-1).astype(_complexx()))


def fftshift(x, axes=None, name=None):
    return Tensor(np.fft.fftshift(convert_to_tensor(x)._a, axes=axes))


def hann_window(window_length, periodic=True, dtype=float32, name=None):
    n = int(window_length)
    d = n if periodic else n - 1
    return Tensor((0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / d)).astype(_npdt(dtype)))


def frame(signal, frame_length, frame_step, pad_end=False, pad_value=0, axis=-1, name=None):
    a = convert_to_tensor(signal)._a
    assert axis in (-1, a.ndim - 1)
    fl, fs = int(_raw(frame_length)), int(_raw(frame_step))
    n = a.shape[-1]
    if pad_end:
        nf = -(-n // fs)
        need = (nf - 1) * fs + fl
        a = np.pad(a, [(0, 0)] * (a.ndim - 1) + [(0, max(need - n, 0))], constant_values=pad_value)
    else:
        nf = max(0, 1 + (n - fl) // fs)
    idx = np.arange(nf)[:, None] * fs + np.arange(fl)[None, :]
    return Tensor(a[..., idx])


def overlap_and_add(signal, frame_step, name=None):
    a = convert_to_tensor(signal)._a
    fs = int(_raw(frame_step))
    nf, fl = a.shape[-2:]
    out = np.zeros(a.shape[:-2] + ((nf - 1) * fs + fl,), a.dtype)
    for i in range(nf):
        out[..., i * fs:i * fs + fl] += a[..., i, :]
    return Tensor(out)


def stft(signals, frame_length, frame_step, fft_length=None, window_fn=hann_window, pad_end=False, name=None):
    fl = int(frame_length)
    n = fft_length or (1 << (fl - 1).bit_length())
    fr = frame(signals, fl, frame_step, pad_end=pad_end)._a
    if window_fn is not None:
        fr = fr * window_fn(fl, dtype=as_dtype(fr.dtype))._a
    return Tensor(np.fft.rfft(fr, n=int(n), axis=-1).astype(_complexx()))


def linear_to_mel_weight_matrix(num_mel_bins=20, num_spectrogram_bins=129, sample_rate=8000, lower_edge_hertz=125.0,
                                upper_edge_hertz=3800.0, dtype=float32, name=None):
    """tf.signal.linear_to_mel_weight_matrix (mel_ops.py): HTK mel scale 1127 ln(1 + f / 700); the DC bin is dropped and
    zero-padded back; band edges = linspace over mel of num_mel_bins + 2 points; triangles are linear IN MEL; computed in
    float64 and cast (TF does the same since 2.x)."""
    def hz_to_mel(f):
        return 1127.0 * np.log1p(np.asarray(f, np.float64) / 700.0)
    nyquist = float(sample_rate) / 2.0
    lin = np.linspace(0.0, nyquist, int(num_spectrogram_bins))[1:]
    spec_mel = hz_to_mel(lin)[:, None]
    edges = np.linspace(hz_to_mel(lower_edge_hertz), hz_to_mel(upper_edge_hertz), int(num_mel_bins) + 2)
    lower, center, upper = edges[:-2][None, :], edges[1:-1][None, :], edges[2:][None, :]
    lo = (spec_mel - lower) / (center - lower)
    up = (upper - spec_mel) / (upper - center)
    w = np.maximum(0.0, np.minimum(lo, up))
    w = np.pad(w, [[1, 0], [0, 0]])
    return Tensor(w.astype(_npdt(dtype)))


# ---------------------------------------------------------------------------------------------------------------------
# tf.random (only ever used by the reference to build shapes: `_build()` feeds random input through the model)
# ---------------------------------------------------------------------------------------------------------------------
_RNG = [np.random.default_rng(0)]


def set_seed(seed):
    _RNG[0] = np.random.default_rng(seed)


def random_uniform(shape, minval=0, maxval=None, dtype=float32, seed=None, name=None):
    dt = _npdt(dtype)
    if dt.kind in "iu":
        return Tensor(_RNG[0].integers(minval, maxval, _shape_arg(shape)).astype(dt))
    return Tensor(_RNG[0].uniform(minval, 1.0 if maxval is None else maxval, _shape_arg(shape)).astype(dt))


def random_normal(shape, mean=0.0, stddev=1.0, dtype=float32, seed=None, name=None):
    return Tensor(_RNG[0].normal(mean, stddev, _shape_arg(shape)).astype(_npdt(dtype)))


E = math.e
