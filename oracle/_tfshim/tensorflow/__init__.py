"""A NumPy-backed stand-in for the slice of the TensorFlow 2.8 API that the reference's model code calls, so that
/root/reference's own `asr/models/*.py`, `asr/models/layers/*.py`, `leaf_audio/*.py` run UNMODIFIED in a container without
TensorFlow and produce the golden vectors under tests/golden/tf_*.npz (tests/golden/make_tf_goldens.py).

TEST INFRASTRUCTURE ONLY (oracle/ rules): never imported by the product, never on sys.path unless a test or the golden
recipe puts it there.  See oracle/_tfshim/README.md for what is restated, how the stand-in itself is gated
(tests/test_tfshim.py: torch.nn.functional twins for every primitive, and the reference's CTCDecoder class executed here
against the reference's exported ctc_model.onnx), and what that does and does not prove."""
from . import _core as _c
from ._core import (DType, Tensor, TensorShape, TensorSpec, Variable, GradientTape, name_scope, newaxis, convert_to_tensor, constant,
                    zeros, ones, fill, zeros_like, ones_like, shape, size, rank, reshape, transpose, expand_dims, squeeze, concat,
                    stack, unstack, split, pad, tile, repeat, roll, reverse, gather, dynamic_stitch, cast, identity, stop_gradient,
                    sequence_mask, sqrt, exp, sin, cos, tanh, square, sign, floor, round_ as round, negative, add, subtract,
                    multiply, maximum, minimum, divide, less, less_equal, greater, greater_equal, equal, not_equal, logical_and,
                    logical_or, logical_not, sigmoid, reduce_sum, reduce_max, reduce_min, reduce_mean, reduce_prod, reduce_any,
                    reduce_all, argmax, argsort, where, clip_by_value, einsum, tensordot, matmul, while_loop, scan, cond, function,
                    numpy_function, float16, float32, float64, complex64, complex128, int8, uint8, int16, int32, int64, string,
                    as_dtype, set_wide, is_wide)
from ._core import range_ as range, abs_ as abs, pow_ as pow, print_ as print, bool_ as bool, complex_ as complex   # noqa: A001
from . import keras, linalg, math, nn, random, signal, train, io, losses, dtypes   # noqa: F401,E402

__version__ = "2.8-numpy-standin"
truediv = divide
