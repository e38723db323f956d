from ._core import (log, exp, sqrt, rsqrt, sin, cos, tanh, abs_ as abs, square, real, imag, conj, ceil, floor, minimum, maximum,   # noqa: F401,A001
                    pow_ as pow, is_nan, is_inf, reduce_sum, reduce_max, reduce_min, reduce_mean, reduce_prod, argmax, sigmoid,
                    add, subtract, multiply, divide, less, greater, equal, not_equal, logical_and, logical_or, logical_not,
                    negative, sign, round_ as round, log_softmax, softmax, floordiv)
