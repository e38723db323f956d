def mse(y_true, y_pred):
    from ._core import reduce_mean, square
    return reduce_mean(square(y_pred - y_true), axis=-1)
