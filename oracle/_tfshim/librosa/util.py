import numpy as np


def normalize(S, norm=np.inf, axis=0, threshold=None, fill=None):
    S = np.asarray(S)
    mag = np.abs(S).astype(float)
    if norm == np.inf:
        length = mag.max(axis=axis, keepdims=True)
    elif norm == 1:
        length = mag.sum(axis=axis, keepdims=True)
    elif norm == 2:
        length = np.sqrt((mag ** 2).sum(axis=axis, keepdims=True))
    else:
        raise ValueError(norm)
    tiny = np.finfo(np.float32).tiny if threshold is None else threshold
    length = np.where(length < tiny, 1.0, length)
    return S / length
