"""librosa stand-in (TEST INFRASTRUCTURE): the three functions asr/models/layers/backend.py calls.  librosa is absent from
this image, so what is restated here is librosa's published algorithm (filters.py, convert.py of librosa 0.8 - 0.10), not
the reference's code: the mel matrix stays a *weight input* of the product and the fixtures carry the
matrix this stand-in produced, labelled with this module's version string."""
import numpy as np

from . import filters, util   # noqa: F401

__version__ = "0.8-numpy-standin"


def fft_frequencies(sr=22050, n_fft=2048):
    return np.fft.rfftfreq(n=n_fft, d=1.0 / sr)
