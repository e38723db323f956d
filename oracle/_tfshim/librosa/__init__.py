"""librosa stand-in (TEST INFRASTRUCTURE): the three functions asr/models/layers/backend.py calls.  librosa is absent from
this image, so what is restated here is librosa's published algorithm (filters.py, convert.py of librosa 0.8 - 0.10), not
the reference's code: the mel matrix stays a *weight input* of the product and the fixtures carry the
matrix this stand-in produced, labelled with this module's version string."""
import numpy as np

from . import filters, util   # noqa: F401

__version__ = "0.8-numpy-standin"


def fft_frequencies(sr=22050, n_fft=2048):
    return np.fft.rfftfreq(n=n_fft, d=1.0 / sr)


def load(path, sr=22050, mono=True, **_):
    """librosa.load for what the reference's fixture wavs are: integer PCM RIFF files already at the requested rate (librosa
    hands such a file to soundfile, which returns int / 2^(bits - 1) as float32; no resampling happens when the rates agree).
    Anything else raises: the stand-in does not restate librosa's resampler."""
    import wave
    with wave.open(str(path), "rb") as f:
        fs, nch, sw, n = f.getframerate(), f.getnchannels(), f.getsampwidth(), f.getnframes()
        raw = f.readframes(n)
    if sr is not None and fs != sr:
        raise NotImplementedError("librosa stand-in: %s is at %d Hz, asked for %d (resampling is not restated)" % (path, fs, sr))
    if sw == 2:
        x = np.frombuffer(raw, "<i2").astype(np.float32) / np.float32(32768.0)
    elif sw == 4:
        x = (np.frombuffer(raw, "<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
    elif sw == 1:
        x = (np.frombuffer(raw, np.uint8).astype(np.float32) - np.float32(128.0)) / np.float32(128.0)
    else:
        raise NotImplementedError("librosa stand-in: %d-byte PCM" % sw)
    x = x.reshape(-1, nch)
    x = x.mean(axis=1, dtype=np.float32) if (mono and nch > 1) else x[:, 0] if nch == 1 else x.T
    return np.ascontiguousarray(x, np.float32), fs
