"""Host-side constants of the Melspectrogram layer: the windowed DFT kernels and the mel filterbank the
reference builds at `Melspectrogram.build` time and then stores as (non-trainable) Keras weights
(asr/models/layers/time_frequency.py:51-60,160-164; backend.py:13-69).  They are ordinary weights of the
model here too: `_build()` fills them with these defaults and `load_weights` may overwrite them."""
import numpy as np


def stft_kernels(n_dft=1024):
    """backend.py:27-69: cos / -sin DFT bases times a periodic Hann window -> [n_dft,1,1,n_dft//2+1] x2."""
    nb = n_dft // 2 + 1
    t = np.arange(n_dft, dtype=np.float64)
    w = np.arange(nb, dtype=np.float64) * (2.0 * np.pi / n_dft)
    phase = np.outer(t, w)
    hann = (0.5 - 0.5 * np.cos(2.0 * np.pi * t / n_dft)).astype(np.float32).astype(np.float64)
    real = (np.cos(phase) * hann[:, None]).astype(np.float32)
    imag = (-np.sin(phase) * hann[:, None]).astype(np.float32)
    return real.reshape(n_dft, 1, 1, nb), imag.reshape(n_dft, 1, 1, nb)


def _slaney_hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    lin = f * 3.0 / 200.0
    log = 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) * (27.0 / np.log(6.4))
    return np.where(f >= 1000.0, log, lin)


def _slaney_mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    lin = m * 200.0 / 3.0
    log = 1000.0 * np.exp((m - 15.0) * (np.log(6.4) / 27.0))
    return np.where(m >= 15.0, log, lin)


def freq2mel(sr=16000, n_dft=1024, n_mels=80, fmin=0.0, fmax=None, norm=1):
    """backend.py:13-24 -> librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk=False, norm=1), transposed
    to the [n_freq, n_mels] layout the layer multiplies with (time_frequency.py:156-158).

    `norm=1` means per-filter L1 normalisation in librosa >= 0.8 and Slaney area normalisation in
    librosa <= 0.7 (pass norm='slaney' for the latter)."""
    fmax = sr / 2.0 if fmax is None else fmax
    nfreq = n_dft // 2 + 1
    freqs = np.linspace(0.0, sr / 2.0, nfreq)
    edges = _slaney_mel_to_hz(np.linspace(_slaney_hz_to_mel(fmin), _slaney_hz_to_mel(fmax), n_mels + 2))
    fb = np.zeros((n_mels, nfreq), dtype=np.float32)
    for i in range(n_mels):
        up = (freqs - edges[i]) / (edges[i + 1] - edges[i])
        down = (edges[i + 2] - freqs) / (edges[i + 2] - edges[i + 1])
        fb[i] = np.maximum(0.0, np.minimum(up, down))
    if norm == "slaney":
        fb *= (2.0 / (edges[2:] - edges[:-2]))[:, None].astype(np.float32)
    elif norm == 1:
        s = np.abs(fb).sum(axis=1, keepdims=True)
        fb = fb / np.where(s < np.finfo(np.float32).tiny, 1.0, s)
    elif norm is not None:
        raise ValueError("norm must be 1, 'slaney' or None")
    return np.ascontiguousarray(fb.T.astype(np.float32))
