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


def leaf_default_weights(n_filters=80, sample_rate=16000, prefix="mel_layer"):
    """Initial values of the LEAF frontend as ConformerEncoder constructs it (conformer_blocks.py:316): PreempInit(0.97),
    GaborInit(sample_rate, min_freq = 30 * (sr // 8000), max_freq = 3900 * (sr // 8000)) -- centre frequency and width
    of each Gabor filter read off a 512-point HTK mel bank (leaf_audio/melfilters.py:58-92) -- Gaussian pooling sigma
    0.4, PCEN alpha 0.96 / smoothing 0.04 / delta 2 / root 2, instance-norm gamma 1 / beta 0 (frontend.py:106-160)."""
    n_fft = 512
    lo, hi = 30.0 * (sample_rate // 8000), 3900.0 * (sample_rate // 8000)
    nbins = n_fft // 2 + 1
    to_mel = lambda f: 1127.0 * np.log1p(np.asarray(f, np.float64) / 700.0)
    spec = to_mel(np.linspace(0.0, sample_rate / 2.0, nbins)[1:])[:, None]       # tf.signal skips the DC bin ...
    edges = np.linspace(to_mel(lo), to_mel(hi), n_filters + 2)
    lower, center, upper = edges[None, :-2], edges[None, 1:-1], edges[None, 2:]
    bank = np.maximum(0.0, np.minimum((spec - lower) / (center - lower), (upper - spec) / (upper - center)))
    bank = np.sqrt(np.pad(bank, ((1, 0), (0, 0))).T)                              # ... and pads it back as zeros
    centers = bank.argmax(1).astype(np.float64)
    fwhm = (bank >= bank.max(1, keepdims=True) / 2.0).sum(1).astype(np.float64)
    kern = np.stack([centers * 2 * np.pi / n_fft, np.sqrt(2.0 * np.log(2.0)) * n_fft / (np.pi * fwhm)], 1)
    one = np.ones(n_filters, np.float32)
    return {prefix + "/tfbanks_preemp/kernel": np.array([-0.97, 1.0], np.float32).reshape(2, 1, 1),
            prefix + "/tfbanks_complex_conv/kernel": kern.astype(np.float32),
            prefix + "/learnable_pooling/kernel": np.full((1, 1, n_filters, 1), 0.4, np.float32),
            prefix + "/PCEN/alpha": 0.96 * one, prefix + "/PCEN/delta": 2.0 * one, prefix + "/PCEN/root": 2.0 * one,
            prefix + "/PCEN/EMA/smooth": 0.04 * one,
            prefix + "/tfbanks_instancenorm/gamma": one.copy(), prefix + "/tfbanks_instancenorm/beta": 0.0 * one}
