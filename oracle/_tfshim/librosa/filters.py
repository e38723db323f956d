import numpy as np
import scipy.signal

from . import util


def get_window(window, Nx, fftbins=True):
    # librosa.filters.get_window: strings go to scipy.signal.get_window
    return scipy.signal.get_window(window, Nx, fftbins=fftbins)


def hz_to_mel(frequencies, htk=False):
    f = np.asanyarray(frequencies, dtype=float)
    if htk:
        return 2595.0 * np.log10(1.0 + f / 700.0)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def mel_to_hz(mels, htk=False):
    m = np.asanyarray(mels, dtype=float)
    if htk:
        return 700.0 * (10.0 ** (m / 2595.0) - 1.0)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_frequencies(n_mels=128, fmin=0.0, fmax=11025.0, htk=False):
    return mel_to_hz(np.linspace(hz_to_mel(fmin, htk), hz_to_mel(fmax, htk), n_mels), htk)


def mel(sr, n_fft, n_mels=128, fmin=0.0, fmax=None, htk=False, norm="slaney", dtype=np.float32):
    """librosa.filters.mel (0.8+): triangles on the Slaney (or HTK) mel scale, linear in Hz; norm='slaney' scales each filter
    by 2 / (f[i+2] - f[i]); a numeric norm goes to util.normalize(weights, norm=norm, axis=-1) -- the reference passes norm=1
    (asr/models/layers/backend.py:13-24), i.e. every filter is scaled to unit L1 norm.  (librosa <= 0.7 read norm=1 as the
    Slaney area normalisation; tests/test_oracle.py keeps that variant switchable on the oracle's side.)"""
    if fmax is None:
        fmax = float(sr) / 2
    n_mels = int(n_mels)
    weights = np.zeros((n_mels, int(1 + n_fft // 2)), dtype=dtype)
    fftfreqs = np.fft.rfftfreq(n=n_fft, d=1.0 / sr)
    mel_f = mel_frequencies(n_mels + 2, fmin=fmin, fmax=fmax, htk=htk)
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    if norm == "slaney":
        enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
        weights *= enorm[:, np.newaxis]
    elif norm is not None:
        weights = util.normalize(weights, norm=norm, axis=-1)
    return weights
