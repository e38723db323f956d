from ._core import rfft, irfft, fft, fftshift, frame, overlap_and_add, stft, hann_window, linear_to_mel_weight_matrix   # noqa: F401
