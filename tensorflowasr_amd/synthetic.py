"""Deterministic synthetic utterances for benchmarking (no datasets in this environment): white noise plus
three gated sinusoids, peak-normalised the way the reference's batch loader does
(asr/dataloaders/am_dataloader.py:151), float32 at 16 kHz."""
import numpy as np


def synth_wave(utt_index, length=160000, sr=16000):
    rng = np.random.default_rng(1234 + utt_index)
    t = np.arange(length) / sr
    x = 0.1 * rng.standard_normal(length)
    env = ((t // 0.3).astype(np.int64) % 2 == 0).astype(np.float64)
    for _ in range(3):
        f = rng.uniform(100, 3000)
        a = rng.uniform(0.05, 0.3)
        x += a * np.sin(2 * np.pi * f * t) * env
    x = x / np.max(np.abs(x))
    return x.astype(np.float32)


def synth_batch(first_index, batch, length=160000, sr=16000):
    return np.stack([synth_wave(first_index + i, length, sr) for i in range(batch)])
