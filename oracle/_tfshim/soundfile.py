"""soundfile stand-in (TEST INFRASTRUCTURE): utils/speech_featurizers.py imports the module at the top; its `read` is only
reached for in-memory bytes (speech_featurizers.py:14-18).  PCM RIFF only, through the standard library."""
import io
import wave

import numpy as np


def read(file, dtype="float64"):
    f = wave.open(io.BytesIO(file) if isinstance(file, (bytes, bytearray)) else file, "rb")
    with f:
        fs, nch, sw = f.getframerate(), f.getnchannels(), f.getsampwidth()
        raw = f.readframes(f.getnframes())
    if sw != 2:
        raise NotImplementedError("soundfile stand-in: 16-bit PCM only")
    x = (np.frombuffer(raw, "<i2").astype(np.float64) / 32768.0).astype(dtype).reshape(-1, nch)
    return (x[:, 0] if nch == 1 else x), fs
