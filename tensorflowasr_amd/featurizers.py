"""I/O surface of the reference kept as is: `SpeechFeaturizer` (utils/speech_featurizers.py:55-77) and
`TextFeaturizer` (utils/text_featurizers.py:7-99), without librosa / soundfile / tensorflow."""
import codecs
import io
import os
import wave

import numpy as np


def _pcm_to_float(raw, sampwidth, nch):
    if sampwidth == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif sampwidth == 4:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    elif sampwidth == 1:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    elif sampwidth == 3:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v & 0x800000, v - 0x1000000, v)
        x = v.astype(np.float32) / 8388608.0
    else:
        raise ValueError("unsupported PCM sample width %d" % sampwidth)
    if nch > 1:
        x = x.reshape(-1, nch).mean(axis=1)          # librosa.load(mono=True) averages channels
    return x.astype(np.float32)


def _resample(x, sr_in, sr_out):
    """Band-limited polyphase resampling (librosa.load resamples when the file rate differs)."""
    from math import gcd
    from scipy.signal import resample_poly
    g = gcd(int(sr_in), int(sr_out))
    return resample_poly(x, sr_out // g, sr_in // g).astype(np.float32)


def read_raw_audio(audio, sample_rate=16000):
    """utils/speech_featurizers.py:10-22: path | bytes | ndarray -> float32 mono in [-1, 1) at `sample_rate`."""
    if isinstance(audio, np.ndarray):
        return audio
    if isinstance(audio, (str, os.PathLike)):
        f = wave.open(os.path.expanduser(str(audio)), "rb")
    elif isinstance(audio, bytes):
        f = wave.open(io.BytesIO(audio), "rb")
    else:
        raise ValueError("input audio must be either a path or bytes")
    with f:
        sr, nch, sw = f.getframerate(), f.getnchannels(), f.getsampwidth()
        x = _pcm_to_float(f.readframes(f.getnframes()), sw, nch)
    if sr != sample_rate:
        x = _resample(x, sr, sample_rate)
    return x


def normalize_signal(signal):
    """utils/speech_featurizers.py:34-37."""
    return signal * (1.0 / (np.max(np.abs(signal)) + 1e-9))


class SpeechFeaturizer:
    def __init__(self, speech_config: dict):
        self.sample_rate = speech_config["sample_rate"]
        try:
            self.frame_length = int(self.sample_rate * (speech_config["frame_ms"] / 1000))
            self.frame_step = int(self.sample_rate * (speech_config["stride_ms"] / 1000))
            self.num_feature_bins = speech_config["num_feature_bins"]
        except Exception:
            pass

    def load_wav(self, path):
        return read_raw_audio(path, self.sample_rate)

    def compute_time_dim(self, seconds: float) -> int:
        total_frames = seconds * self.sample_rate + 2 * (self.frame_length // 2)
        return int(1 + (total_frames - self.frame_length) // self.frame_step)

    def pad_signal(self, wavs, max_length):
        """keras pad_sequences(wavs, max_length, 'float32', 'post', 'post') (speech_featurizers.py:75-77)."""
        max_length = int(max_length)
        out = np.zeros((len(wavs), max_length), dtype=np.float32)
        for i, w in enumerate(wavs):
            w = np.asarray(w, dtype=np.float32)[:max_length]     # truncating='post'
            out[i, :len(w)] = w
        return out


class TextFeaturizer:
    """Vocabulary file -> token <-> index tables; the blank is appended LAST unless `blank_at_zero`
    (utils/text_featurizers.py:42-70)."""

    def __init__(self, decoder_config: dict, show=False):
        self.decoder_config = decoder_config
        path = os.path.abspath(os.path.expanduser(self.decoder_config["vocabulary"]))
        self.decoder_config["vocabulary"] = path
        self.scorer = None
        with codecs.open(path, "r", "utf-8") as fin:
            lines = fin.readlines()
        if show:
            print("load token at {}".format(path))
        self.token_to_index, self.index_to_token, self.vocab_array = {}, {}, []
        index = 0
        if self.decoder_config["blank_at_zero"]:
            self.blank = 0
            index = 1
        for line in lines:
            line = line.strip()
            if line.startswith("#") or not line or line == "\n":
                continue
            if line == "[SPACE]":
                line = " "
            self.token_to_index[line] = index
            self.index_to_token[index] = line
            self.vocab_array.append(line)
            index += 1
        self.num_classes = index
        if not self.decoder_config["blank_at_zero"]:
            self.blank = index
            self.num_classes += 1
        self.pad = 0
        self.stop = -1

    def startid(self):
        return self.token_to_index["<S>"]

    def endid(self):
        return self.token_to_index["</S>"]

    def extract(self, tokens):
        return [self.token_to_index[t] for t in tokens]

    def iextract(self, feat):
        if isinstance(feat, list):
            return [self.index_to_token[i] for i in feat]
        return self.index_to_token[feat]
