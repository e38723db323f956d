"""Shared helpers for the parity tests (test infrastructure: may import oracle/)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import conformer_oracle as co  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def golden_ctc_weights():
    return dict(np.load(os.path.join(GOLDEN, "ctc_decoder_weights.npz")))


def golden_ctc_io():
    return np.load(os.path.join(GOLDEN, "ctc_decoder_io.npz"))


def small_cfg(num_blocks=2, base=None):
    cfg = dict(base or co.CONFORMER_S)
    cfg["num_blocks"] = num_blocks
    return cfg


def waves(n, length, start=0):
    return np.stack([co.synth_wave(start + i, length) for i in range(n)])


def encoder_kwargs(cfg, chunk_size=0):
    return dict(dmodel=cfg["dmodel"], reduction_factor=cfg["reduction_factor"], num_blocks=cfg["num_blocks"],
                head_size=cfg["head_size"], num_heads=cfg["num_heads"], kernel_size=cfg["kernel_size"],
                fc_factor=cfg["fc_factor"], sample_rate=cfg["sample_rate"], n_mels=cfg["n_mels"],
                stride_ms=cfg["stride_ms"], mel_layer_type="Melspectrogram", chunk_size=chunk_size)


def maxdiff(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max())


def argmax_mismatch_report(gpu_logits, ref_logits):
    """frames where argmax differs, with the reference's own top-2 margin there."""
    ga, ra = gpu_logits.argmax(-1), ref_logits.argmax(-1)
    bad = np.argwhere(ga != ra)
    out = []
    for idx in bad:
        row = np.sort(ref_logits[tuple(idx)])[::-1]
        out.append((tuple(int(i) for i in idx), float(row[0] - row[1])))
    return out


def chunk_config_dict(cfg):
    """oracle-style flat chunk config -> the reference's nested model YAML (asr/configs/chunk_conformerS.yml)."""
    common = dict(dmodel=cfg["dmodel"], head_size=cfg["head_size"], num_heads=cfg["num_heads"],
                  kernel_size=cfg["kernel_size"], fc_factor=cfg["fc_factor"], dropout=0.0)
    return {"model_config": {
        "name": "ChunkConformer",
        "ChunkConformerFront": dict(dmodel=cfg["dmodel"], reduction_factor=4, dropout=0.0, sample_rate=cfg["sample_rate"],
                                    n_mels=cfg["n_mels"], mel_layer_trainable=False, stride_ms=cfg["stride_ms"], chunk_num=16),
        "ChunkConformerEncoder": dict(common, num_blocks=cfg["enc_num_blocks"], win_front=cfg["enc_win_front"],
                                      win_back=cfg["enc_win_back"], padding="causal", name="chunk_conformer_encoder"),
        "ChunkCTCPicker": dict(common, num_classes=cfg["picker_num_classes"], num_blocks=cfg["picker_num_blocks"],
                               win_front=cfg["picker_win_front"], win_back=cfg["picker_win_back"]),
        "ChunkCTCDecoder": dict(common, num_classes=cfg["decoder_num_classes"], num_blocks=cfg["decoder_num_blocks"],
                                win_front=cfg["decoder_win_front"], win_back=cfg["decoder_win_back"]),
        "ContextHelper": dict(common, num_classes=cfg["picker_num_classes"], num_blocks=cfg["helper_num_blocks"],
                              win_front=cfg["helper_win_front"], win_back=cfg["helper_win_back"])}}
