"""Timing of the BASELINE.json configurations that are NOT the headline bench line (bench.py measures config 2 / 4):

  config 3  StreamingConformerCTC (d=256, 4 blocks, k=5; CTCDecoder 1 block k=32), batch = 64 streaming chunks of
            0.5 s: one encoder step over the 64 new chunks + the "global CTC" over 10 s of history per stream.
  config 5  ChunkConformer + CTC prefix beam (externals/ctc_decoders), 16 x 30 s utterances per GPU
            (= 128 over 8 GPUs), beam 10 and 100, cutoff_prob 0.99, cutoff_top_n 40.

    python tests/bench_configs.py [--steps K]      -> one JSON line per configuration (synthetic data, random init)

Kept out of bench.py so that the driver's contract line stays the headline metric on the headline config."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tensorflowasr_amd.models import (ChunkConformer, CTCDecoder, StreamingConformerEncoder, ctc_greedy_decode,  # noqa: E402
                                      ctc_prefix_beam_decode)
from tensorflowasr_amd.synthetic import synth_batch  # noqa: E402


def timed(fn, steps, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def config3(steps, gemm_dtype):
    B, chunk, hist = 64, 8000, 20
    enc = StreamingConformerEncoder(dmodel=256, reduction_factor=4, num_blocks=4, head_size=64, num_heads=4, kernel_size=5,
                                    fc_factor=0.5, sample_rate=16000, n_mels=80, stride_ms=10,
                                    mel_layer_type="Melspectrogram", gemm_dtype=gemm_dtype)
    enc.add_chunk_size(chunk, 80, 640)
    enc._build(seed=0)
    ctc = CTCDecoder(num_classes=1332, dmodel=256, num_blocks=1, head_size=64, num_heads=4, kernel_size=32, fc_factor=0.5,
                     gemm_dtype=gemm_dtype)
    ctc._build(seed=1)
    wav = torch.from_numpy(synth_batch(0, B, chunk)).cuda()
    history = torch.randn(B, hist * 13, 256, device="cuda")

    def enc_step():
        return enc(wav)

    def full_step():
        e = enc(wav)
        h = torch.cat([history[:, 13:], e], 1)
        _, amax = ctc(h, return_argmax=True, return_logits=False)
        return ctc_greedy_decode(amax, None, blank=1331)

    te, tf = timed(enc_step, steps), timed(full_step, steps)
    return {"config": "StreamingConformerCTC 15M, batch=64 streaming chunks (0.5 s each), global CTC over 10 s history",
            "dtype": "bf16 GEMM inputs, f32 accumulate" if gemm_dtype == "bfloat16" else "f32",
            "ms_encoder_step": round(te * 1e3, 3), "ms_step_with_global_ctc": round(tf * 1e3, 3),
            "chunks_per_s": round(B / tf, 1), "audio_frames_per_s": round(B * 50 / tf, 1),
            "rtf_per_stream": round(tf / 0.5, 6)}


def config5(steps):
    from oracle import conformer_oracle as co     # configuration dictionary / random weights only (not on the timed path)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import chunk_config_dict
    cfg = dict(co.CHUNK_S)
    w = co.chunk_weights(cfg, seed=0, picker_blank_bias=0.0)
    m = ChunkConformer(chunk_config_dict(cfg), phone=cfg["picker_num_classes"], txt=cfg["decoder_num_classes"])
    m.load_weights(w, by_name=False)
    B, L = 16, 480000
    wav = torch.from_numpy(synth_batch(0, B, L)).cuda()
    out = {}

    def predict():
        out["logits"], out["counts"] = m.predict(wav)

    tp = timed(predict, steps)
    logits, counts = out["logits"], out["counts"]
    res = {"config": "ChunkConformer 15M + CTC prefix beam, 16 x 30 s utterances per GPU (B=128 over 8 GPUs)",
           "dtype": "f32", "ms_predict": round(tp * 1e3, 2), "picked_frames_max": int(logits.shape[1]),
           "audio_frames_per_s_predict": round(B * 3000 / tp, 1)}
    for beam in (10, 100):
        def decode():
            return ctc_prefix_beam_decode(logits, counts, beam_width=beam, cutoff_prob=0.99, cutoff_top_n=40, is_logits=True)
        tb = timed(decode, max(1, steps // 2), warmup=1)
        res["ms_beam%d" % beam] = round(tb * 1e3, 2)
        res["audio_frames_per_s_beam%d" % beam] = round(B * 3000 / (tp + tb), 1)
    return res


def frontend_variants(steps, names=None):
    """the headline shape (ConformerCTC(S), 64 x 10 s, fp32) with the reference's other frontend options, and the
    larger model sizes"""
    from tensorflowasr_amd.models import ConformerCTC
    B, L = 64, 160000
    wav = torch.from_numpy(synth_batch(0, B, L)).cuda()
    res = {"config": "ConformerCTC 64 x 10 s, recognize() ms/step by frontend option / model size", "dtype": "f32"}
    variants = [("S_mel", {}), ("S_mel_wavinfo", dict(add_wav_info=True)), ("S_leaf", dict(mel_layer_type="leaf")),
                ("S_leaf_wavinfo", dict(mel_layer_type="leaf", add_wav_info=True)),
                ("M_mel", dict(dmodel=256, num_blocks=13, head_size=64, num_heads=4)),
                ("L_mel", dict(dmodel=512, num_blocks=13, head_size=64, num_heads=8))]
    for name, kw in variants:
        if names and name not in names:
            continue
        m = ConformerCTC(1332, **kw)
        m._build()
        m.prepare(B, L)
        t = timed(lambda: m.recognize(wav), steps)
        res[name] = {"ms_step": round(t * 1e3, 3), "audio_frames_per_s": round(B * 1000 / t, 1)}
        del m
        torch.cuda.empty_cache()
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--only", type=int, default=0)
    ap.add_argument("--c3-dtype", default="both", choices=["both", "bf16", "f32"])
    ap.add_argument("--variants", default="", help="comma-separated subset of the --only 2 variants (S_mel, M_mel, L_mel, ...)")
    a = ap.parse_args()
    if a.only in (0, 3):
        if a.c3_dtype in ("both", "bf16"):
            print(json.dumps(config3(a.steps, "bfloat16")), flush=True)
        if a.c3_dtype in ("both", "f32"):
            print(json.dumps(config3(a.steps, "float32")), flush=True)
    if a.only in (0, 5):
        print(json.dumps(config5(a.steps)), flush=True)
    if a.only in (0, 2):
        print(json.dumps(frontend_variants(a.steps, [v for v in a.variants.split(',') if v])), flush=True)
