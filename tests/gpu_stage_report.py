"""Stage-by-stage GPU-vs-oracle report (runs every stage even if an earlier one is off, so one gpurun call
localises every mismatch).  Writes gpurun_out/stage_report.json.  Test infrastructure, not product."""
import json
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402
from helpers import co, encoder_kwargs, golden_ctc_io, golden_ctc_weights, maxdiff, small_cfg, waves  # noqa: E402
from tensorflowasr_amd.models import ConformerCTC, ConformerEncoder, CTCDecoder, ctc_greedy_decode  # noqa: E402

report = {}


def stage(name):
    def deco(fn):
        t0 = time.time()
        try:
            report[name] = fn()
        except Exception as e:  # noqa: BLE001
            report[name] = {"error": repr(e), "trace": traceback.format_exc()[-1500:]}
        report[name]["sec"] = round(time.time() - t0, 2)
        print(name, json.dumps(report[name])[:600], flush=True)
        return fn
    return deco


def stats(gpu, ref):
    gpu = np.asarray(gpu, np.float64)
    ref = np.asarray(ref, np.float64)
    d = np.abs(gpu - ref)
    return {"max_abs": float(d.max()), "mean_abs": float(d.mean()), "ref_absmax": float(np.abs(ref).max()),
            "nan": int(np.isnan(gpu).sum()), "argmax_pos": [int(i) for i in np.unravel_index(d.argmax(), d.shape)]}


cfg2 = small_cfg(2)
w_enc = co.encoder_weights(cfg2, seed=0)
enc = ConformerEncoder(**encoder_kwargs(cfg2))
enc.load_weights(w_enc, by_name=False)
wav = waves(2, 32000)


@stage("mel_L32000")
def _():
    ref = co.melspectrogram(wav.astype(np.float64), w_enc)
    return stats(enc.melspectrogram(wav).cpu().numpy(), ref)


@stage("mel_L67263_odd_pad")
def _():
    x = waves(1, 67263, 5) * 0.05
    ref = co.melspectrogram(x.astype(np.float64), w_enc)
    return stats(enc.melspectrogram(x).cpu().numpy(), ref)


@stage("conv_subsampling")
def _():
    mel = co.melspectrogram(wav.astype(np.float64), w_enc)
    ref = co.conv_subsampling(mel, w_enc)
    return stats(enc.conv_subsampling(mel.astype(np.float32)).cpu().numpy(), ref)


@stage("conv_subsampling_F50_odd")
def _():
    rng = np.random.default_rng(3)
    mel = -80 * rng.random((3, 50, 80))
    ref = co.conv_subsampling(mel, w_enc)
    return stats(enc.conv_subsampling(mel.astype(np.float32)).cpu().numpy(), ref)


@stage("encoder_block0_T50")
def _():
    rng = np.random.default_rng(4)
    x = rng.standard_normal((2, 50, 144))
    ref = co.conformer_block(x, w_enc, "conformer_block_0", 36)
    return stats(enc.conformer_block(0, x.astype(np.float32)).cpu().numpy(), ref)


@stage("encoder_block1_T250")
def _():
    rng = np.random.default_rng(5)
    x = rng.standard_normal((3, 250, 144))
    ref = co.conformer_block(x, w_enc, "conformer_block_1", 36)
    return stats(enc.conformer_block(1, x.astype(np.float32)).cpu().numpy(), ref)


@stage("encoder_block_T300_two_keyblocks")
def _():
    rng = np.random.default_rng(6)
    x = rng.standard_normal((1, 300, 144))
    ref = co.conformer_block(x, w_enc, "conformer_block_1", 36)
    return stats(enc.conformer_block(1, x.astype(np.float32)).cpu().numpy(), ref)


@stage("sub_stages_block0")
def _():
    # localise inside a block: run the oracle module by module against GPU outputs obtained by zeroing weights is
    # overkill; instead compare FF1-only by using a block whose other modules are identity-free is not possible.
    # So: report per-module oracle norms to help reading block errors.
    rng = np.random.default_rng(4)
    x = rng.standard_normal((2, 50, 144))
    p = "conformer_block_0"
    a = co.ff_module(x, w_enc, p + "/ff_module_1")
    b = co.mhsa_module(a, w_enc, p + "/mhsa_module", 36)
    c = co.conv_module(b, w_enc, p + "/conv_module")
    d = co.ff_module(c, w_enc, p + "/ff_module_2")
    return {"ff1_delta": float(np.abs(a - x).max()), "mhsa_delta": float(np.abs(b - a).max()),
            "conv_delta": float(np.abs(c - b).max()), "ff2_delta": float(np.abs(d - c).max())}


@stage("encoder_full_2blocks")
def _():
    ref, inter = co.conformer_encoder(wav.astype(np.float64), w_enc, cfg2, return_intermediates=True)
    return stats(enc(wav).cpu().numpy(), ref)


@stage("ctc_decoder_golden_reference_graph")
def _():
    io = golden_ctc_io()
    dec = CTCDecoder(1332, dmodel=144, num_blocks=1, head_size=36, num_heads=4, kernel_size=32)
    dec.load_weights(golden_ctc_weights(), by_name=False)
    la, aa = dec(io["x_a"], return_argmax=True)
    la, aa = la.cpu().numpy(), aa.cpu().numpy()
    lb, ab = dec(io["x_b"], return_argmax=True)
    r = stats(la, io["logits_a"])
    r["argmax_equal_a"] = bool((la.argmax(-1) == io["logits_a"].argmax(-1)).all())
    r["kernel_argmax_equals_logits_argmax"] = bool((aa == la.argmax(-1)).all())
    r["argmax_equal_b"] = bool((ab.cpu().numpy() == io["argmax_b"]).all())
    r["max_b_diff"] = maxdiff(lb.cpu().numpy().max(-1), io["max_b"])
    return r


@stage("greedy_kats")
def _():
    kats = json.load(open(os.path.join(ROOT, "tests", "golden", "greedy_kat.json")))
    ok = 0
    for k in kats:
        fa = np.array(k["probs"], np.float32).argmax(-1) if "probs" in k else np.array(k["frame_argmax"])
        ids, n = ctc_greedy_decode(fa[None].astype(np.int32), None, k["blank"], device="cuda:0")
        got = ids.cpu().numpy()[0, :int(n.cpu().numpy()[0])].tolist()
        ok += got == k["expect"]
    return {"ok": ok, "total": len(kats)}


@stage("recognize_S_full_13blocks_B2")
def _():
    cfg = dict(co.CONFORMER_S)
    w = co.encoder_weights(cfg, seed=0)
    w.update(golden_ctc_weights())
    m = ConformerCTC(1332, **{k: v for k, v in encoder_kwargs(cfg).items() if k != "mel_layer_type"})
    m.load_weights(w, by_name=False)
    x = waves(2, 48000)
    ids, lens = m.recognize(x)
    ids, lens = ids.cpu().numpy(), lens.cpu().numpy()
    enc_ref = co.conformer_encoder(x.astype(np.float64), w, cfg)
    logits_ref = co.ctc_decoder(enc_ref, w, cfg)
    rid, rlen = co.ctc_greedy(logits_ref, [logits_ref.shape[1]] * 2, 1331)
    enc_gpu = m.encode(x).cpu().numpy()
    r = stats(enc_gpu, enc_ref)
    lg = m.ctc_logits(enc_gpu).cpu().numpy()
    r["logits_max_abs"] = maxdiff(lg, logits_ref)
    r["ids_equal"] = bool((ids == rid).all() and (lens == rlen).all())
    r["lens"] = [int(v) for v in lens]
    r["ref_lens"] = [int(v) for v in rlen]
    return r


@stage("streaming_d256_k5")
def _():
    cfg = small_cfg(2, co.STREAMING_S)
    w = co.encoder_weights(cfg, seed=2)
    e = ConformerEncoder(**encoder_kwargs(cfg, chunk_size=8000))
    e.load_weights(w, by_name=False)
    x = waves(2, 24000, 9)
    ref = co.streaming_conformer_encoder(x.astype(np.float64), w, cfg, 8000)
    return stats(e(x).cpu().numpy(), ref)


@stage("timing_B64")
def _():
    cfg = dict(co.CONFORMER_S)
    w = co.encoder_weights(cfg, seed=0)
    w.update(golden_ctc_weights())
    m = ConformerCTC(1332, **{k: v for k, v in encoder_kwargs(cfg).items() if k != "mel_layer_type"})
    m.load_weights(w, by_name=False)
    x = torch.from_numpy(waves(8, 160000)).cuda().repeat(8, 1)
    m.recognize(x)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3):
        ids, lens = m.recognize(x)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 3
    return {"ms_per_batch64": dt * 1e3, "frames_per_s": 64 * 1000 / dt, "lens": [int(v) for v in lens[:8].cpu()]}


os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "stage_report.json"), "w") as f:
    json.dump(report, f, indent=1)
print("WROTE stage_report.json")
