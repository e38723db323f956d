"""Generates the committed golden fixtures.  Runs ONLY in the build container, where
/root/reference exists (it does not exist on the GPU box; nothing at test time reads it).

  python tests/golden/make_golden.py

Outputs (all under tests/golden/):
  ctc_decoder_weights.npz  trained CTCDecoder weights pulled out of the reference's exported
                           graph Inference/PythonInference/asr/models/offline/ctc_model.onnx,
                           renamed to the Keras layout/names used by oracle/conformer_oracle.py
  ctc_decoder_io.npz       inputs + the logits the reference graph itself produces for them
                           (executed by oracle/onnx_mini.py), used to pin the oracle
  greedy_kat.json          known-answer tests produced by the reference's C++
                           ctc_greedy_decoder.h (compiled by oracle/Makefile into oracle/_ref)
  beam_kat.npz             known-answer tests produced by the reference's scorer-less prefix beam search
                           (externals/ctc_decoders.zip compiled by oracle/Makefile target ref_beam)
"""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import onnx_mini  # noqa: E402

REF = "/root/reference"
ONNX = REF + "/Inference/PythonInference/asr/models/offline/ctc_model.onnx"
OUT = os.path.join(ROOT, "tests", "golden")
H, HS, D = 4, 36, 144


def extract_weights(inits):
    blk = "decoder_conformer_block_0"
    g = lambda n: np.array(inits[n], dtype=np.float32)
    w = {}
    w["project/kernel"] = g("dense_53/Tensordot/ReadVariableOp:0")
    w["project/bias"] = g("dense_53/BiasAdd/ReadVariableOp:0")

    def ln(dst, idx, sub):
        base = "%s/%slayer_normalization_%d" % (blk, sub, idx)
        w[dst + "/gamma"] = g(base + "/mul_3/ReadVariableOp:0")
        w[dst + "/beta"] = g(base + "/add/ReadVariableOp:0")

    ln(blk + "/ff_module_1/ln", 65, "ff_module_1/")
    ln(blk + "/mhsa_module/ln", 66, "mhsa_module/")
    ln(blk + "/conv_module/ln", 67, "conv_module/")
    ln(blk + "/ff_module_2/ln", 68, "ff_module_2/")
    ln(blk + "/ln", 69, "")
    for ff, a, b in (("ff_module_1", 54, 55), ("ff_module_2", 56, 57)):
        w["%s/%s/ffn1/kernel" % (blk, ff)] = g("%s/%s/dense_%d/Tensordot/ReadVariableOp:0" % (blk, ff, a))
        w["%s/%s/ffn1/bias" % (blk, ff)] = g("%s/%s/dense_%d/BiasAdd/ReadVariableOp:0" % (blk, ff, a))
        w["%s/%s/ffn2/kernel" % (blk, ff)] = g("%s/%s/dense_%d/Tensordot/ReadVariableOp:0" % (blk, ff, b))
        w["%s/%s/ffn2/bias" % (blk, ff)] = g("%s/%s/dense_%d/BiasAdd/ReadVariableOp:0" % (blk, ff, b))
    # MHA: tf2onnx folded einsum "BNI,HIO->BNHO" into Gemm(transB=1) with W[h*hs+o, i] = kernel[h,i,o].
    # Roles established by tracing the graph (the Gemm whose output is multiplied by truediv_recip
    # is the query; the one MatMul'ed against it is the key; the remaining one the value).
    m = blk + "/mhsa_module/mha"
    for nm, const in (("query_kernel", "const_fold_opt__9843"), ("key_kernel", "const_fold_opt__9837"),
                      ("value_kernel", "const_fold_opt__9842")):
        w[m + "/" + nm] = np.ascontiguousarray(g(const).reshape(H, HS, D).transpose(0, 2, 1))
    # projection: const [1,1,O,H,I] ; projection_kernel[h,i,o]
    w[m + "/projection_kernel"] = np.ascontiguousarray(g("const_fold_opt__9840")[0, 0].transpose(1, 2, 0))
    w[m + "/projection_bias"] = g(blk + "/mhsa_module/multi_head_attention_13/add/ReadVariableOp:0")
    c = blk + "/conv_module"
    w[c + "/pw_conv_1/kernel"] = np.ascontiguousarray(g(c + "/pw_conv_1/conv1d/ExpandDims_1:0")[:, :, 0, 0].T[None])
    w[c + "/pw_conv_1/bias"] = g("const_fold_opt__9530").reshape(-1)
    w[c + "/dw_conv/depthwise_kernel"] = np.ascontiguousarray(g("const_fold_opt__9512")[:, 0, 0, :].T[:, :, None])
    w[c + "/dw_conv/pointwise_kernel"] = np.ascontiguousarray(g(c + "/dw_conv/ExpandDims_2:0")[:, :, 0, 0].T[None])
    w[c + "/dw_conv/bias"] = g(c + "/dw_conv/BiasAdd/ReadVariableOp:0")
    # BatchNorm arrives folded (mul, shift). Express as Keras params with mean 0, var 1-eps so
    # gamma*(x-0)/sqrt(var+eps)+beta == scale*x+shift.
    scale = g(c + "/batch_normalization_13/batchnorm/mul:0").reshape(-1)
    shift = g("const_fold_opt__9544").reshape(-1)
    w[c + "/bn/gamma"] = scale
    w[c + "/bn/beta"] = shift
    w[c + "/bn/moving_mean"] = np.zeros_like(scale)
    w[c + "/bn/moving_variance"] = np.full_like(scale, 1.0 - 1e-3)
    w[c + "/pw_conv_2/kernel"] = np.ascontiguousarray(g(c + "/pw_conv_2/conv1d/ExpandDims_1:0")[:, :, 0, 0].T[None])
    w[c + "/pw_conv_2/bias"] = g("const_fold_opt__9516").reshape(-1)
    w["fully_connected/kernel"] = g("fully_connected/Tensordot/ReadVariableOp:0")
    w["fully_connected/bias"] = g("fully_connected/BiasAdd/ReadVariableOp:0")
    return w


def make_onnx_fixtures():
    nodes, inits, gin, gout = onnx_mini.load(ONNX)
    w = extract_weights(inits)
    np.savez_compressed(os.path.join(OUT, "ctc_decoder_weights.npz"), **w)
    rng = np.random.default_rng(0)
    xa = rng.standard_normal((2, 50, D)).astype(np.float32)
    xb = (2.0 * rng.standard_normal((1, 250, D))).astype(np.float32)
    ya = onnx_mini.run(nodes, inits, {gin[0]: xa}, gout)[0]
    yb = onnx_mini.run(nodes, inits, {gin[0]: xb}, gout)[0]
    zb = yb.astype(np.float64)
    lse = np.log(np.exp(zb - zb.max(-1, keepdims=True)).sum(-1)) + zb.max(-1)
    np.savez_compressed(os.path.join(OUT, "ctc_decoder_io.npz"),
                        x_a=xa, logits_a=ya.astype(np.float32),
                        x_b=xb, argmax_b=yb.argmax(-1).astype(np.int32),
                        max_b=yb.max(-1).astype(np.float32), lse_b=lse.astype(np.float32),
                        logits_b_every8=yb[:, ::8].astype(np.float32))
    print("onnx fixtures: logits_a", ya.shape, "mean %.3f std %.3f" % (ya.mean(), ya.std()),
          "| blank share b: %.2f" % (yb.argmax(-1) == 1331).mean())


def make_greedy_kats():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_ctc_greedy.so"))
    lib.ref_ctc_greedy.restype = ctypes.c_int
    lib.ref_ctc_greedy.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   ctypes.POINTER(ctypes.c_int)]
    rng = np.random.default_rng(7)
    kats = []
    shapes = [(6, 4), (1, 3), (17, 5), (40, 8), (64, 1332), (250, 1332), (3, 2), (12, 6)]
    for T, V in shapes:
        for variant in range(3):
            blank = V - 1
            if variant == 0:        # generic random probabilities
                p = rng.random((T, V)).astype(np.float32)
            elif variant == 1:      # quantised -> many exact ties (first max must win) and repeats
                p = (rng.integers(0, 3, (T, V)) / 2.0).astype(np.float32)
            else:                   # blank-dominated with runs of repeated labels
                ids = np.repeat(rng.integers(0, V, (T + 2) // 3), 3)[:T]
                ids[rng.random(T) < 0.5] = blank
                p = np.full((T, V), 0.01, np.float32)
                p[np.arange(T), ids] = 0.9
            out = (ctypes.c_int * T)()
            n = lib.ref_ctc_greedy(p.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), T, V, blank, out)
            kat = {"T": T, "V": V, "blank": blank, "expect": [int(out[i]) for i in range(n)]}
            if V <= 8:
                kat["probs"] = [[float(v) for v in row] for row in p]
            else:               # large cases: store the per-frame argmax instead of the matrix
                kat["frame_argmax"] = [int(v) for v in p.argmax(-1)]
                kat["seeded"] = False
            kats.append(kat)
    # the survey's hand KAT: rows argmax [1,1,3,1,0,tie->0], blank 3 -> [1,1,0]
    p = np.array([[0, 1, 0, 0], [0, .9, 0, 0], [0, 0, 0, 1], [0, 1, 0, 0], [1, 0, 0, 0], [.5, .5, 0, 0]], np.float32)
    out = (ctypes.c_int * 6)()
    n = lib.ref_ctc_greedy(p.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), 6, 4, 3, out)
    kats.append({"T": 6, "V": 4, "blank": 3, "expect": [int(out[i]) for i in range(n)],
                 "probs": [[float(v) for v in row] for row in p]})
    with open(os.path.join(OUT, "greedy_kat.json"), "w") as f:
        json.dump(kats, f)
    print("greedy KATs:", len(kats), "hand KAT ->", kats[-1]["expect"])


def make_beam_kats():
    """Known-answer tests from the reference's own prefix beam search (scorer-less), oracle/_ref/libref_ctc_beam.so."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref_beam"])
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_ctc_beam.so"))
    lib.ref_ctc_beam_search.restype = ctypes.c_int
    lib.ref_ctc_beam_search.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double),
                                        ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    rng = np.random.default_rng(11)
    cases = [(6, 4, 3, 1.0, 40, 1.0), (20, 8, 5, 1.0, 40, 1.0), (30, 12, 10, 0.99, 5, 1.0), (50, 30, 8, 0.9, 10, 3.0),
             (40, 60, 20, 1.0, 40, 2.0), (100, 200, 10, 0.999, 40, 4.0), (25, 6, 4, 0.5, 3, 1.0), (15, 5, 100, 1.0, 40, 1.0),
             (1, 7, 4, 1.0, 40, 1.0), (60, 1332, 10, 0.99, 40, 6.0), (250, 1332, 4, 0.999, 40, 8.0)]
    out = {}
    meta = []
    for k, (T, V, beam, cp, tn, temp) in enumerate(cases):
        z = rng.standard_normal((T, V)) * temp
        z[:, -1] += 1.0                                   # blank-leaning, like a CTC model
        p = np.exp(z - z.max(-1, keepdims=True))
        p = (p / p.sum(-1, keepdims=True)).astype(np.float32)   # float32-representable probabilities
        pd = np.ascontiguousarray(p, np.float64)
        sc = (ctypes.c_double * beam)()
        ids = (ctypes.c_int * (beam * T))()
        ln = (ctypes.c_int * beam)()
        n = lib.ref_ctc_beam_search(pd.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), T, V, beam, cp, tn, T, sc, ids, ln)
        e_ids = np.full((n, T), -1, np.int32)
        for i in range(n):
            e_ids[i, :ln[i]] = [ids[i * T + j] for j in range(ln[i])]
        out["probs_%d" % k] = p
        out["ids_%d" % k] = e_ids
        out["lens_%d" % k] = np.array([ln[i] for i in range(n)], np.int32)
        out["scores_%d" % k] = np.array([sc[i] for i in range(n)], np.float64)
        meta.append({"T": T, "V": V, "beam": beam, "cutoff_prob": cp, "cutoff_top_n": tn, "n": n})
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, "beam_kat.npz"), **out)
    print("beam KATs:", len(meta), "cases; best of the last:", out["ids_%d" % (len(cases) - 1)][0][:out["lens_%d" % (len(cases) - 1)][0]][:10])


def make_long_beam_kats():
    """The reference's own decoder on LONG inputs (hundreds of frames, scores of -1000 and below, where a float32 ulp is
    1e-4 and hypotheses tie in score all the time): beam_long_kat.npz.  prefix_compare orders by (score, last character)
    only and the reference leaves the rest to std::nth_element over its DFS-ordered vector, so these vectors pin what IS
    specified -- the ranked float32 scores, bit for bit -- and record how many hypotheses coincide beyond that."""
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_ctc_beam.so"))
    lib.ref_ctc_beam_search.restype = ctypes.c_int
    lib.ref_ctc_beam_search.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double),
                                        ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    rng = np.random.default_rng(2024)
    cases = [(500, 120, 10, 0.99, 40, 0.05), (500, 120, 100, 0.99, 40, 1.0), (400, 200, 25, 0.9999, 25, 0.3), (600, 96, 10, 0.99, 40, 2.5)]
    out, meta = {}, []
    for k, (T, V, beam, cp, tn, temp) in enumerate(cases):
        z = rng.standard_normal((T, V)) * temp
        p = np.exp(z - z.max(-1, keepdims=True))
        p = (p / p.sum(-1, keepdims=True)).astype(np.float32)
        pd = np.ascontiguousarray(p, np.float64)
        sc = (ctypes.c_double * beam)()
        ids = (ctypes.c_int * (beam * T))()
        ln = (ctypes.c_int * beam)()
        n = lib.ref_ctc_beam_search(pd.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), T, V, beam, cp, tn, T, sc, ids, ln)
        lens = np.array([ln[i] for i in range(n)], np.int32)
        e_ids = np.full((n, int(lens.max()) if n else 0), -1, np.int32)
        for i in range(n):
            e_ids[i, :ln[i]] = [ids[i * T + j] for j in range(ln[i])]
        out["probs_%d" % k] = p
        out["ids_%d" % k] = e_ids
        out["lens_%d" % k] = lens
        out["scores_%d" % k] = np.array([sc[i] for i in range(n)], np.float64)
        meta.append({"T": T, "V": V, "beam": beam, "cutoff_prob": cp, "cutoff_top_n": tn, "n": n})
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, "beam_long_kat.npz"), **out)
    print("long beam KATs:", len(meta), "cases")


def make_stateful_beam_kats():
    """The reference's stateful BeamDecoder (ctc_beam_search_decoder.cpp:217-405) fed in pieces, with a reset in
    between: the beam after every decode() call."""
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_ctc_beam.so"))
    lib.ref_beam_decoder_new.restype = ctypes.c_void_p
    lib.ref_beam_decoder_new.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int]
    lib.ref_beam_decoder_decode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.ref_beam_decoder_reset.argtypes = [ctypes.c_void_p]
    lib.ref_beam_decoder_free.argtypes = [ctypes.c_void_p]
    rng = np.random.default_rng(99)
    out = {}
    cases = [(9, 6, 1.0, 40, [5, 1, 7, 4]), (30, 10, 0.99, 8, [13, 12]), (5, 3, 0.9, 3, [1, 1, 1, 1, 9]),
             (1332, 10, 0.99, 40, [20, 11, 19])]
    for ci, (V, beam, cp, ctn, pieces) in enumerate(cases):
        T = sum(pieces)
        sharp = 4.0 if V > 100 else 1.5
        x = rng.standard_normal((2, T, V)) * sharp
        x[..., V - 1] += 2.0                                   # blank-heavy, like a trained CTC model
        p = np.exp(x - x.max(-1, keepdims=True))
        p = (p / p.sum(-1, keepdims=True)).astype(np.float32)
        h = lib.ref_beam_decoder_new(V, beam, cp, ctn)
        res = []
        for u in range(2):                                     # second utterance after reset()
            if u:
                lib.ref_beam_decoder_reset(h)
            o = 0
            for n in pieces:
                pd = np.ascontiguousarray(p[u, o:o + n], np.float64)
                o += n
                sc = np.zeros(beam)
                ids = -np.ones((beam, T), np.int32)
                ln = np.zeros(beam, np.int32)
                k = lib.ref_beam_decoder_decode(h, pd.ctypes.data, n, V, T, sc.ctypes.data, ids.ctypes.data, ln.ctypes.data)
                res.append((k, sc.astype(np.float32), ids, ln))
        lib.ref_beam_decoder_free(h)
        out["c%d_meta" % ci] = np.array([V, beam, ctn, len(pieces)], np.int32)
        out["c%d_cp" % ci] = np.array([cp])
        out["c%d_pieces" % ci] = np.array(pieces, np.int32)
        out["c%d_probs" % ci] = p
        out["c%d_n" % ci] = np.array([r[0] for r in res], np.int32)
        out["c%d_scores" % ci] = np.stack([r[1] for r in res])
        out["c%d_ids" % ci] = np.stack([r[2] for r in res])
        out["c%d_lens" % ci] = np.stack([r[3] for r in res])
    np.savez_compressed(os.path.join(OUT, "beam_stateful_kat.npz"), **out)
    print("beam_stateful_kat.npz:", len(cases), "cases")


def make_wer_kats():
    """(S+I+D)/N and the S / D / I split from the reference's own utils/xer.py (numpy-only, importable here)."""
    from utils import xer
    rng = np.random.default_rng(7)
    kats = []
    for n in range(60):
        lr, lh = int(rng.integers(1, 14)), int(rng.integers(0, 14))
        r = rng.integers(0, 6, lr).tolist()
        h = rng.integers(0, 6, lh).tolist()
        if n % 5 == 0:
            h = list(r)
        if n % 7 == 0 and len(r) > 2:
            h = r[1:] + [9]
        score, s, d, i = xer.wer(r, h)
        kats.append({"r": r, "h": h, "score": score, "s": s, "d": d, "i": i})
    json.dump(kats, open(os.path.join(OUT, "wer_kat.json"), "w"))
    print("wer_kat.json:", len(kats), "cases")


if __name__ == "__main__":
    sys.path.insert(0, REF)
    make_wer_kats()
    make_stateful_beam_kats()
    make_onnx_fixtures()
    make_greedy_kats()
    make_beam_kats()
    make_long_beam_kats()
