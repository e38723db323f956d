"""GPU parity tests on the BASELINE.json configurations THEMSELVES: the shapes that are
benchmarked are the shapes that are checked against the oracle.

  config 2  ConformerCTC(S), 13 + 1 blocks, 64 x 10 s, fp32, built exactly as bench.py builds it (Keras-default random
            encoder, the reference's exported CTCDecoder weights)
  config 3  StreamingConformerCTC (d = 256, 4 blocks, k = 5), batch = 64 streaming chunks, bf16 MFMA inputs
  config 5  ChunkConformer at chunk_conformerS.yml dimensions (15 + 1 + 2 + 1 blocks, 277 / 9171 classes), 30 s
            utterances (T = 750, band attention), prefix beam search

Token ids are compared with the oracle UNCONDITIONALLY (`assert_frames_and_ids`): every frame's argmax must be the
oracle's unless the oracle's own margin between the two candidates is inside ten times the measured logit error (an
fp32 forward cannot resolve such a frame against an fp64 one -- the reference's TF fp32 forward could not either);
those frames are listed, bounded in number, and the ids must then equal the collapse of the oracle argmax with exactly
those frames patched.  Nothing is skipped."""
import os

import numpy as np
import pytest

from helpers import (assert_frames_and_ids, assert_own_argmax, chunk_config_dict, co, golden_ctc_io, golden_ctc_weights, maxdiff, waves)

pytestmark = pytest.mark.gpu
TOL = 1e-3


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def oracle_weights(w):
    """model.get_weights_dict() (Keras variable shapes) -> what the oracle takes (2-D DFT kernels)."""
    w = dict(w)
    for k in list(w):
        if k.endswith(("mel_layer/real_kernels", "mel_layer/imag_kernels")):
            w[k] = np.asarray(w[k]).reshape(1024, 513)
    return w


# ---------------------------------------------------------------------------------------------------------------
# config 2
# ---------------------------------------------------------------------------------------------------------------
SAMPLED = [0, 21, 42, 63]


@pytest.fixture(scope="module")
def bench_model(torch_cuda):
    """exactly bench.build_model(): _build(seed=0) + tests/golden/ctc_decoder_weights.npz"""
    import bench
    m = bench.build_model(torch_cuda.device("cuda", 0), 0, 1, False)
    return m, oracle_weights(m.get_weights_dict())


def test_config2_batch64_10s_as_benched_ids_and_logits_vs_oracle(bench_model, torch_cuda):
    from tensorflowasr_amd.synthetic import synth_batch
    m, w = bench_model
    cfg = dict(co.CONFORMER_S)
    x = synth_batch(0, 64, 160000)                        # the bench's own inputs (rank 0)
    xd = torch_cuda.from_numpy(x).cuda()
    ids, lens = m.recognize(xd)
    ids, lens = ids.cpu().numpy().copy(), lens.cpu().numpy().copy()
    assert ids.shape == (64, 250)
    enc = m.encode(xd)
    logits, amax = m.ctc_logits(enc, return_argmax=True)
    enc, logits, amax = enc.cpu().numpy(), logits.cpu().numpy(), amax.cpu().numpy()
    enc_ref = co.conformer_encoder(x[SAMPLED].astype(np.float64), w, cfg)
    lg_ref = co.ctc_decoder(enc_ref, w, cfg)
    e_enc = maxdiff(enc[SAMPLED], enc_ref)
    assert e_enc < TOL
    err, report = assert_frames_and_ids(logits[SAMPLED], amax[SAMPLED], ids[SAMPLED], lens[SAMPLED], lg_ref,
                                        [250] * len(SAMPLED), 1331, tag="config2_as_benched_trained_ctc_head")
    print("config 2 as benched: encoder max|d| %.3g, logits max|d| %.3g, undecided frames %d / %d"
          % (e_enc, err, len(report), 250 * len(SAMPLED)))
    # every one of the 64: the integer path is exact given the kernel's own logits
    gid, glen = co.ctc_collapse(amax, [250] * 64, 1331)
    assert np.array_equal(ids, gid) and np.array_equal(lens, glen)


def test_config2_batch64_10s_nonblank_head_ids_vs_oracle(torch_cuda):
    """the same shape with a CTC head that emits tokens (the trained CTCDecoder answers `blank` to a random encoder):
    13 + 1 blocks, synthetic head centred on the batch so that the argmax follows the per-frame deviations."""
    from tensorflowasr_amd.models import ConformerCTC
    from tensorflowasr_amd.synthetic import synth_batch
    import bench
    cfg = dict(co.CONFORMER_S)
    w = co.encoder_weights(cfg, seed=0)
    w.update(co.ctc_decoder_weights(cfg, 1332, seed=1))
    x = synth_batch(0, 64, 160000)
    enc_ref = co.conformer_encoder(x[SAMPLED].astype(np.float64), w, cfg)
    w["fully_connected/bias"] = np.zeros(1332, np.float32)
    w["fully_connected/bias"] = (-co.ctc_decoder(enc_ref, w, cfg).mean(axis=(0, 1))).astype(np.float32)
    lg_ref = co.ctc_decoder(enc_ref, w, cfg)
    m = ConformerCTC(1332, **bench.S_CFG)
    m.load_weights(w, by_name=False)
    xd = torch_cuda.from_numpy(x).cuda()
    ids, lens = m.recognize(xd)
    ids, lens = ids.cpu().numpy().copy(), lens.cpu().numpy().copy()
    logits, amax = m.ctc_logits(m.encode(xd), return_argmax=True)
    logits, amax = logits.cpu().numpy(), amax.cpu().numpy()
    err, report = assert_frames_and_ids(logits[SAMPLED], amax[SAMPLED], ids[SAMPLED], lens[SAMPLED], lg_ref,
                                        [250] * len(SAMPLED), 1331, max_undecided=0.02, tag="config2_token_emitting_head")
    assert lens[SAMPLED].min() > 100                      # real token sequences
    print("config 2, token-emitting head: logits max|d| %.3g, undecided frames %d / 1000 %s" % (err, len(report), report[:4]))


def _all64_against_fixture(m, head, torch, fixture="config2_oracle_b64.npz"):
    """Every one of the 64 benched utterances against tests/golden/config2_oracle_b64.npz (fp64 oracle, written by
    tests/golden/make_config2_b64.py): encoder rows, logits (all classes of every 50th frame, the oracle's four largest
    classes of EVERY frame), per-frame argmax, greedy ids and lengths.  A frame may differ in argmax only when the oracle's
    own margin between the two classes is within ten times the measured logit error; ids must then equal the collapse of the
    oracle argmax with exactly those frames patched (helpers.assert_frames_and_ids, on the stored top-4 instead of full rows)."""
    from helpers import _parity_log
    from tensorflowasr_amd.synthetic import synth_batch
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", fixture))
    x = synth_batch(0, 64, 160000)
    xd = torch.from_numpy(x).cuda()
    ids, lens = m.recognize(xd)
    ids, lens = ids.cpu().numpy().copy(), lens.cpu().numpy().copy()
    enc = m.encode(xd)
    logits, amax = m.ctc_logits(enc, return_argmax=True)
    enc, logits, amax = enc.cpu().numpy(), logits.cpu().numpy(), amax.cpu().numpy()
    e_enc = maxdiff(enc[:, ::10], g[head + "_enc_every10"])
    top_idx, top_val = g[head + "_top4_idx"].astype(np.int64), g[head + "_top4_val"]
    e_lg = max(maxdiff(logits[:, ::50], g[head + "_logits_every50"]), maxdiff(np.take_along_axis(logits, top_idx, -1), top_val))
    assert e_enc < TOL and e_lg < TOL, (e_enc, e_lg)
    assert np.array_equal(amax, logits.argmax(-1)), "in-kernel argmax != argmax of the kernel's own logits"
    ra = top_idx[..., 0]
    report = []
    for b, t in np.argwhere(amax != ra):
        pos = np.flatnonzero(top_idx[b, t] == amax[b, t])
        assert pos.size, "frame (%d, %d): the kernel's class %d is not among the oracle's four largest" % (b, t, amax[b, t])
        margin = float(top_val[b, t, 0] - top_val[b, t, pos[0]])
        assert margin <= 10 * e_lg, "argmax differs on a frame the oracle decides clearly: (%d, %d) margin %.3g, logit error %.3g" % (b, t, margin, e_lg)
        report.append((int(b), int(t), margin))
    _parity_log({"tag": ("config2_all64_vs_reference_code_" if fixture.startswith("tf_") else "config2_all64_") + head, "frames": int(ra.size), "logits_max_abs_err": e_lg, "excused_frames": len(report),
                 "excused": [{"utt": r[0], "frame": r[1], "oracle_margin": r[2]} for r in report[:20]]})
    assert len(report) <= 0.005 * ra.size
    patched = ra.copy()
    for b, t, _ in report:
        patched[b, t] = amax[b, t]
    rid, rlen = co.ctc_collapse(patched.astype(np.int32), [250] * 64, 1331)
    assert np.array_equal(lens, rlen) and np.array_equal(ids, rid)
    if not report:
        assert np.array_equal(ids, g[head + "_ids"]) and np.array_equal(lens, g[head + "_lens"])
    print("config 2, all 64 utterances, head '%s': encoder max|d| %.3g, logits max|d| %.3g, %d / 16000 frames excused, %d tokens"
          % (head, e_enc, e_lg, len(report), int(lens.sum())))
    return e_lg, report


def test_config2_all_64_utterances_as_benched_vs_oracle_fixture(bench_model, torch_cuda):
    m, _ = bench_model
    _all64_against_fixture(m, "trained", torch_cuda)


def test_config2_all_64_utterances_token_emitting_head_vs_oracle_fixture(torch_cuda):
    """the head that emits tokens (15 411 over the batch): ids of all 64 utterances, every frame's argmax"""
    from tensorflowasr_amd.models import ConformerCTC
    import bench
    cfg = dict(co.CONFORMER_S)
    w = co.encoder_weights(cfg, seed=0)
    w.update(co.ctc_decoder_weights(cfg, 1332, seed=1))
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config2_oracle_b64.npz"))
    w["fully_connected/bias"] = g["tokens_fc_bias"]
    m = ConformerCTC(1332, **bench.S_CFG)
    m.load_weights(w, by_name=False)
    _, report = _all64_against_fixture(m, "tokens", torch_cuda)
    assert len(report) <= 0.002 * 16000


def test_config2_all_64_utterances_vs_the_reference_code_both_heads(bench_model, torch_cuda):
    """Round 5: the same comparison against tests/golden/tf_config2_b64.npz -- the 64 benched utterances through the reference's
    OWN ConformerEncoder + CTCDecoder + ctc_decode (float32 run on the NumPy stand-in, tests/golden/make_tf_config2_b64.py):
    encoder rows, logits, every frame's arg-max and the greedy ids of all 64 utterances for the benched model and for the
    token-emitting head, with the same rule for frames the reference's own float32 logits cannot decide."""
    from tensorflowasr_amd.models import ConformerCTC
    import bench
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tf_config2_b64.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/tf_config2_b64.npz not generated (python tests/golden/make_tf_config2_b64.py)")
    m, _ = bench_model
    _, rep1 = _all64_against_fixture(m, "trained", torch_cuda, fixture="tf_config2_b64.npz")
    cfg = dict(co.CONFORMER_S)
    w = co.encoder_weights(cfg, seed=0)
    w.update(co.ctc_decoder_weights(cfg, 1332, seed=1))
    w["fully_connected/bias"] = np.load(os.path.join(os.path.dirname(path), "config2_oracle_b64.npz"))["tokens_fc_bias"]
    m2 = ConformerCTC(1332, **bench.S_CFG)
    m2.load_weights(w, by_name=False)
    _, rep2 = _all64_against_fixture(m2, "tokens", torch_cuda, fixture="tf_config2_b64.npz")
    print("config 2, all 64 utterances against the reference's own code: excused frames %d (benched model) / %d (token head) of 16 000"
          % (len(rep1), len(rep2)))
    assert len(rep1) <= 8 and len(rep2) <= 0.002 * 16000


def test_config2_model_on_64_utterances_of_20_s_vs_oracle(bench_model, torch_cuda):
    """Round 6: the benched model on 64 x 20 s -- T = 500 encoder frames, beyond the 256 of the single-block attention kernel: the
    key-block kernel (attention_split_long_kernel) at the benched batch size.  Utterances 0 and 63 against the fp64 oracle (logits,
    every frame's arg-max, greedy ids), every utterance's in-kernel arg-max against its own logits."""
    from tensorflowasr_amd.synthetic import synth_batch
    m, w = bench_model
    x = synth_batch(0, 64, 320000)
    xd = torch_cuda.from_numpy(x).cuda()
    ids, lens = m.recognize(xd)
    ids, lens = ids.cpu().numpy().copy(), lens.cpu().numpy().copy()
    logits, amax = m.ctc_logits(m.encode(xd), return_argmax=True)
    logits, amax = logits.cpu().numpy(), amax.cpu().numpy()
    assert logits.shape == (64, 500, 1332) and np.array_equal(amax, logits.argmax(-1))
    pick = [0, 63]
    cfg = dict(co.CONFORMER_S)
    enc_ref = co.conformer_encoder(x[pick].astype(np.float64), w, cfg)
    lg_ref = co.ctc_decoder(enc_ref, w, cfg)
    err, report = assert_frames_and_ids(logits[pick], amax[pick], ids[pick], lens[pick], lg_ref, [500] * 2, 1331, tag="config2_model_64x20s")
    print("64 x 20 s: logits max|d| %.3g, undecided frames %d / 1000" % (err, len(report)))
    m.prepare(64, 160000)


def test_config2_trained_ctc_decoder_at_batch64(torch_cuda):
    """the reference's exported CTCDecoder on 64 x 250 frames (the fused block kernels' row count) with inputs that make
    it emit tokens: argmax identical to the reference graph's own output (tests/golden/ctc_decoder_io.npz, 26 % of
    the frames non-blank), every copy bit-identical."""
    from tensorflowasr_amd.models import CTCDecoder, ctc_greedy_decode
    io = golden_ctc_io()
    dec = CTCDecoder(1332, dmodel=144, num_blocks=1, head_size=36, num_heads=4, kernel_size=32)
    dec.load_weights(golden_ctc_weights(), by_name=False)
    xb = np.tile(io["x_b"], (64, 1, 1))
    assert xb.shape == (64, 250, 144)
    lg, am = dec(xb, return_argmax=True)
    lg, am = lg.cpu().numpy(), am.cpu().numpy()
    assert np.array_equal(am[0], io["argmax_b"][0])
    assert maxdiff(lg[0, ::8], io["logits_b_every8"][0]) < TOL
    assert all(np.array_equal(am[0], am[b]) for b in range(64))
    ids, lens = ctc_greedy_decode(am, None, 1331)
    rid, rlen = co.ctc_collapse(np.tile(io["argmax_b"], (64, 1)), [250] * 64, 1331)
    assert np.array_equal(ids.cpu().numpy(), rid) and np.array_equal(lens.cpu().numpy(), rlen) and rlen[0] > 10


# ---------------------------------------------------------------------------------------------------------------
# config 3: StreamingConformerCTC, batch = 64 streaming chunks, bf16 MFMA inputs
# ---------------------------------------------------------------------------------------------------------------
def test_config3_64_streaming_chunks_bf16_vs_rounding_oracle(torch_cuda):
    """Streaming_ConformerS.yml at full depth (4 blocks, d = 256, k = 5; CTCDecoder 1 block k = 32) on 64 chunks of
    8000 samples = 4 streams x 16 chunks, the global CTC over each stream's 208 frames.  Against the oracle that
    rounds both GEMM operands to bf16 (pins the arithmetic), and against the exact fp32 oracle (what the mode costs)."""
    from tensorflowasr_amd.models import ConformerCTC
    cfg = dict(co.STREAMING_S)
    V = 1332
    w = co.encoder_weights(cfg, seed=51)
    w.update(co.ctc_decoder_weights(cfg, V, seed=52))
    kw = dict(dmodel=256, reduction_factor=4, num_blocks=4, head_size=64, num_heads=4, kernel_size=5, fc_factor=0.5,
              sample_rate=16000, n_mels=80, stride_ms=10, chunk_size=8000, ctcdecoder_num_blocks=1,
              ctcdecoder_kernel_size=32, ctcdecoder_fc_factor=0.5)
    assert all(cfg[k] == kw[k] for k in ("dmodel", "num_blocks", "head_size", "num_heads", "kernel_size"))
    x = waves(4, 16 * 8000, 300)                                        # 4 streams x 16 chunks = 64 chunks
    out = {}
    for dt in ("bfloat16", "float32"):
        m = ConformerCTC(V, gemm_dtype=dt, **kw)
        m.load_weights(w, by_name=False)
        as_chunks = m.encode(x.reshape(64, 8000))                       # batch = 64 chunks, one per "stream"
        enc = m.encode(x)                                               # the same 64 chunks as 4 streams
        assert enc.shape == (4, 16 * 13, 256)
        assert np.array_equal(as_chunks.cpu().numpy().reshape(4, 208, 256), enc.cpu().numpy())
        logits, amax = m.ctc_logits(enc, return_argmax=True)
        ids, lens = m.recognize(x)
        out[dt] = (enc.cpu().numpy(), logits.cpu().numpy(), amax.cpu().numpy(), ids.cpu().numpy().copy(), lens.cpu().numpy().copy())
    exact_enc = co.streaming_conformer_encoder(x.astype(np.float64), w, cfg, 8000)
    exact_lg = co.ctc_decoder(exact_enc, w, cfg)
    co.GEMM_ROUND_BF16 = True
    try:
        r_enc = co.streaming_conformer_encoder(x.astype(np.float64), w, cfg, 8000)
        r_lg = co.ctc_decoder(r_enc, w, cfg)
    finally:
        co.GEMM_ROUND_BF16 = False
    enc16, lg16, am16, ids16, lens16 = out["bfloat16"]
    enc32, lg32, am32, ids32, lens32 = out["float32"]
    # fp32 mode: the usual contract, ids unconditionally
    assert maxdiff(enc32, exact_enc) < TOL
    assert_frames_and_ids(lg32, am32, ids32, lens32, exact_lg, [208] * 4, V - 1, max_undecided=0.02)
    # bf16 mode vs the rounding oracle: operands within 1e-7 of a rounding boundary flip, each flip is one bf16 ulp
    e_enc, e_lg = np.abs(enc16 - r_enc), np.abs(lg16 - r_lg)
    print("config 3 bf16 vs rounding oracle: encoder max %.3g mean %.3g; logits max %.3g mean %.3g"
          % (e_enc.max(), e_enc.mean(), e_lg.max(), e_lg.mean()))
    assert e_enc.max() < 2e-2 and e_enc.mean() < 2e-3
    assert e_lg.max() < 4e-2 and e_lg.mean() < 3e-3
    # integer path exact given the kernel's own logits
    assert_own_argmax(am16, lg16)
    gid, glen = co.ctc_collapse(am16, [208] * 4, V - 1)
    assert np.array_equal(ids16, gid) and np.array_equal(lens16, glen)
    # and what bf16 costs against exact arithmetic (report max abs diff and ids agreement)
    agree = float((am16 == co.frame_argmax(exact_lg)).mean())
    print("config 3 bf16 vs exact fp64: encoder max|d| %.3g, logits max|d| %.3g, argmax agreement %.4f"
          % (maxdiff(enc16, exact_enc), maxdiff(lg16, exact_lg), agree))
    assert maxdiff(enc16, exact_enc) < 0.2 and maxdiff(lg16, exact_lg) < 0.3 and agree > 0.95


def test_config3_as_benched_64_streams_with_260_frames_of_history(torch_cuda):
    """Round-4 review: bench.py's config-3 step is 64 STREAMS -- one new 0.5 s chunk each through the encoder (64 x 13 rows)
    and the reference's global CTC over every stream's 10 s of history (64 x 260 frames: conformer_blocks.py:574-594, :385-438;
    test_asr.py:116-165) -- while the parity test above runs 4 streams x 208 frames.  The same objects, shapes and bf16 mode as
    bench.extra_config3: every stream's encoder rows and logits against the bf16-rounding oracle on streams 0, 21 and 63 (the
    streams are independent; two more are run alone and must stay within the same bound of their rows in the batch of 64),
    the in-kernel argmax against the kernel's own logits on all 64, ids against their collapse."""
    torch = torch_cuda
    from tensorflowasr_amd.models import CTCDecoder, StreamingConformerEncoder, ctc_greedy_decode
    cfg = dict(co.STREAMING_S)
    B, chunk, hist, d, V = 64, 8000, 20, 256, 1332
    w = co.encoder_weights(cfg, seed=53)
    w.update(co.ctc_decoder_weights(cfg, V, seed=54))
    enc = StreamingConformerEncoder(dmodel=d, reduction_factor=4, num_blocks=4, head_size=64, num_heads=4, kernel_size=5, fc_factor=0.5,
                                    sample_rate=16000, n_mels=80, stride_ms=10, mel_layer_type="Melspectrogram", gemm_dtype="bfloat16")
    enc.add_chunk_size(chunk, 80, 640)
    enc.load_weights({k: v for k, v in w.items() if not k.startswith(("project/", "decoder_conformer_block_", "fully_connected/"))}, by_name=False)
    ctc = CTCDecoder(num_classes=V, dmodel=d, num_blocks=1, head_size=64, num_heads=4, kernel_size=32, fc_factor=0.5, gemm_dtype="bfloat16")
    ctc.load_weights({k: v for k, v in w.items() if k.startswith(("project/", "decoder_conformer_block_", "fully_connected/"))}, by_name=False)
    x = waves(B, chunk, 400)
    history = np.random.default_rng(7).standard_normal((B, (hist - 1) * 13, d)).astype(np.float32)
    e = enc(x)
    assert tuple(e.shape) == (B, 13, d)
    h = torch.cat([torch.from_numpy(history).to(e.device), e], 1)
    assert tuple(h.shape) == (B, hist * 13, d)
    logits, amax = ctc(h, return_argmax=True)
    ids, lens = ctc_greedy_decode(amax, None, blank=V - 1)
    e, logits, amax, ids, lens = (t.cpu().numpy() for t in (e, logits, amax, ids, lens))
    assert_own_argmax(amax, logits)
    gid, glen = co.ctc_collapse(amax, [hist * 13] * B, V - 1)
    assert np.array_equal(ids, gid) and np.array_equal(lens, glen)
    # the head without logits (what the bench step runs) keeps the same running argmax
    _, amax2 = ctc(h, return_argmax=True, return_logits=False)
    assert np.array_equal(amax2.cpu().numpy(), amax)
    pick = [0, 21, 63]
    co.GEMM_ROUND_BF16 = True
    try:
        r_enc = co.streaming_conformer_encoder(x[pick].astype(np.float64), w, cfg, chunk)
        r_lg = co.ctc_decoder(np.concatenate([history[pick].astype(np.float64), r_enc], 1), w, cfg)
    finally:
        co.GEMM_ROUND_BF16 = False
    e_enc, e_lg = np.abs(e[pick] - r_enc), np.abs(logits[pick] - r_lg)
    print("config 3 as benched, bf16 vs rounding oracle: encoder max %.3g mean %.3g; logits max %.3g mean %.3g"
          % (e_enc.max(), e_enc.mean(), e_lg.max(), e_lg.mean()))
    assert e_enc.max() < 2e-2 and e_enc.mean() < 2e-3
    assert e_lg.max() < 4e-2 and e_lg.mean() < 3e-3
    # the other 61 streams: a stream run alone (other kernels are selected at 13 / 260 rows than at 832 / 16 640: a hidden value
    # on a bf16 rounding boundary may fall the other way, one bf16 ulp) stays inside the bound the oracle comparison uses
    for b in (5, 40):
        e1 = enc(x[b:b + 1])
        d_e = np.abs(e1.cpu().numpy()[0] - e[b])
        l1 = ctc(torch.cat([torch.from_numpy(history[b:b + 1]).to(e1.device), e1], 1))
        d_l = np.abs(l1.cpu().numpy()[0] - logits[b])
        assert d_e.max() < 2e-2 and d_e.mean() < 2e-3 and d_l.max() < 4e-2 and d_l.mean() < 3e-3, (b, d_e.max(), d_l.max())


# ---------------------------------------------------------------------------------------------------------------
# config 5: ChunkConformer (full chunk_conformerS dims) on 30 s utterances + prefix beam search
# ---------------------------------------------------------------------------------------------------------------
def test_config5_chunk_conformer_2x30s_stages_and_beam_vs_oracle(torch_cuda):
    from tensorflowasr_amd.models import ChunkConformer, ctc_prefix_beam_decode
    from test_gpu_parity import _pick_bias_for_ragged_counts
    torch = torch_cuda
    cfg = dict(co.CHUNK_S)
    assert (cfg["enc_num_blocks"], cfg["picker_num_classes"], cfg["decoder_num_classes"]) == (15, 277, 9171)
    w = co.chunk_weights(cfg, seed=5)
    x = waves(2, 480000, 200)
    w["picker/fully_connected/bias"][-1] = _pick_bias_for_ragged_counts(cfg, w, x)
    ref = co.chunk_predict(x.astype(np.float64), w, cfg)
    assert ref["front"].shape == (2, 750, 144)
    m = ChunkConformer(chunk_config_dict(cfg), cfg["picker_num_classes"], cfg["decoder_num_classes"])
    m.load_weights(w, by_name=False)
    got = m.predict(x, stages=True)
    for k in ("front", "enc", "picker_logits", "picker_hidden"):
        e = maxdiff(got[k].cpu().numpy(), ref[k])
        print("config 5 stage %-14s max|d| %.3g" % (k, e))
        assert e < TOL, k
    # the picker's decisions drive feature_pick: they must be the oracle's on every frame
    pa_g, pa_r = got["picker_logits"].cpu().numpy().argmax(-1), ref["picker_logits"].argmax(-1)
    blank = cfg["picker_num_classes"] - 1
    assert np.array_equal(pa_g == blank, pa_r == blank)
    assert np.array_equal(got["counts"], ref["counts"])
    assert 100 < ref["counts"].min() and ref["counts"].max() < 750
    for k in ("picked", "helper", "text_logits"):
        assert got[k].shape == ref[k].shape, k
        e = maxdiff(got[k].cpu().numpy(), ref[k])
        print("config 5 stage %-14s max|d| %.3g" % (k, e))
        assert e < TOL, k
    # prefix beam search: device path (fused softmax + top-n on the GPU logits, host search) against the HOST path
    # (`mi355asr_ctc_prefix_beam_host`, bit-exact vs the reference decoder's KATs) on the ORACLE's probabilities
    counts = ref["counts"]
    p_ref = co.softmax(ref["text_logits"]).astype(np.float32)
    for beam in (10, 100):
        a = ctc_prefix_beam_decode(got["text_logits"], counts, beam, 0.99, 40, is_logits=True)
        b = ctc_prefix_beam_decode(p_ref, counts, beam, 0.99, 40)
        for u in range(2):
            na, nb = a[1][u, 0], b[1][u, 0]
            # best hypothesis: same token sequence, score within the logit error accumulated over the frames
            assert na == nb and np.array_equal(a[0][u, 0, :na], b[0][u, 0, :nb]), (beam, u)
            assert abs(a[2][u, 0] - b[2][u, 0]) < 1e-4 * counts[u] + 1e-3
            assert na > 20
        # the whole beam: neighbouring hypotheses are ~1e-3 apart at scores of -4400 (fp32 resolution 5e-4), so ranks
        # below the first may swap under a 1e-6 perturbation of the logits; the hypothesis SETS must agree
        same, common, tot = 0, 0, 0
        for u in range(2):
            n = int(min(a[3][u], b[3][u]))
            ha = [tuple(a[0][u, i, :a[1][u, i]]) for i in range(n)]
            hb = [tuple(b[0][u, i, :b[1][u, i]]) for i in range(n)]
            same += sum(int(p == q) for p, q in zip(ha, hb))
            common += len(set(ha) & set(hb))
            tot += n
            sa, sb = dict(zip(ha, a[2][u, :n])), dict(zip(hb, b[2][u, :n]))
            for hyp in set(ha) & set(hb):                      # the same hypothesis scores the same on both paths
                assert abs(sa[hyp] - sb[hyp]) < 1e-4 * counts[u] + 1e-3
        print("config 5 beam %d: %d / %d hypotheses identical in rank, %d in common" % (beam, same, tot, common))
        assert common >= 0.9 * tot


def test_config5_batch16_as_benched_vs_oracle_and_pipelined_beam(torch_cuda):
    """BASELINE config 5 at the per-GPU batch bench.py runs (16 x 30 s): grid shapes of pick / gather / beam kernels and the
    ragged T_pick padding differ from the 2-utterance test above.  The oracle follows two sampled utterances (an utterance's
    result does not depend on its batch); the prefix beam search of the whole batch is compared with the host search
    (bit-exact against the reference decoder's KATs) on the same GPU logits, sequentially and through ChunkBeamPipeline
    (search of batch n on a second stream while batch n + 1 is predicted)."""
    from tensorflowasr_amd.models import ChunkBeamPipeline, ChunkConformer, ctc_prefix_beam_decode
    from test_gpu_parity import _pick_bias_for_ragged_counts
    torch = torch_cuda
    cfg = dict(co.CHUNK_S)
    w = co.chunk_weights(cfg, seed=5)
    B = 16
    x = waves(B, 480000, 200)
    w["picker/fully_connected/bias"][-1] = _pick_bias_for_ragged_counts(cfg, w, x[[1, 13]])   # no frame of the two sampled
    m = ChunkConformer(chunk_config_dict(cfg), cfg["picker_num_classes"], cfg["decoder_num_classes"])   # utterances sits on the pick boundary
    m.load_weights(w, by_name=False)
    got = m.predict(x, stages=True)
    counts = got["counts"]
    assert counts.shape == (B,) and len(set(counts.tolist())) > 4, counts          # ragged
    Tp = got["text_logits"].shape[1]
    assert Tp == counts.max()
    lg = got["text_logits"].cpu().numpy()
    hs, fc = cfg["head_size"], cfg["fc_factor"]
    for u in (1, 13):
        ref = co.chunk_predict(x[u:u + 1].astype(np.float64), w, cfg)
        n = int(ref["counts"][0])
        assert counts[u] == n
        for k in ("enc", "picker_hidden"):
            assert maxdiff(got[k][u].cpu().numpy(), ref[k][0]) < TOL, (k, u)
        # helper + text decoder see the picked frames zero-padded to the BATCH maximum (feature_pick), and the padding takes
        # part in the band attention / causal conv of the frames near the end: the oracle follows with the same padding
        picked = np.zeros((1, Tp, cfg["dmodel"]))
        picked[0, :n] = ref["picked"][0]
        _, hlp = co.chunk_stack(picked, w, "helper", "block_", cfg["helper_num_blocks"], hs, cfg["helper_win_front"],
                                cfg["helper_win_back"], fc, False, False)
        tl, _ = co.chunk_stack(hlp, w, "decoder", "block_", cfg["decoder_num_blocks"], hs, cfg["decoder_win_front"],
                               cfg["decoder_win_back"], fc, True, True)
        e = maxdiff(lg[u], tl[0])
        print("config 5, batch 16, utterance %d: %d of %d picked frames, text logits max|d| %.3g" % (u, n, Tp, e))
        assert e < TOL
        assert np.array_equal(got["text_argmax"][u].cpu().numpy(), lg[u].argmax(-1))
    # beam search of the 16 utterances: device search == host search on the SAME probabilities, bit for bit -- ids, lengths,
    # float32 scores, hypothesis counts, at beam 10 (the one-key-per-thread path) and beam 100 (the radix path).  Rounds 2-3
    # accepted 80 % set overlap here: their device search rounded expf / logf "correctly", the host's C library does not
    # (csrc/refmath.h), and over ~600 frames an ulp reordered hypotheses that are ~1e-3 apart.
    probs = torch.softmax(got["text_logits"], -1)
    probs_h = probs.cpu().numpy()
    for beam in (10, 100):
        dev = ctc_prefix_beam_decode(probs, counts, beam, 0.99, 40)
        host = ctc_prefix_beam_decode(probs_h, counts, beam, 0.99, 40)
        for name, a, b in zip(("ids", "lens", "scores", "n_hyp"), dev, host):
            bad = np.argwhere(np.asarray(a) != np.asarray(b))
            assert bad.size == 0, "beam %d: %s differ at %s (%d entries)" % (beam, name, bad[:3].tolist(), len(bad))
        full = host[3] == beam                 # the other utterances' text logits are blank-dominated: the empty prefix alone survives
        assert full.sum() >= B // 2 and host[1][full, 0].min() > 20 and (host[3][~full] == 1).all()
        print("config 5, batch 16, beam %d: device search == host search on %d hypotheses (ids, lens, scores bit for bit)"
              % (beam, int(host[3].sum())))
    host = ctc_prefix_beam_decode(probs_h, counts, 10, 0.99, 40)
    fused = ctc_prefix_beam_decode(got["text_logits"], counts, 10, 0.99, 40, is_logits=True)     # softmax fused into the top-n kernel
    for u in range(B):
        assert abs(fused[2][u, 0] - host[2][u, 0]) < 1e-4 * counts[u] + 1e-3
    # pipelined: three batches through ChunkBeamPipeline == the sequential calls
    pipe = ChunkBeamPipeline(m, beam_width=10, cutoff_prob=0.99, cutoff_top_n=40)
    xs = [x, x[::-1].copy(), x]
    outs = [pipe.push(xx) for xx in xs] + [pipe.flush()]
    pipe.close()
    assert outs[0] is None
    for xx, res in zip(xs, outs[1:]):
        lgs, cs = m.predict(xx)
        seq = ctc_prefix_beam_decode(lgs, cs, 10, 0.99, 40, is_logits=True)
        for a, b in zip(res, seq):
            assert np.array_equal(a, b)
    assert np.array_equal(outs[1][3], outs[3][3]) and np.array_equal(outs[1][1], outs[3][1])


# ---------------------------------------------------------------------------------------------------------------
# dmodel 256 / 512 at the benched row counts: the slab-ring GEMM path against the kernels it replaces
# ---------------------------------------------------------------------------------------------------------------
def test_ring_gemm_at_16000_rows_equals_the_fp32_and_bf16_kernels(torch_cuda, tmp_path):
    """gemm_ring.hip takes over from 1500 rows on, where the oracle is too slow to follow every utterance (its own parity
    tests force it for small batches; here it follows ONE sampled utterance of the ConformerM batch).  Here the launch shapes of the benched sizes themselves -- ConformerM and ConformerL on 64 x 10 s
    (16 000 rows: two row tiles per wave, several column chunks per workgroup), and the bf16 CTC decoder on 64 x 260
    history frames -- against the same build with MI355ASR_GEMM_RING=0 (fp32-MFMA chains / per-wave bf16 streams, both
    pinned to the oracle at small sizes)."""
    import subprocess
    import sys
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, "tests")
from helpers import co, waves
from tensorflowasr_amd.models import ConformerCTC, CTCDecoder
out = {}
x = waves(64, 160000, 77)
for name, kw in (("M", dict(dmodel=256, num_blocks=2, head_size=64, num_heads=4)), ("L", dict(dmodel=512, num_blocks=1, head_size=64, num_heads=8))):
    m = ConformerCTC(1332, **kw)
    m._build(seed=3)
    enc = m.encode(x)
    logits = m.ctc_logits(enc)
    out[name + "_enc"] = enc.cpu().numpy()[::7]
    out[name + "_logits"] = logits.cpu().numpy()[::7]
    if name == "M" and len(sys.argv) > 2:           # the oracle follows one sampled utterance through the benched launch shapes
        cfg = dict(co.CONFORMER_S, dmodel=256, num_blocks=2, head_size=64, num_heads=4)
        wo = {k: (np.asarray(v).reshape(1024, 513) if k.endswith(("mel_layer/real_kernels", "mel_layer/imag_kernels")) else v)
              for k, v in m.get_weights_dict().items()}
        out["M_oracle_enc21"] = co.conformer_encoder(x[21:22].astype(np.float64), wo, cfg)
        out["M_gpu_enc21"] = enc[21:22].cpu().numpy()
    del m
    torch.cuda.empty_cache()
ctc = CTCDecoder(num_classes=1332, dmodel=256, num_blocks=1, head_size=64, num_heads=4, kernel_size=32, fc_factor=0.5,
                 gemm_dtype="bfloat16")
ctc._build(seed=4)
h = torch.from_numpy(np.random.default_rng(5).standard_normal((64, 260, 256)).astype(np.float32)).cuda()
lg, am = ctc(h, return_argmax=True)
out["bf16_logits"] = lg.cpu().numpy()
out["bf16_argmax"] = am.cpu().numpy()
np.savez(sys.argv[1], **out)
'''
    res = {}
    for tag, extra in (("ring", {}), ("plain", {"MI355ASR_GEMM_RING": "0"})):
        f = str(tmp_path / (tag + ".npz"))
        r = subprocess.run([sys.executable, "-c", code, f] + (["oracle"] if tag == "ring" else []), env=dict(os.environ, **extra), capture_output=True, text=True,
                           timeout=1200, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0, r.stderr[-2000:]
        res[tag] = np.load(f)
    d_or = maxdiff(res["ring"]["M_gpu_enc21"], res["ring"]["M_oracle_enc21"])
    print("ConformerM (2 blocks) at 64 x 10 s, utterance 21, ring kernels vs the fp64 oracle: max |d| = %.3g" % d_or)
    assert d_or < TOL
    for k in ("M_enc", "M_logits", "L_enc", "L_logits"):
        d = maxdiff(res["ring"][k], res["plain"][k])
        print(k, "ring vs fp32-MFMA kernels: max |d| = %.3g" % d)
        assert d < TOL, (k, d)
        assert np.abs(res["plain"][k]).max() > 0.1                      # not a comparison of zeros
    # bf16 mode: same rounding points, other summation order -- a handful of activations land on the other side of a
    # bf16 rounding boundary (one bf16 ulp each)
    e = np.abs(res["ring"]["bf16_logits"] - res["plain"]["bf16_logits"])
    agree = float((res["ring"]["bf16_argmax"] == res["plain"]["bf16_argmax"]).mean())
    print("bf16 CTC decoder, ring vs per-wave kernels: max %.3g mean %.3g, argmax agreement %.4f" % (e.max(), e.mean(), agree))
    assert e.max() < 4e-2 and e.mean() < 1e-3 and agree > 0.99
