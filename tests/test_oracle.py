"""CPU tests of the oracle itself: pinned against the reference's golden data where the reference has any
(exported CTCDecoder graph, C++ greedy decoder), cross-checked against independent restatements elsewhere."""
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN, co, golden_ctc_io, golden_ctc_weights


# ---- pinned rows ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_ctc_decoder_matches_reference_exported_graph(dtype):
    """ConformerBlock + CTCDecoder restatement vs logits the reference's own ctc_model.onnx produces."""
    w, io = golden_ctc_weights(), golden_ctc_io()
    y = co.ctc_decoder(io["x_a"], w, co.CONFORMER_S, dtype)
    assert np.abs(y - io["logits_a"]).max() < 2e-4           # fp32 graph vs fp64 restatement: 8.4e-5 observed
    assert (y.argmax(-1) == io["logits_a"].argmax(-1)).all()
    yb = co.ctc_decoder(io["x_b"], w, co.CONFORMER_S, dtype)
    assert (yb.argmax(-1) == io["argmax_b"]).all()
    assert np.abs(yb.max(-1) - io["max_b"]).max() < 2e-4
    assert np.abs(yb[:, ::8] - io["logits_b_every8"]).max() < 2e-4
    z = yb.astype(np.float64)
    lse = np.log(np.exp(z - z.max(-1, keepdims=True)).sum(-1)) + z.max(-1)
    assert np.abs(lse - io["lse_b"]).max() < 2e-4


def test_greedy_matches_reference_cpp_decoder_kats():
    kats = json.load(open(os.path.join(GOLDEN, "greedy_kat.json")))
    assert len(kats) >= 20
    for k in kats:
        if "probs" in k:
            p = np.array(k["probs"], np.float32)
            fa = p.argmax(-1)       # first max, as the strict '<' in ctc_greedy_decoder.h:13
        else:
            fa = np.array(k["frame_argmax"])
        ids, n = co.ctc_collapse(fa[None].astype(np.int32), [k["T"]], k["blank"])
        assert ids[0, :n[0]].tolist() == k["expect"]
        assert (ids[0, n[0]:] == -1).all()


def test_greedy_semantics_hand_cases():
    ids, n = co.ctc_collapse(np.array([[1, 1, 3, 1, 0, 0]], np.int32), [6], 3)
    assert ids[0, :n[0]].tolist() == [1, 1, 0]          # survey KAT
    ids, n = co.ctc_collapse(np.array([[1, 1, 3, 1, 0, 0]], np.int32), [2], 3)
    assert ids[0, :n[0]].tolist() == [1]                # input_length cuts the utterance
    ids, n = co.ctc_collapse(np.array([[3, 3, 3]], np.int32), [3], 3)
    assert n[0] == 0 and (ids == -1).all()              # all blank
    ids, n = co.ctc_collapse(np.zeros((1, 4), np.int32), [0], 3)
    assert n[0] == 0                                    # empty


def test_frame_argmax_is_argmax_of_log_softmax_first_max():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((3, 7, 11)).astype(np.float32)
    x[0, 0, 3] = x[0, 0, 8] = 9.0                       # exact tie -> lowest index
    a = co.frame_argmax(x)
    assert a[0, 0] == 3
    assert (a == x.argmax(-1)).all()


# ---- SAME padding (SURVEY 8a cheat sheet) -------------------------------------------------------------------
@pytest.mark.parametrize("n,k,s,expect", [
    (160000, 1024, 160, (1000, 432, 432)), (67263, 1024, 160, (421, 480, 481)),
    (1000, 3, 2, (500, 0, 1)), (80, 3, 2, (40, 0, 1)), (50, 3, 2, (25, 0, 1)), (25, 3, 2, (13, 1, 1)),
    (250, 32, 1, (250, 15, 16)), (13, 5, 1, (13, 2, 2)),
])
def test_same_padding_rule(n, k, s, expect):
    assert co.same_pad(n, k, s) == expect


# ---- independent cross-checks (these rows are also pinned by the reference's own code: tests/test_tf_goldens.py) ------
def test_power_spectrogram_equals_rfft_of_windowed_zero_padded_frames():
    rng = np.random.default_rng(1)
    for L in (16000, 4000, 8123):
        x = rng.standard_normal((2, L))
        re, im = co.stft_kernels(1024)
        p = co.power_spectrogram(x, re, im, 160)
        nf, lo, hi = co.same_pad(L, 1024, 160)
        xp = np.pad(x, ((0, 0), (lo, hi)))
        win = co.hann_periodic(1024).astype(np.float32).astype(np.float64)
        frames = np.stack([xp[:, f * 160:f * 160 + 1024] for f in range(nf)], 1) * win
        ref = np.abs(np.fft.rfft(frames, 1024, axis=-1)) ** 2
        assert p.shape == (2, nf, 513)
        assert np.abs(p - ref).max() < 1e-4 * ref.max()      # kernels are stored in fp32 (as in the reference)


def test_valid_mode_left_pads_n_dft_minus_1():
    x = np.random.default_rng(2).standard_normal((1, 2560))
    fr = co.frame_signal(x, 1024, 160, "valid")
    assert fr.shape[1] == (2560 + 1023 - 1024) // 160 + 1
    assert (fr[0, 0, :1023] == 0).all() and fr[0, 0, 1023] == x[0, 0]


def test_decibel_is_max_normalised_and_floored():
    p = np.array([[[1e-20, 1.0, 100.0, 1e-3]]])
    db = co.amplitude_to_decibel(p)
    assert np.allclose(db, [[[-80.0, -20.0, 0.0, -50.0]]])
    assert np.allclose(co.chunk_amplitude_to_decibel(p), [[[-10.0, 0.0, 2.0, -3.0]]])


def test_mel_filterbank_properties():
    fb = co.mel_filterbank(16000, 1024, 80, 0.0, 8000.0, False, 1)
    assert fb.shape == (80, 513) and fb.dtype == np.float32 and (fb >= 0).all()
    assert np.allclose(fb.sum(1), 1.0, atol=1e-5)                       # L1-normalised filters
    peaks = fb.argmax(1)
    assert (np.diff(peaks) > 0).all()                                   # monotone centre frequencies
    sl = co.mel_filterbank(16000, 1024, 80, 0.0, 8000.0, False, "slaney")
    assert np.allclose(sl / sl.sum(1, keepdims=True), fb, atol=1e-6)    # same shapes, different scaling
    # Slaney scale is linear below 1 kHz: first filters are equally spaced
    assert abs((peaks[2] - peaks[1]) - (peaks[1] - peaks[0])) <= 1


def test_mel_filters_are_narrow_bands_which_the_banded_kernel_relies_on():
    """mel_band_kernel (frontend.hip) is chosen when every column of freq2mel has its non-zeros inside one run of at most 64
    bins: true for backend.mel()'s triangles (the widest, at the top of the scale, spans 37 bins at 80 mels / 1024-point DFT),
    and the band form reproduces the dense product exactly as a sum over the band."""
    from tensorflowasr_amd import frontend_consts
    f2m = frontend_consts.freq2mel(16000, 1024, 80)                      # [513, 80], what ConformerEncoder._build loads
    assert np.array_equal(f2m, co.mel_filterbank(16000, 1024, 80, 0.0, 8000.0, False, 1).T)
    widths, x = [], np.random.default_rng(0).standard_normal((5, 513))
    banded = np.zeros((5, 80))
    for m in range(80):
        nz = np.flatnonzero(f2m[:, m])
        assert nz.size and np.array_equal(nz, np.arange(nz[0], nz[-1] + 1))     # one contiguous run, no holes
        widths.append(nz.size)
        banded[:, m] = x[:, nz[0]:nz[-1] + 1] @ f2m[nz[0]:nz[-1] + 1, m].astype(np.float64)
    assert max(widths) <= 64 and int(np.count_nonzero(f2m)) == sum(widths) == 1001
    assert np.allclose(banded, x @ f2m.astype(np.float64), rtol=0, atol=1e-12)


def test_plain_spectrogram_layer_is_the_input_of_the_mel_matrix():
    """Spectrogram.call (time_frequency.py:74-89) is what Melspectrogram.call multiplies by freq2mel (:173-181): the oracle's
    two frontends must agree on that, and the Spectrogram encoder consumes 513 bins (F2 = 129)."""
    w = co.encoder_weights(dict(co.CONFORMER_S, num_blocks=0), seed=1)
    x = np.random.default_rng(2).standard_normal((2, 4000))
    sp = co.spectrogram(x, w)
    assert sp.shape == (2, 25, 513) and sp.max() == 0.0 and sp.min() >= -80.0
    assert np.allclose(co.melspectrogram(x, w), sp @ w["mel_layer/freq2mel"].astype(np.float64))
    ws = co.encoder_weights(dict(co.CONFORMER_S, num_blocks=0, mel_layer_type="Spectrogram"), seed=1)
    assert "mel_layer/freq2mel" not in ws and ws["conv_subsampling/linear/kernel"].shape == (129 * 144, 144)


def test_conv2d_same_matches_torch():
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(3)
    for (H, W, C, O, s) in [(50, 80, 1, 8, (2, 2)), (25, 40, 8, 8, (2, 2)), (13, 7, 4, 6, (2, 2))]:
        x = rng.standard_normal((2, H, W, C))
        k = rng.standard_normal((3, 3, C, O))
        b = rng.standard_normal(O)
        y = co.conv2d_same(x, k, b, s)
        oh, pt, pb = co.same_pad(H, 3, s[0])
        ow, pl, pr = co.same_pad(W, 3, s[1])
        xt = F.pad(torch.from_numpy(x).permute(0, 3, 1, 2), (pl, pr, pt, pb))
        ref = F.conv2d(xt, torch.from_numpy(k).permute(3, 2, 0, 1), torch.from_numpy(b), stride=s)
        assert np.abs(y - ref.permute(0, 2, 3, 1).numpy()).max() < 1e-10


def test_layernorm_depthwise_mha_against_torch():
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(4)
    x = rng.standard_normal((2, 17, 24))
    g, b = rng.standard_normal(24), rng.standard_normal(24)
    ref = F.layer_norm(torch.from_numpy(x), (24,), torch.from_numpy(g), torch.from_numpy(b), eps=1e-3).numpy()
    assert np.abs(co.layer_norm(x, g, b) - ref).max() < 1e-12
    # depthwise k=32: 15 left / 16 right
    dk = rng.standard_normal((32, 24, 1))
    y = co.depthwise_conv1d_same(x, dk)
    xt = F.pad(torch.from_numpy(x).permute(0, 2, 1), (15, 16))
    ref = F.conv1d(xt, torch.from_numpy(dk[:, :, 0].T.copy())[:, None, :], groups=24).permute(0, 2, 1).numpy()
    assert np.abs(y - ref).max() < 1e-12
    # MHA vs torch scaled_dot_product_attention
    H, hs, d = 4, 6, 24
    w = {"m/query_kernel": rng.standard_normal((H, d, hs)), "m/key_kernel": rng.standard_normal((H, d, hs)),
         "m/value_kernel": rng.standard_normal((H, d, hs)), "m/projection_kernel": rng.standard_normal((H, hs, d)),
         "m/projection_bias": rng.standard_normal(d)}
    out = co.mha(x, x, w, "m", hs)
    q = torch.einsum("bni,hio->bhno", torch.from_numpy(x), torch.from_numpy(w["m/query_kernel"]))
    k = torch.einsum("bni,hio->bhno", torch.from_numpy(x), torch.from_numpy(w["m/key_kernel"]))
    v = torch.einsum("bni,hio->bhno", torch.from_numpy(x), torch.from_numpy(w["m/value_kernel"]))
    o = F.scaled_dot_product_attention(q, k, v)
    ref = torch.einsum("bhni,hio->bno", o, torch.from_numpy(w["m/projection_kernel"])).numpy() + w["m/projection_bias"]
    assert np.abs(out - ref).max() < 1e-10


def test_streaming_encoder_is_blockwise_encoder():
    cfg = dict(co.STREAMING_S, num_blocks=1)
    w = co.encoder_weights(cfg, seed=5)
    x = np.stack([co.synth_wave(i, 16000) for i in range(2)])
    y = co.streaming_conformer_encoder(x, w, cfg, 8000)
    assert y.shape == (2, 26, 256)
    y0 = co.conformer_encoder(x[:, :8000], w, cfg)
    assert np.abs(y[:, :13] - y0).max() < 1e-12
    with pytest.raises(AssertionError):
        co.streaming_conformer_encoder(x[:, :12000], w, cfg, 8000)


def test_fp32_and_fp64_oracle_agree_end_to_end():
    cfg = dict(co.CONFORMER_S, num_blocks=2)
    w = co.encoder_weights(cfg, seed=0)
    x = np.stack([co.synth_wave(i, 16000) for i in range(2)])
    y64 = co.conformer_encoder(x, w, cfg, np.float64)
    y32 = co.conformer_encoder(x, w, cfg, np.float32)
    assert y64.shape == (2, 25, 144)
    assert np.abs(y64 - y32).max() < 1e-3


# ---- prefix beam search (row a14): pinned against the reference's own decoder ----------------------------------
def _beam_kats():
    k = np.load(os.path.join(GOLDEN, "beam_kat.npz"))
    meta = json.loads(str(k["meta"]))
    return k, meta


def test_beam_oracle_matches_reference_decoder_kats():
    from oracle import ctc_beam_oracle as bo
    k, meta = _beam_kats()
    assert len(meta) >= 10
    for i, m in enumerate(meta):
        if m["T"] * m["V"] > 30000:
            continue                                  # pure-Python restatement: small cases only
        res = bo.ctc_beam_search(k["probs_%d" % i], m["beam"], m["cutoff_prob"], m["cutoff_top_n"])
        assert len(res) == m["n"]
        for j, (score, ids) in enumerate(res):
            L = int(k["lens_%d" % i][j])
            assert ids == k["ids_%d" % i][j, :L].tolist()
            assert abs(float(score) - k["scores_%d" % i][j]) < 1e-4


def test_beam_pruning_quirk_cutoff_prob_one_disables_top_n():
    from oracle import ctc_beam_oracle as bo
    p = np.array([0.5, 0.2, 0.15, 0.1, 0.05])
    assert len(bo.pruned_log_probs(p, 1.0, 2)) == 5           # decoder_utils.cpp:18-31: top-n ignored at cutoff 1.0
    assert [c for c, _ in bo.pruned_log_probs(p, 0.99, 2)] == [0, 1]
    assert [c for c, _ in bo.pruned_log_probs(p, 0.6, 40)] == [0, 1]


def test_positional_encoding_known_values_and_translator_shapes():
    """positional_encoding.py:19-36: even columns sin, odd columns cos of pos / 10000^(2*(i//2)/size)."""
    pe = co.positional_encoding(6, 8)
    assert pe.shape == (6, 8)
    assert np.allclose(pe[0], [0, 1, 0, 1, 0, 1, 0, 1])
    assert np.isclose(pe[1, 0], np.sin(1.0)) and np.isclose(pe[1, 1], np.cos(1.0))
    assert np.isclose(pe[3, 2], np.sin(3.0 / 10.0)) and np.isclose(pe[3, 3], np.cos(3.0 / 10.0))
    assert np.isclose(pe[5, 6], np.sin(5.0 / 1000.0)) and np.isclose(pe[5, 7], np.cos(5.0 / 1000.0))
    cfg = dict(co.CONFORMER_S, translator_num_blocks=1, translator_kernel_size=32, translator_fc_factor=0.5)
    w = co.translator_weights(cfg, 50, 60, seed=5)
    rng = np.random.default_rng(0)
    ids = rng.integers(0, 50, (2, 9))
    enc = rng.standard_normal((2, 20, 144))
    y = co.translator(ids, enc, w, cfg)
    assert y.shape == (2, 9, 60) and np.isfinite(y).all()
    # the positional term matters (same token at two positions gives different outputs) ...
    same = np.full((1, 4), 7)
    ys = co.translator(same, enc[:1], w, cfg)
    assert np.abs(ys[0, 0] - ys[0, 3]).max() > 1e-3
    # ... and the encoder output is only seen through the cross-attention (permuting its frames changes nothing)
    yp = co.translator(ids[:1], enc[:1, ::-1], w, cfg)
    assert np.abs(yp - y[:1]).max() < 1e-9


def test_chunk_streaming_restatement_equals_offline_predict():
    """The cache design of chunk_conformer_blocks.py:799-866: feeding chunk_num * hop = 2560 samples per call with
    the caches carried over reproduces the offline predict() -- picker logits on every frame, text logits on every
    frame that has its win_back = 8 frames of right context."""
    cfg = dict(co.CHUNK_S, enc_num_blocks=1, picker_num_classes=12, decoder_num_classes=15)
    w = co.chunk_weights(cfg, seed=4)
    n = 22
    x = co.synth_wave(2, length=2560 * n)[None]
    z = co.chunk_predict(x, w, cfg)["picker_logits"]
    gap = np.sort(z[..., :-1].max(-1) - z[..., -1], axis=None)
    w["picker/fully_connected/bias"][-1] = 0.5 * (gap[gap.size // 3] + gap[gap.size // 3 + 1])   # ~2/3 of the frames kept
    off = co.chunk_predict(x, w, cfg)
    pc, dc = co.chunk_init_picker_caches(cfg), co.chunk_init_decoder_caches(cfg)
    ph, txt, unv = [], [], None
    for i in range(n):
        vp, _, vh, pc = co.chunk_picker_stream_predict(x[:, i * 2560:(i + 1) * 2560], pc, w, cfg)
        assert vp.shape[1] == 4                                   # 16 mel frames -> 4 encoder frames per call
        ph.append(vp)
        f, _ = co.feature_pick(vh, vp, cfg["picker_num_classes"] - 1)
        if f.shape[1]:
            vt, unv, dc = co.chunk_decoder_stream_predict(f, dc, w, cfg)
            txt.append(vt)
    assert np.abs(np.concatenate(ph, 1) - off["picker_logits"]).max() < 1e-10
    txt = np.concatenate(txt, 1)
    assert 0 < off["counts"][0] < 4 * n                           # the picker drops some frames
    assert txt.shape[1] == off["text_logits"].shape[1] - 8
    assert np.abs(txt - off["text_logits"][:, :txt.shape[1]]).max() < 1e-10
    assert np.abs(unv - off["text_logits"][:, -8:]).max() < 1e-10
    # caches are cut to the attention window / conv kernel / carry-over lengths
    assert pc["enc_mha"][0].shape[1] == 36 and pc["enc_cnn"][0].shape[1] == 32 and pc["front_wav"].shape[1] == 2560
    assert dc["dec_inp"].shape[1] == 8 and pc["dec_inp"].shape[1] == 0


def test_bf16_rounding_emulation():
    """bf16_round = round-to-nearest-even on the top 16 bits (what v_cvt_pk_bf16_f32 does)."""
    x = np.array([1.0, 1.00390625, 1.01171875, -3.14159274, 65504.0, 1e-30, 0.0], np.float32)
    r = co.bf16_round(x)
    assert r.dtype == np.float32 and (r.view(np.uint32) & 0xFFFF == 0).all()
    assert r[0] == 1.0 and r[1] == 1.0 and r[2] == 1.015625        # tie -> even ; above the tie -> up
    assert abs(r[3] + 3.140625) < 1e-7
    assert np.all(np.abs(r[:5] - x[:5]) <= np.abs(x[:5]) * 2.0 ** -8)
    cfg = dict(co.CONFORMER_S)
    w = co.block_weights(np.random.default_rng(0), "b", 144, 4, 36, 32)
    xin = np.random.default_rng(1).standard_normal((1, 20, 144))
    exact = co.conformer_block(xin, w, "b", 36)
    co.GEMM_ROUND_BF16 = True
    try:
        rounded = co.conformer_block(xin, w, "b", 36)
    finally:
        co.GEMM_ROUND_BF16 = False
    d = np.abs(rounded - exact).max()
    assert 1e-4 < d < 0.1                                           # visibly bf16, still the same function
    assert np.abs(co.conformer_block(xin, w, "b", 36) - exact).max() == 0.0


def test_wave_pick_model_against_torch_convs():
    """add_wav_info branch (wav_model.py:108-146): strides from hop_size, and the oracle's conv stack against an
    independent torch.nn.functional restatement with Keras SAME padding."""
    import torch
    import torch.nn.functional as F
    assert co.wave_pick_scales(640) == [8, 5, 4, 4]
    assert co.wave_pick_scales(160 * 4 * 2) == [16, 5, 4, 4]      # 2^8 * 5: pairs of the smallest factors merge first
    assert int(np.prod(co.wave_pick_scales(480))) == 480
    d, hop = 144, 640
    w = co.wave_pick_weights(d, hop, seed=5)
    x = np.stack([co.synth_wave(i, 640 * 6) for i in range(2)])
    got = co.wave_pick_model(x, w, d, hop)
    assert got.shape == (2, 6, d)

    def same(t, k, s):                                  # Keras/TF SAME: total = max((ceil(n/s)-1)*s + k - n, 0), left = total // 2
        n = t.shape[-1]
        tot = max((-(-n // s) - 1) * s + k - n, 0)
        return F.pad(t, (tot // 2, tot - tot // 2))

    def conv(t, name, s=1, pad=True):
        k = torch.from_numpy(w[name + "/kernel"].astype(np.float64)).permute(2, 1, 0)      # [k, cin, cout] -> [cout, cin, k]
        b = torch.from_numpy(w[name + "/bias"].astype(np.float64))
        return F.conv1d(same(t, k.shape[-1], s) if pad else t, k, b, stride=s)

    t = torch.from_numpy(x.astype(np.float64))[:, None, :]
    dw = torch.from_numpy(w["wav_layer/sep_conv/depthwise_kernel"].astype(np.float64)).reshape(1, 1, 7)
    t = F.conv1d(same(t, 7, 8), dw, stride=8)
    t = t * torch.from_numpy(w["wav_layer/sep_conv/pointwise_kernel"].astype(np.float64)).reshape(1, 32, 1) \
        + torch.from_numpy(w["wav_layer/sep_conv/bias"].astype(np.float64)).reshape(1, 32, 1)
    t = F.leaky_relu(t, 0.3)
    for i, s in zip((1, 2, 3), (5, 4, 4)):
        t = conv(t, "wav_layer/conv_%d" % i, s)
        a = F.pad(F.leaky_relu(t, 0.3), (2, 2), mode="reflect")
        a = conv(a, "wav_layer/res_%d/conv5" % i, pad=False)
        a = conv(F.leaky_relu(a, 0.3), "wav_layer/res_%d/conv1" % i, pad=False)
        t = conv(t, "wav_layer/res_%d/shortcut" % i, pad=False) + a
    t = conv(t, "wav_layer/final")
    assert np.abs(t.permute(0, 2, 1).numpy() - got).max() < 1e-10


def _tf_greedy_argmax_fp32(row):
    """per-frame class of the reference's greedy path in its own arithmetic: tf.nn.softmax (fp32) -> ctc_decode's
    log(p + 1e-7) (fp32) -> argmax with strict '<' (test_asr.py:196-198, ctc_greedy_decoder.h:9-20)"""
    x = np.asarray(row, np.float32)
    e = np.exp(x - x.max(), dtype=np.float32)
    p = (e / e.sum(dtype=np.float32)).astype(np.float32)
    return int(np.argmax(np.log(p + np.float32(1e-7), dtype=np.float32)))


def sub_ulp_tie_row(V=8):
    """two top logits one ulp apart (0.25 and the next float above it), the LARGER one at the higher class index: exp()
    maps both onto the same fp32 value, so the reference's log(softmax + 1e-7) ties and its strict '<' keeps the lower index,
    while an argmax on the logits themselves keeps the higher one"""
    row = np.full(V, -3.0, np.float32)
    row[2] = 0.25
    row[5] = np.nextafter(np.float32(0.25), np.float32(1.0))
    return row


def test_documented_deviation_argmax_on_logits_vs_log_softmax_on_sub_ulp_ties():
    """the head takes the per-frame argmax on the logits.  It differs from the reference's
    log(softmax + 1e-7) argmax only when two logits are closer than fp32 exp() resolves (|x| < 0.5: one ulp); this test
    constructs that case so that the difference is a known, pinned one -- and shows that a gap exp() does resolve
    (eight ulps) gives the same class on both routes."""
    row = sub_ulp_tie_row()
    assert row[5] > row[2] and row[5] - row[2] < 3e-8
    assert _tf_greedy_argmax_fp32(row) == 2            # the reference: tie after exp -> first maximum
    assert int(co.frame_argmax(row[None, None])[0, 0]) == 2     # the oracle restates the reference's formula
    assert int(np.argmax(row)) == 5                    # the device kernels (head epilogue, mi355asr_frame_argmax): the larger
    wide = row.copy()                                  # logit -- asserted on the GPU in tests/test_gpu_parity.py
    for _ in range(8):
        wide[5] = np.nextafter(wide[5], np.float32(1.0))
    assert _tf_greedy_argmax_fp32(wide) == 5 == int(co.frame_argmax(wide[None, None])[0, 0]) == int(np.argmax(wide))


def test_config2_b64_fixture_is_what_its_script_produces():
    """tests/golden/config2_oracle_b64.npz (all 64 benched utterances through the fp64 oracle) -- one utterance per head is
    recomputed with tests/golden/make_config2_b64.py's own functions and must reproduce the stored rows."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("make_config2_b64", os.path.join(root, "tests", "golden", "make_config2_b64.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    fx = np.load(g.OUT)
    assert fx["trained_top4_idx"].shape == (64, 250, 4) and int(fx["tokens_lens"].sum()) > 10000
    g._W["trained"] = g.bench_weights()
    u = 37
    head, uu, enc10, idx, val, lg50 = g.one(("trained", u))
    assert np.array_equal(idx, fx["trained_top4_idx"][u]) and np.array_equal(val, fx["trained_top4_val"][u])
    assert np.array_equal(enc10, fx["trained_enc_every10"][u]) and np.array_equal(lg50, fx["trained_logits_every50"][u])
    ids, lens = co.ctc_collapse(fx["tokens_top4_idx"][..., 0].astype(np.int32), [250] * 64, 1331)
    assert np.array_equal(ids, fx["tokens_ids"]) and np.array_equal(lens, fx["tokens_lens"])

