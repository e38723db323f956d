"""Parity against fixtures produced by the reference's own Python (tests/golden/make_tf_goldens.py).  Round 5: the fixtures
exist -- asr/models/{conformer_blocks,chunk_conformer_blocks,wav_model}.py, asr/models/layers/*.py and leaf_audio/*.py were
imported UNMODIFIED from /root/reference and executed on the NumPy stand-in for TensorFlow / librosa under oracle/_tfshim
(no TensorFlow can be installed here; the stand-in is gated by tests/test_tfshim.py).  Each fixture holds two runs of the
same reference code: `<key>` computed in float32 (what a TensorFlow CPU forward does, up to summation order) and
`<key>_f64` with float32 tensors carried in float64 (constants still rounded where the reference rounds them).

* CPU: the NumPy oracle must reproduce `<key>` within the contract's 1e-3 AND `<key>_f64` within 1e-9 (relative to the
  tensor's scale): every reshape order, padding rule, per-block dB maximum, cache slice, band mask and compaction of the
  reference's composition is then pinned digit for digit, not within a tolerance that could hide a misplaced frame.
* GPU (`-m gpu`): libmi355asr.so against `<key>` within 1e-3, ids equal to the reference's ctc_decode.
SURVEY rows a2-a5, a10, a11, a15, 8f-1 and 8f-4 are pinned by these fixtures."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, chunk_config_dict, co, encoder_kwargs, maxdiff, small_cfg, waves

TOL = 1e-3
EXACT = 1e-9          # oracle (float64) vs the reference's code carried in float64: relative to max |reference|


def exact(got, fx, key, tol=EXACT):
    ref = fx[key + "_f64"].reshape(np.shape(got))
    err = maxdiff(got, ref) / max(1.0, float(np.abs(ref).max()))
    assert err < tol, "%s: oracle differs from the reference's float64 run by %.3g (relative)" % (key, err)


def fixture(name):
    path = os.path.join(GOLDEN, name)
    if not os.path.exists(path):
        pytest.skip("%s not generated yet (python tests/golden/make_tf_goldens.py on a TensorFlow box)" % name)
    return np.load(path, allow_pickle=False)


def with_reference_mel(w, fx, prefix="mel_layer/"):
    """the filterbank as the reference's librosa built it (version dependent); ours must agree with it to 1e-6"""
    w = dict(w)
    if "freq2mel" in fx.files and fx["freq2mel"].ndim == 2:
        assert np.abs(w[prefix + "freq2mel"] - fx["freq2mel"]).max() < 1e-6, "default mel matrix differs from the reference's librosa"
        w[prefix + "freq2mel"] = fx["freq2mel"]
    return w


# ---- CPU: the oracle against the reference ---------------------------------------------------------------------------
@pytest.mark.parametrize("L", [32000, 67263])
def test_oracle_mel_vs_tf(L):
    fx = fixture("tf_mel_L%d.npz" % L)
    w = with_reference_mel(co.encoder_weights(small_cfg(1), seed=0), fx)
    bins = [0, 1, 37, 256, 512]
    assert np.abs(w["mel_layer/real_kernels"].reshape(1024, -1)[:, bins] - fx["real_kernels_bins"]).max() < 1e-6
    assert np.abs(w["mel_layer/imag_kernels"].reshape(1024, -1)[:, bins] - fx["imag_kernels_bins"]).max() < 1e-6
    x = waves(2, L, int(fx["wave_seed"]))
    got = co.melspectrogram(x.astype(np.float64), w)
    assert maxdiff(got, fx["mel"]) < TOL
    exact(got, fx, "mel")


def test_oracle_conv_subsampling_vs_tf():
    fx = fixture("tf_conv_subsampling.npz")
    w = co.encoder_weights(small_cfg(2), seed=int(fx["weights_seed"]))
    mel = (-80.0 * np.random.default_rng(int(fx["mel_seed"])).random((3, 200, 80, 1))).astype(np.float32)[..., 0]
    got = co.conv_subsampling(mel.astype(np.float64), w)
    assert maxdiff(got, fx["out"]) < TOL
    exact(got, fx, "out")


def test_oracle_encoder_ctc_vs_tf():
    fx = fixture("tf_encoder_ctc.npz")
    cfg = small_cfg(2)
    w = with_reference_mel(co.encoder_weights(cfg, seed=int(fx["enc_weights_seed"])), fx)
    V = int(fx["num_classes"])
    w.update(co.ctc_decoder_weights(cfg, V, seed=int(fx["ctc_weights_seed"])))
    x = waves(2, int(fx["L"]), int(fx["wave_seed"]))
    enc = co.conformer_encoder(x.astype(np.float64), w, cfg)
    lg = co.ctc_decoder(enc, w, cfg)
    assert maxdiff(enc, fx["enc"]) < TOL and maxdiff(lg, fx["logits"]) < TOL
    exact(enc, fx, "enc")
    exact(lg, fx, "logits")
    ids, lens = co.ctc_greedy(fx["logits"], [lg.shape[1]] * 2, V - 1)        # the decode rule on the reference's own logits
    ref = fx["ctc_decode"]
    for b in range(2):
        assert ids[b, :lens[b]].tolist() == [int(t) for t in ref[b] if t >= 0]


def test_oracle_streaming_encoder_vs_tf():
    fx = fixture("tf_streaming_encoder.npz")
    cfg = small_cfg(2, co.STREAMING_S)
    w = with_reference_mel(co.encoder_weights(cfg, seed=int(fx["weights_seed"])), fx)
    x = waves(2, int(fx["L"]), int(fx["wave_seed"]))
    got = co.streaming_conformer_encoder(x.astype(np.float64), w, cfg, 8000)
    assert maxdiff(got, fx["enc"]) < TOL
    exact(got, fx, "enc")


def _translator_case(fx):
    cfg = dict(co.CONFORMER_S, translator_num_blocks=2, translator_kernel_size=32, translator_fc_factor=0.5)
    w = co.translator_weights(cfg, 60, 80, seed=int(fx["weights_seed"]))
    rng = np.random.default_rng(int(fx["seed"]))
    ids = rng.integers(0, 60, (3, 40)).astype(np.int32)
    enc = rng.standard_normal((3, 250, 144)).astype(np.float32)
    return cfg, w, ids, enc


def test_oracle_translator_vs_tf():
    fx = fixture("tf_translator.npz")
    cfg, w, ids, enc = _translator_case(fx)
    got = co.translator(ids, enc.astype(np.float64), w, cfg)
    assert maxdiff(got, fx["logits"]) < TOL
    exact(got, fx, "logits")


def test_oracle_wave_pick_vs_tf():
    fx = fixture("tf_wave_pick.npz")
    w = co.wave_pick_weights(144, 640, seed=int(fx["weights_seed"]))
    x = waves(2, int(fx["L"]), int(fx["wave_seed"]))
    got = co.wave_pick_model(x.astype(np.float64), w, 144, 640)
    assert maxdiff(got, fx["out"]) < TOL
    exact(got, fx, "out")


def _leaf_weights_from(fx, suffix=""):
    """the LEAF variables under the names the reference's Keras objects gave them, through the product's own name map.
    suffix "_f64": the values the stand-in's wide pass ran on (its own float32-rounded initial values; GaborInit computed in
    float64 and in float32 differ in the last float32 digit, and PCEN amplifies that to 1e-5)"""
    from tensorflowasr_amd import checkpoint
    names = {k[:len(k) - len(suffix)].replace("|", "/"): k for k in fx.files if k.endswith(":0" + suffix)}
    m = checkpoint.keras_names_to_abi(list(names))
    assert len(m) == len(names) == 9, sorted(set(names) - set(m))
    return {m[n]: fx[k] for n, k in names.items()}


def test_oracle_leaf_vs_tf():
    fx = fixture("tf_leaf.npz")
    w = _leaf_weights_from(fx)
    assert set(co.leaf_default_weights()) <= set(w), sorted(w)
    for k, v in co.leaf_default_weights().items():              # the oracle's restated initial values == the reference's
        assert maxdiff(v, w[k]) < 1e-6 * max(1.0, np.abs(w[k]).max()), k
    x = waves(2, int(fx["L"]), int(fx["wave_seed"]))
    assert maxdiff(co.leaf_frontend(x.astype(np.float64), w), fx["out"].reshape(2, -1, 80)) < TOL
    exact(co.leaf_frontend(x.astype(np.float64), _leaf_weights_from(fx, "_f64")), fx, "out")


def _chunk_case(fx, L=None):
    cfg = dict(co.CHUNK_S, enc_num_blocks=2, picker_num_classes=30, decoder_num_classes=40)
    w = co.chunk_weights(cfg, seed=int(fx["weights_seed"]))
    if "freq2mel" in fx.files and fx["freq2mel"].ndim == 2:
        w["front/mel_layer/freq2mel"] = fx["freq2mel"]
    w["picker/fully_connected/bias"][-1] = fx["picker_blank_bias"]
    return cfg, w, waves(2, int(fx["L"]) if L is None else L, int(fx["wave_seed"]))


def test_oracle_chunk_predict_vs_tf():
    """6 s utterances: 150 frames, the band mask cuts (win_front 36); about half of the frames are picked, ragged per utterance:
    the oracle's feature_pick against the reference's tf.while_loop compaction (zero padding to the batch maximum included)"""
    fx = fixture("tf_chunk_predict.npz")
    cfg, w, x = _chunk_case(fx)
    r = co.chunk_predict(x.astype(np.float64), w, cfg)
    assert fx["enc"].shape[1] == 150 and 20 < fx["picked"].shape[1] < 130 and r["counts"].min() < r["counts"].max()
    for k in ("front", "enc", "picker_logits", "picker_hidden", "picked", "text_logits"):
        assert r[k].shape == fx[k].shape and maxdiff(r[k], fx[k]) < TOL, k
        exact(r[k], fx, k)


def test_oracle_chunk_streaming_vs_tf():
    """The reference's streaming entry points with explicit caches, 30 steps of 2560 samples: the oracle's restatement of the
    stream_call chain (what the GPU streaming test compares libmi355asr.so with) against the reference's own -- every valid
    output, which steps produced text, the look-ahead rows and the caches after the last step, digit for digit."""
    from helpers import stream_oracle
    fx = fixture("tf_chunk_stream.npz")
    cfg, w, x = _chunk_case(fx, L=96000)
    n, samples = int(fx["nchunks"]), int(fx["samples"])
    ph, hid, txt, unv, pc, dc, steps = stream_oracle(x[:1, :n * samples].astype(np.float64), w, cfg, n, samples)
    assert [list(s) for s in steps] == fx["steps"].tolist() and len(steps) >= 8
    for k, got in (("picker_logits", ph), ("picker_hidden", hid), ("text_logits", txt), ("unvalid_text_logits", unv),
                   ("cache_front_wav", pc["front_wav"][..., None]), ("cache_front_sub", pc["front_sub"][..., None]),
                   ("cache_enc_mha", np.stack(pc["enc_mha"])), ("cache_enc_cnn", np.stack(pc["enc_cnn"])),
                   ("cache_picker_mha", np.stack(pc["picker_mha"])), ("cache_picker_cnn", np.stack(pc["picker_cnn"])),
                   ("cache_picker_dec_inp", pc["dec_inp"]), ("cache_helper_mha", np.stack(dc["helper_mha"])),
                   ("cache_decoder_mha", np.stack(dc["decoder_mha"])), ("cache_decoder_cnn", np.stack(dc["decoder_cnn"])),
                   ("cache_decoder_dec_inp", dc["dec_inp"])):
        assert got.shape == fx[k].shape, (k, got.shape, fx[k].shape)
        if got.size:
            assert maxdiff(got, fx[k]) < TOL, k
            exact(got, fx, k)


def test_oracle_benched_batch_fixture_vs_the_reference_code_on_all_64_utterances():
    """tests/golden/tf_config2_b64.npz: the 64 benched utterances (BASELINE config 2) through the reference's own ConformerEncoder +
    CTCDecoder + ctc_decode, both weight sets of config2_oracle_b64.npz.  The fp64 oracle's fixture must agree with it: greedy
    ids and lengths of all 64 utterances for both heads, every frame's arg-max (the reference's float64 run; a frame may differ only
    where its own top-2 margin is below 1e-9: none does), the four largest logits of every frame within 1e-5 (the oracle fixture
    stores float32), and the reference's float32 run within 1e-3 of them with arg-max flips only inside its own rounding."""
    fx, orc = fixture("tf_config2_b64.npz"), fixture("config2_oracle_b64.npz")
    for head in ("trained", "tokens"):
        ref_idx, ref_val = fx[head + "_top4_idx_f64"].astype(np.int64), fx[head + "_top4_val_f64"]
        o_idx, o_val = orc[head + "_top4_idx"].astype(np.int64), orc[head + "_top4_val"]
        assert ref_idx.shape == o_idx.shape == (64, 250, 4)
        margin = ref_val[..., 0] - ref_val[..., 1]
        flips = np.argwhere(ref_idx[..., 0] != o_idx[..., 0])
        assert all(margin[b, t] < 1e-9 for b, t in flips), [(int(b), int(t), float(margin[b, t])) for b, t in flips[:5]]
        same = ref_idx == o_idx                                   # compare values class by class where the order agrees (ties may swap)
        assert same.mean() > 0.999
        assert float(np.abs(ref_val - o_val)[same].max()) < 1e-5
        # the float32 run of the reference's code against its float64 run: what a TensorFlow forward is allowed to differ by
        v32, i32 = fx[head + "_top4_val"], fx[head + "_top4_idx"].astype(np.int64)
        agree = i32[..., 0] == ref_idx[..., 0]
        assert float(np.abs(v32 - ref_val)[i32 == ref_idx].max()) < TOL
        assert all(margin[b, t] < 1e-3 for b, t in np.argwhere(~agree)), head
        # ids: the reference's ctc_decode (float32 run) == the oracle's collapse wherever the two arg-max sequences agree
        ok = agree.all(axis=1)
        assert ok.sum() >= 60, (head, int(ok.sum()))
        assert np.array_equal(fx[head + "_ids"][ok], orc[head + "_ids"][ok]) and np.array_equal(fx[head + "_lens"][ok], orc[head + "_lens"][ok])
        assert maxdiff(fx[head + "_enc_every10"], orc[head + "_enc_every10"]) < 1e-4
    assert int(fx["tokens_lens"].sum()) > 15000 and int(fx["trained_lens"].sum()) == int(orc["trained_lens"].sum())


@pytest.mark.skipif(not os.path.isdir("/root/reference/asr/models"), reason="needs the reference checkout")
def test_benched_batch_fixture_is_what_the_recipe_makes_for_two_utterances(tmp_path):
    """the committed tf_config2_b64.npz against a fresh run of its recipe on utterances 0 and 63 (a few seconds per head)"""
    import subprocess
    import sys
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import make_tf_config2_b64 as m; "
            "r = m.run(utterances=[0, 63]); np.savez(sys.argv[1], **r)" % GOLDEN)
    out = str(tmp_path / "two.npz")
    r = subprocess.run([sys.executable, "-c", code, out], capture_output=True, text=True, timeout=900, cwd=os.path.dirname(GOLDEN))
    assert r.returncode == 0, r.stderr[-3000:]
    two, fx = np.load(out), fixture("tf_config2_b64.npz")
    for k in two.files:
        a, b = two[k], fx[k][[0, 63]]
        if a.dtype.kind == "f":
            assert maxdiff(a, b) <= 2e-6 * max(1.0, float(np.abs(b).max())), k
        else:
            assert np.array_equal(a, b), k


# ---- GPU: libmi355asr.so against the reference -------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("L", [32000, 67263])
def test_gpu_mel_vs_tf(L):
    from tensorflowasr_amd.models import ConformerEncoder
    fx = fixture("tf_mel_L%d.npz" % L)
    cfg = small_cfg(1)
    w = with_reference_mel(co.encoder_weights(cfg, seed=0), fx)
    e = ConformerEncoder(**encoder_kwargs(cfg))
    e.load_weights(w, by_name=False)
    assert maxdiff(e.melspectrogram(waves(2, L, int(fx["wave_seed"]))).cpu().numpy(), fx["mel"]) < TOL


@pytest.mark.gpu
def test_gpu_encoder_ctc_vs_tf():
    from tensorflowasr_amd.models import ConformerCTC
    fx = fixture("tf_encoder_ctc.npz")
    cfg = small_cfg(2)
    V = int(fx["num_classes"])
    w = with_reference_mel(co.encoder_weights(cfg, seed=int(fx["enc_weights_seed"])), fx)
    w.update(co.ctc_decoder_weights(cfg, V, seed=int(fx["ctc_weights_seed"])))
    m = ConformerCTC(V, **{k: v for k, v in encoder_kwargs(cfg).items() if k != "mel_layer_type"})
    m.load_weights(w, by_name=False)
    x = waves(2, int(fx["L"]), int(fx["wave_seed"]))
    enc = m.encode(x)
    assert maxdiff(enc.cpu().numpy(), fx["enc"]) < TOL
    assert maxdiff(m.ctc_logits(enc).cpu().numpy(), fx["logits"]) < TOL
    ids, lens = m.recognize(x)
    ids, lens = ids.cpu().numpy(), lens.cpu().numpy()
    for b in range(2):
        assert ids[b, :lens[b]].tolist() == [int(t) for t in fx["ctc_decode"][b] if t >= 0]
    sub = fixture("tf_conv_subsampling.npz")
    mel = (-80.0 * np.random.default_rng(int(sub["mel_seed"])).random((3, 200, 80, 1))).astype(np.float32)[..., 0]
    assert maxdiff(m.conv_subsampling(mel).cpu().numpy(), sub["out"]) < TOL


@pytest.mark.gpu
def test_gpu_streaming_translator_wavepick_leaf_chunk_vs_tf():
    from tensorflowasr_amd.models import ChunkConformer, ConformerEncoder, StreamingConformerEncoder, Translator
    fx = fixture("tf_streaming_encoder.npz")
    cfg = small_cfg(2, co.STREAMING_S)
    e = StreamingConformerEncoder(**encoder_kwargs(cfg))
    e.add_chunk_size(8000, 80, 640)
    e.load_weights(co.encoder_weights(cfg, seed=int(fx["weights_seed"])), by_name=False)
    assert maxdiff(e(waves(2, int(fx["L"]), int(fx["wave_seed"]))).cpu().numpy(), fx["enc"]) < TOL
    fx = fixture("tf_translator.npz")
    tcfg, w, ids, enc = _translator_case(fx)
    tr = Translator(inp_classes=60, tar_classes=80, dmodel=144, num_blocks=2, head_size=36, num_heads=4, kernel_size=32)
    tr.load_weights(w, by_name=False)
    assert maxdiff(tr([ids, enc]).cpu().numpy(), fx["logits"]) < TOL
    fx = fixture("tf_wave_pick.npz")
    cfg0 = small_cfg(0)
    w0 = {k: v for k, v in co.encoder_weights(cfg0, seed=21).items() if not k.startswith("conformer_block_")}
    ww = co.wave_pick_weights(144, 640, seed=int(fx["weights_seed"]))
    x = waves(2, int(fx["L"]), int(fx["wave_seed"]))
    ea = ConformerEncoder(**dict(encoder_kwargs(cfg0), add_wav_info=True))
    ea.load_weights(dict(w0, **ww), by_name=False)
    eb = ConformerEncoder(**encoder_kwargs(cfg0))
    eb.load_weights(w0, by_name=False)
    assert maxdiff(ea(x).cpu().numpy().astype(np.float64) - eb(x).cpu().numpy(), fx["out"]) < TOL
    fx = fixture("tf_chunk_predict.npz")
    ccfg, wc, xc = _chunk_case(fx)
    m = ChunkConformer(chunk_config_dict(ccfg), 30, 40)
    m.load_weights(wc, by_name=False)
    got = m.predict(xc, stages=True)
    for k in ("front", "enc", "picker_logits", "picker_hidden", "picked", "text_logits"):
        assert got[k].shape == fx[k].shape and maxdiff(got[k].cpu().numpy(), fx[k]) < TOL, k
    # the streaming entry points against the reference's own, step by step (test_chunk_asr.py:60-83)
    fs = fixture("tf_chunk_stream.npz")
    n, samples = int(fs["nchunks"]), int(fs["samples"])
    pc, dc = m.init_picker_caches(1), m.init_decoder_caches(1)
    ph, txt, steps, unv = [], [], [], None
    for i in range(n):
        vp, _, vh, pc = m.picker_stream_predict(xc[:1, i * samples:(i + 1) * samples, None], pc)
        if vp.shape[1] == 0:
            continue
        ph.append(vp)
        f, _ = m.feature_pick(vh, vp)
        if f.shape[1] != 0:
            vt, unv, dc = m.decoder_stream_predict(f, dc)
            txt.append(vt)
            steps.append([i, int(vt.shape[1])])
    import torch
    assert steps == fs["steps"].tolist()
    assert maxdiff(torch.cat(ph, 1).cpu().numpy(), fs["picker_logits"]) < TOL and maxdiff(torch.cat(txt, 1).cpu().numpy(), fs["text_logits"]) < TOL
    assert maxdiff(unv.cpu().numpy(), fs["unvalid_text_logits"]) < TOL
    assert maxdiff(pc[2].cpu().numpy(), fs["cache_enc_mha"]) < TOL and maxdiff(dc[2].cpu().numpy(), fs["cache_decoder_mha"]) < TOL
    fx = fixture("tf_leaf.npz")
    cfg1 = small_cfg(1)
    wl = {k: v for k, v in co.encoder_weights(cfg1, seed=7).items() if not k.startswith("mel_layer/")}
    wl.update(_leaf_weights_from(fx))
    el = ConformerEncoder(**dict(encoder_kwargs(cfg1), mel_layer_type="leaf"))
    el.load_weights(wl, by_name=False)
    assert maxdiff(el.melspectrogram(waves(2, int(fx["L"]), int(fx["wave_seed"]))).cpu().numpy(), fx["out"].reshape(2, -1, 80)) < TOL
