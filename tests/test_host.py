"""CPU tests of the host side: the C-ABI library loads and exports every symbol include/mi355asr.h declares,
argument validation / error reporting, weight-name surface, shape helpers, featurizers, config, constants.
No kernel is launched here (no GPU in this container)."""
import ctypes
import json
import os
import re
import struct
import wave

import numpy as np
import pytest

from helpers import ROOT, co, encoder_kwargs, golden_ctc_weights, small_cfg
from tensorflowasr_amd import _lib, frontend_consts
from tensorflowasr_amd.config import UserConfig
from tensorflowasr_amd.featurizers import SpeechFeaturizer, TextFeaturizer, read_raw_audio
from tensorflowasr_amd.models import ConformerCTC, ConformerEncoder, CTCDecoder, StreamingConformerEncoder
from tensorflowasr_amd.parallel import shard_range
from tensorflowasr_amd.synthetic import synth_wave


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "mi355asr.h")).read()
    declared = set(re.findall(r"\b(mi355asr_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 18
    lib = _lib.lib()
    for name in declared:
        assert hasattr(lib, name), "library does not export %s" % name
    assert declared == set(_lib.SIGNATURES), "ctypes table out of sync with the header"
    assert b"gfx950" in lib.mi355asr_version()


def test_config_struct_matches_header_layout():
    hdr = open(os.path.join(ROOT, "include", "mi355asr.h")).read()
    body = hdr[hdr.index("typedef struct {"):hdr.index("} mi355asr_config;")]
    fields = re.findall(r"^\s*(int32_t|float)\s+(\w+);", body, re.M)
    assert [f for _, f in fields] == [n for n, _ in _lib.Config._fields_]
    for (ctype, _), (_, pyt) in zip(fields, _lib.Config._fields_):
        assert (ctype == "float") == (pyt is ctypes.c_float)


def _create(**over):
    cfg = dict(dmodel=144, num_blocks=1, head_size=36, num_heads=4, kernel_size=32, fc_factor=0.5, reduction_factor=4,
               n_mels=80, sample_rate=16000, stride_ms=10, n_dft=1024, chunk_size=0, has_encoder=1, num_classes=0,
               ctc_num_blocks=0, ctc_kernel_size=32, ctc_fc_factor=0.5, gemm_dtype=0, mel_layer_type=0)
    cfg.update(over)
    lib = _lib.lib()
    p = ctypes.c_void_p()
    rc = lib.mi355asr_create(ctypes.byref(_lib.Config(**cfg)), ctypes.byref(p))
    return lib, rc, p


@pytest.mark.parametrize("bad", [dict(dmodel=100), dict(head_size=32), dict(head_size=18, num_heads=8), dict(kernel_size=0), dict(reduction_factor=3), dict(reduction_factor=10),
                                 dict(n_dft=512), dict(n_mels=64), dict(num_heads=3), dict(gemm_dtype=2)])
def test_create_rejects_unsupported_configs_with_message(bad):
    lib, rc, p = _create(**bad)
    assert rc == -1 and not p.value
    assert len(lib.mi355asr_last_error()) > 10


def test_create_accepts_the_three_reference_model_sizes():
    """conformerS / M / L .yml: dmodel 144 (4 x 36), 256 (4 x 64), 512 (8 x 64)."""
    for d, h, hs in ((144, 4, 36), (256, 4, 64), (512, 8, 64)):
        lib, rc, p = _create(dmodel=d, num_heads=h, head_size=hs)
        assert rc == 0 and p.value, lib.mi355asr_last_error()
        lib.mi355asr_destroy(p)
    lib, rc, p = _create(dmodel=320, num_heads=5, head_size=64)
    assert rc == -1 and b"multiples of 128" in lib.mi355asr_last_error()
    # round 6: any kernel size and the head sizes the three dmodels factor into are accepted (general kernels, not an error)
    for over in (dict(kernel_size=7), dict(kernel_size=16), dict(num_heads=3, head_size=48), dict(dmodel=256, num_heads=8, head_size=32),
                 dict(reduction_factor=2), dict(reduction_factor=6), dict(reduction_factor=8)):
        lib, rc, p = _create(**over)
        assert rc == 0 and p.value, (over, lib.mi355asr_last_error())
        lib.mi355asr_destroy(p)


def test_weight_surface_and_shape_validation():
    lib, rc, p = _create(num_classes=1332, ctc_num_blocks=1)
    assert rc == 0
    n = lib.mi355asr_num_weights(p)
    names = [lib.mi355asr_weight_name(p, i).decode() for i in range(n)]
    cfg = small_cfg(1)
    expect = set(co.encoder_weights(cfg, 0)) | set(co.ctc_decoder_weights(cfg, 1332))
    assert set(names) == expect                      # same tensor names as the oracle / reference layout
    a = np.zeros((144, 576), np.float32)
    dims = (ctypes.c_int64 * 2)(144, 576)
    ok = lib.mi355asr_load_weight(p, b"conformer_block_0/ff_module_1/ffn1/kernel", a.ctypes.data_as(ctypes.c_void_p), 2, dims)
    assert ok == 0
    bad = (ctypes.c_int64 * 2)(576, 144)
    assert lib.mi355asr_load_weight(p, b"conformer_block_0/ff_module_1/ffn1/kernel", a.ctypes.data_as(ctypes.c_void_p), 2, bad) == -3
    assert b"does not match" in lib.mi355asr_last_error()
    assert lib.mi355asr_load_weight(p, b"no/such/tensor", a.ctypes.data_as(ctypes.c_void_p), 2, dims) == -3
    # Keras singleton axes are accepted squeezed or not
    k = np.zeros((1024, 513), np.float32)
    assert lib.mi355asr_load_weight(p, b"mel_layer/real_kernels", k.ctypes.data_as(ctypes.c_void_p), 2,
                                    (ctypes.c_int64 * 2)(1024, 513)) == 0
    assert lib.mi355asr_load_weight(p, b"mel_layer/real_kernels", k.ctypes.data_as(ctypes.c_void_p), 4,
                                    (ctypes.c_int64 * 4)(1024, 1, 1, 513)) == 0
    # finalising with tensors missing names the first missing one
    assert lib.mi355asr_finalize_weights(p, None) == -3
    assert b"missing weight" in lib.mi355asr_last_error()
    # forward before finalise is a state error, not a crash
    assert lib.mi355asr_encoder_forward(p, None, 1, 16000, None, None, 0, None) == -2
    assert lib.mi355asr_destroy(p) == 0


@pytest.mark.parametrize("L,F,T", [(160000, 1000, 250), (67263, 421, 106), (16000, 100, 25), (8000, 50, 13), (480000, 3000, 750)])
def test_out_frames(L, F, T):
    lib, rc, p = _create()
    f, t = ctypes.c_int32(), ctypes.c_int32()
    assert lib.mi355asr_out_frames(p, L, ctypes.byref(f), ctypes.byref(t)) == 0
    assert (f.value, t.value) == (F, T)
    lib.mi355asr_destroy(p)


def test_streaming_geometry_and_workspace():
    lib, rc, p = _create(dmodel=256, head_size=64, kernel_size=5, chunk_size=8000)
    f, t = ctypes.c_int32(), ctypes.c_int32()
    assert lib.mi355asr_out_frames(p, 24000, ctypes.byref(f), ctypes.byref(t)) == 0
    assert (f.value, t.value) == (150, 39)                                   # 3 blocks x (50, 13)
    assert lib.mi355asr_out_frames(p, 12000, ctypes.byref(f), ctypes.byref(t)) == -1
    assert b"multiple of chunk_size" in lib.mi355asr_last_error()
    n1, n2 = ctypes.c_size_t(), ctypes.c_size_t()
    assert lib.mi355asr_workspace_bytes(p, 2, 24000, ctypes.byref(n1)) == 0
    assert lib.mi355asr_workspace_bytes(p, 4, 24000, ctypes.byref(n2)) == 0
    assert 0 < n1.value < n2.value <= 2 * n1.value + 4096 * 16
    lib.mi355asr_destroy(p)


def test_workspace_is_modest_at_benchmark_shape():
    lib, rc, p = _create(num_blocks=13, num_classes=1332, ctc_num_blocks=1)
    n = ctypes.c_size_t()
    assert lib.mi355asr_workspace_bytes(p, 64, 160000, ctypes.byref(n)) == 0
    assert 300e6 < n.value < 700e6          # log-power (135 MB) + conv2 output (184 MB) dominate
    lib.mi355asr_destroy(p)


def test_python_models_mirror_reference_constructor_surface():
    cfg = small_cfg(2)
    enc = ConformerEncoder(**encoder_kwargs(cfg), dropout=0.1, add_wav_info=False, mel_layer_trainable=False,
                           name="conformer_encoder")
    assert enc.hop_size == 640                                           # conformer_blocks.py:302
    assert enc.count_params() == sum(int(np.prod(v.shape)) for v in co.encoder_weights(cfg, 0).values())
    # any other mel_layer_type builds the plain Spectrogram layer, as in the reference (conformer_blocks.py:318-323):
    # 513 dB bins, no freq2mel, Dense over F2 = 129 subsampled bins
    sp = ConformerEncoder(mel_layer_type="Spectrogram", num_blocks=1)
    sp_names = {n: tuple(s) for n, s in sp._names_and_shapes()}
    assert "mel_layer/freq2mel" not in sp_names and sp_names["mel_layer/real_kernels"] == (1024, 1, 1, 513)
    assert sp_names["conv_subsampling/linear/kernel"] == (129 * 144, 144)
    assert set(sp_names) == set(sp._h.weight_names())
    ow = co.encoder_weights(dict(small_cfg(1), mel_layer_type="Spectrogram"), 0)      # the oracle's tensors (DFT kernels 2-D)
    assert set(ow) == set(sp_names)
    assert all(int(np.prod(ow[k].shape)) == int(np.prod(sp_names[k])) for k in ow)
    leaf = ConformerEncoder(mel_layer_type="leaf", num_blocks=1)       # the reference's default frontend
    assert "mel_layer/tfbanks_complex_conv/kernel" in leaf._h.weight_names()
    assert "mel_layer/real_kernels" not in leaf._h.weight_names()
    wv = ConformerEncoder(mel_layer_type="Melspectrogram", add_wav_info=True, num_blocks=1)    # WavePickModel branch
    names = {n: tuple(s) for n, s in wv._names_and_shapes()}
    ref_w = co.wave_pick_weights(144, 640)
    assert {k: v for k, v in names.items() if k.startswith("wav_layer/")} == {k: v.shape for k, v in ref_w.items()}
    assert set(names) == set(wv._h.weight_names())                       # the handle expects exactly these tensors
    with pytest.raises(Exception, match="four strides"):
        ConformerEncoder(mel_layer_type="Melspectrogram", add_wav_info=True, stride_ms=1, sample_rate=1000, num_blocks=1)
    dec = CTCDecoder(num_classes=1332, dmodel=144, num_blocks=1, head_size=36, num_heads=4, kernel_size=32,
                     dropout=0.1, fc_factor=0.5)
    assert dec.count_params() == 759_348 + 0 or dec.count_params() > 700_000     # ~0.76 M
    st = StreamingConformerEncoder(**encoder_kwargs(small_cfg(1, co.STREAMING_S)))
    st.add_chunk_size(8000, 80, 640)
    assert st.mel_length == 13 and st.chunk_size == 8000
    full = ConformerCTC(1332, **{k: v for k, v in encoder_kwargs(dict(co.CONFORMER_S)).items() if k != "mel_layer_type"})
    assert abs(full.count_params() - 9.59e6) < 0.05e6                       # 7.74 M + 1.09 M fixed + 0.76 M
    assert full.blank == 1331


def test_translator_surface_and_config_struct():
    """Translator(inp_classes, tar_classes, dmodel, num_blocks, head_size, num_heads, kernel_size, dropout, fc_factor)
    as test_asr.py:76-84 constructs it; SURVEY 8a: 2.62 M parameters for the S config."""
    from tensorflowasr_amd.models import Translator
    hdr = open(os.path.join(ROOT, "include", "mi355asr.h")).read()
    body = hdr[hdr.index("typedef struct {", hdr.index("Translator:")):hdr.index("} mi355asr_translator_config;")]
    fields = re.findall(r"(int32_t|float)\s+([\w, ]+);", body)
    names = [n.strip() for _, group in fields for n in group.split(",")]
    assert names == [n for n, _ in _lib.TranslatorConfig._fields_]
    tr = Translator(inp_classes=1332, tar_classes=9160, dmodel=144, num_blocks=2, head_size=36, num_heads=4,
                    kernel_size=32, dropout=0.1, fc_factor=0.5)
    assert abs(tr.count_params() - 2.62e6) < 0.02e6
    names = tr._h.weight_names()
    assert names[0] == "inp_embedding/embeddings" and names[-1] == "fully_connected/bias"
    assert "decoder_conformer_block_1/mhsa_module/mha/key_kernel" in names
    lib = _lib.lib()
    p = ctypes.c_void_p()
    bad = _lib.TranslatorConfig(dmodel=144, num_blocks=0, head_size=36, num_heads=4, kernel_size=32, fc_factor=0.5,
                                inp_classes=10, tar_classes=10)
    assert lib.mi355asr_translator_create(ctypes.byref(bad), ctypes.byref(p)) == -1
    n = ctypes.c_size_t()
    assert lib.mi355asr_translator_workspace_bytes(tr._h.ptr, 1, 30, 250, ctypes.byref(n)) == 0 and n.value > 0
    assert lib.mi355asr_translator_workspace_bytes(tr._h.ptr, 1, 0, 250, ctypes.byref(n)) == -1


def test_wer_matches_reference_xer_kats():
    """utils/xer.py:211-220 (distance AND the S / D / I split of the reference's tie-breaking); fixtures made by
    importing the reference module (tests/golden/make_golden.py::make_wer_kats)."""
    from tensorflowasr_amd.eval import levenshtein, wer
    kats = json.load(open(os.path.join(ROOT, "tests", "golden", "wer_kat.json")))
    assert len(kats) == 60
    for k in kats:
        score, s, d, i = wer(k["r"], k["h"])
        assert (s, d, i) == (k["s"], k["d"], k["i"]), k
        assert score == k["score"]
    assert levenshtein("kitten", "sitting")[0] == 3
    assert wer(list("abc"), [])[1:] == (0, 3, 0)


def _eval_fixture(tmp_path, streaming):
    import wave
    from tensorflowasr_amd.config import load_yaml
    (tmp_path / "phones.txt").write_text("\n".join(["<S>", "</S>", "[SPACE]", "[UNK]"] + ["p%d" % i for i in range(20)]) + "\n")
    (tmp_path / "chars.txt").write_text("\n".join(["<S>", "</S>", "[SPACE]", "[UNK]"] + [chr(0x4e00 + i) for i in range(30)]) + "\n")
    here = os.path.join(ROOT, "tensorflowasr_amd", "configs")
    cfg = load_yaml(os.path.join(here, "am_data_streaming.yml" if streaming else "am_data.yml"))
    cfg.update(load_yaml(os.path.join(here, "Streaming_ConformerS.yml" if streaming else "conformerS.yml")))
    cfg["inp_config"]["vocabulary"] = str(tmp_path / "phones.txt")
    cfg["tar_config"]["vocabulary"] = str(tmp_path / "chars.txt")
    lines = []
    for n, L in enumerate([12000, 300, 20480, 16000 * 8, 9000]):      # 300: too short, 8 s: over wav_max_duration
        x = (0.3 * np.sin(np.arange(L) * (0.01 + 0.003 * n))).astype(np.float32)
        with wave.open(str(tmp_path / ("u%d.wav" % n)), "wb") as f:
            f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000)
            f.writeframes((x * 32767).astype("<i2").tobytes())
        lines.append("%s\t%s\t%s" % (tmp_path / ("u%d.wav" % n), chr(0x4e00 + n) + chr(0x4e01 + n), "p%d p%d p1" % (n, n + 3)))
    lines.append("%s\t%s\tp1 zz" % (tmp_path / "u0.wav", chr(0x4e00)))     # unknown phone token: skipped
    (tmp_path / "eval.list").write_text("\n".join(lines) + "\n")
    cfg["speech_config"]["eval_list"] = str(tmp_path / "eval.list")
    return cfg


@pytest.mark.parametrize("streaming", [False, True])
def test_eval_list_batches_like_the_reference_loader(tmp_path, streaming):
    """am_dataloader.py:118-227: skips, peak normalisation, in_len, padding (offline and streaming)."""
    from tensorflowasr_amd.eval import EvalList
    from tensorflowasr_amd.featurizers import SpeechFeaturizer, TextFeaturizer
    cfg = _eval_fixture(tmp_path, streaming)
    ds = EvalList(cfg, SpeechFeaturizer(cfg["speech_config"]), TextFeaturizer(cfg["inp_config"]),
                  TextFeaturizer(cfg["tar_config"]), batch_size=3)
    x, in_len, ph, ph_len, txt = ds.eval_data_generator()
    assert x.dtype == np.float32 and x.ndim == 3 and x.shape[0] == 3 and x.shape[2] == 1
    assert ph.tolist() == [[4, 7, 5], [6, 9, 5], [8, 11, 5]] and ph_len.tolist() == [3, 3, 3]   # p0 p3 p1 / p2 p5 p1 / p4 p7 p1
    assert txt[0].tolist() == [4, 5, 1] and txt[1].tolist() == [6, 7, 1]          # chars + </S>
    if not streaming:
        assert x.shape[1] == 20480 and in_len.tolist() == [12000 // 640, 20480 // 640, 9000 // 640]
        assert abs(np.abs(x[0]).max() - 1.0) < 1e-6 and abs(np.abs(x[2]).max() - 1.0) < 1e-6   # peak-normalised
        assert np.all(x[0, 12000:] == 0)
    else:
        # longest 20480 -> 24000 (next block boundary) -> 32000: the reference rounds up twice (am_dataloader.py:198-209),
        # every streaming batch carries one extra all-zero block
        assert x.shape[1] == 32000 and np.all(x[:, 24000:] == 0)
        assert in_len.tolist() == [2 * 13, 3 * 13, 2 * 13]                         # whole blocks x 13 frames
        assert abs(np.abs(x[0]).max() - 0.3) < 1e-3                                # raw samples, not normalised


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    enc = ConformerEncoder(**encoder_kwargs(small_cfg(1)))
    with pytest.raises(Exception):
        enc(np.zeros((1, 16000, 1), np.float32))       # no CPU fallback: must raise, not compute


def test_leaf_default_weights_match_oracle_restatement():
    a, b = frontend_consts.leaf_default_weights(), co.leaf_default_weights()
    assert sorted(a) == sorted(b)
    for k in a:
        assert a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), k
    k = a["mel_layer/tfbanks_complex_conv/kernel"]
    assert k.shape == (80, 2) and np.all(np.diff(k[:, 0]) >= 0) and 0 < k[0, 0] < k[-1, 0] < np.pi   # mel-spaced centres


def test_frontend_constants_match_oracle_restatement():
    re_, im_ = frontend_consts.stft_kernels(1024)
    ro, io = co.stft_kernels(1024)
    assert re_.shape == (1024, 1, 1, 513)
    assert np.array_equal(re_.reshape(1024, 513), ro) and np.array_equal(im_.reshape(1024, 513), io)
    assert np.abs(frontend_consts.freq2mel() - co.mel_filterbank().T).max() < 1e-7
    assert np.abs(frontend_consts.freq2mel(norm="slaney") - co.mel_filterbank(norm="slaney").T).max() < 1e-7
    assert np.array_equal(synth_wave(3, 4000), co.synth_wave(3, 4000))


def test_speech_featurizer_reads_pcm16_like_librosa(tmp_path):
    sr = 16000
    x = (0.3 * np.sin(2 * np.pi * 440 * np.arange(sr // 4) / sr) * 32767).astype("<i2")
    path = str(tmp_path / "a.wav")
    with wave.open(path, "wb") as f:
        f.setnchannels(1); f.setsampwidth(2); f.setframerate(sr); f.writeframes(x.tobytes())
    sf = SpeechFeaturizer({"sample_rate": 16000, "frame_ms": 25, "stride_ms": 10, "num_feature_bins": 80})
    w = sf.load_wav(path)
    assert w.dtype == np.float32 and w.shape == (sr // 4,)
    assert np.array_equal(w, x.astype(np.float32) / 32768.0)
    assert np.array_equal(read_raw_audio(open(path, "rb").read()), w)
    # stereo 8 kHz -> mono 16 kHz
    st = np.stack([x[::2], x[::2]], 1)
    p2 = str(tmp_path / "b.wav")
    with wave.open(p2, "wb") as f:
        f.setnchannels(2); f.setsampwidth(2); f.setframerate(8000); f.writeframes(st.tobytes())
    w2 = sf.load_wav(p2)
    assert abs(len(w2) - len(w)) <= 2
    padded = sf.pad_signal([w[:10], w[:4]], 8)
    assert padded.shape == (2, 8) and (padded[1, 4:] == 0).all() and np.array_equal(padded[0], w[:8])
    assert sf.compute_time_dim(1.0) == 101


def test_text_featurizer_blank_last_and_space(tmp_path):
    vocab = tmp_path / "v.txt"
    vocab.write_text("<S>\n</S>\n[SPACE]\n# comment\n\nni3\nhao3\n", encoding="utf-8")
    tf = TextFeaturizer({"vocabulary": str(vocab), "blank_at_zero": False, "beam_width": 1})
    assert tf.num_classes == 6 and tf.blank == 5
    assert tf.startid() == 0 and tf.endid() == 1 and tf.token_to_index[" "] == 2
    assert tf.iextract([3, 4]) == ["ni3", "hao3"] and tf.extract(["hao3"]) == [4]
    tz = TextFeaturizer({"vocabulary": str(vocab), "blank_at_zero": True, "beam_width": 1})
    assert tz.blank == 0 and tz.num_classes == 6 and tz.token_to_index["<S>"] == 1


def test_user_config_merge_and_missing_key(tmp_path):
    a = tmp_path / "a.yml"
    b = tmp_path / "b.yml"
    a.write_text("speech_config:\n  sample_rate: 16000\nmodel_config:\n  name: X\n")
    b.write_text("model_config:\n  name: OfflineConformerCTC\n  dmodel: 144\n")
    c = UserConfig(str(a), str(b))
    assert c["model_config"]["dmodel"] == 144 and c["speech_config"]["sample_rate"] == 16000
    assert c["learning_config"] is None                                  # utils/user_config.py:24-25


def test_shipped_configs_build_the_documented_models():
    cdir = os.path.join(ROOT, "tensorflowasr_amd", "configs")
    c = UserConfig(os.path.join(cdir, "am_data.yml"), os.path.join(cdir, "conformerS.yml"))
    m = ConformerCTC.from_config(c, 1332)
    assert (m.dmodel, m.num_blocks, m.head_size, m.kernel_size, m.chunk_size) == (144, 13, 36, 32, 0)
    c2 = UserConfig(os.path.join(cdir, "am_data_streaming.yml"), os.path.join(cdir, "Streaming_ConformerS.yml"))
    m2 = ConformerCTC.from_config(c2, 1332)
    assert (m2.dmodel, m2.num_blocks, m2.head_size, m2.kernel_size, m2.chunk_size) == (256, 4, 64, 5, 8000)


def test_shard_range_is_a_partition():
    for n in (512, 64, 10, 3):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


# ---- prefix beam search host path (mi355asr_ctc_prefix_beam_host) vs the reference decoder's KATs -------------------
def test_prefix_beam_host_matches_reference_kats_bit_exact():
    import json
    from tensorflowasr_amd.models import ctc_prefix_beam_decode
    k = np.load(os.path.join(ROOT, "tests", "golden", "beam_kat.npz"))
    meta = json.loads(str(k["meta"]))
    for i, m in enumerate(meta):
        ids, lens, sc, n = ctc_prefix_beam_decode(k["probs_%d" % i][None], None, m["beam"], m["cutoff_prob"],
                                                  m["cutoff_top_n"], num_threads=1)
        assert n[0] == m["n"]
        assert np.array_equal(lens[0, :m["n"]], k["lens_%d" % i])
        assert np.array_equal(ids[0, :m["n"]], k["ids_%d" % i])
        assert np.array_equal(sc[0, :m["n"]].astype(np.float64), k["scores_%d" % i])   # float32 trie scores, same libm


def _long_kat_check(decode):
    """beam_long_kat.npz (the reference's own decoder on 400-600 frames): what the reference SPECIFIES -- the ranked
    float32 scores -- must be reproduced bit for bit; which of several prefixes with the same score and last character
    survives is left to std::nth_element there (half of the scores of these vectors are shared by two or more
    hypotheses), so beyond the scores only the overlap is recorded: hypotheses outside tied groups must coincide."""
    import json
    k = np.load(os.path.join(ROOT, "tests", "golden", "beam_long_kat.npz"))
    meta = json.loads(str(k["meta"]))
    res = []
    for i, m in enumerate(meta):
        ids, lens, sc, n = decode(k["probs_%d" % i][None], m["beam"], m["cutoff_prob"], m["cutoff_top_n"])
        nn = m["n"]
        assert n[0] == nn
        ref_sc = k["scores_%d" % i]
        assert np.array_equal(sc[0, :nn].astype(np.float64), ref_sc), i
        ours = [tuple(ids[0, j, :lens[0, j]]) for j in range(nn)]
        ref = [tuple(k["ids_%d" % i][j, :k["lens_%d" % i][j]]) for j in range(nn)]
        assert len(set(ours)) == nn
        tied = np.array([(ref_sc == v).sum() > 1 for v in ref_sc])
        if not tied.any():
            assert ours == ref, i                                        # no ties: the whole beam, in order
        assert len(set(ours) & set(ref)) >= 0.9 * nn, i
        res.append((ids, lens, sc, n))
    assert meta[0]["T"] >= 500 and float(k["scores_0"][0]) < -1000
    return res


def test_prefix_beam_host_long_inputs_vs_the_reference_decoder():
    from tensorflowasr_amd.models import ctc_prefix_beam_decode
    _long_kat_check(lambda p, beam, cp, tn: ctc_prefix_beam_decode(p, None, beam, cp, tn, num_threads=1))


def test_prefix_beam_decode_of_zero_frames_is_the_empty_prefix():
    """T = 0 (a ChunkConformer batch in which the picker kept no frame): one hypothesis per utterance, the empty prefix with
    log-probability 0 -- what an utterance of length 0 inside a longer batch gives -- instead of an error from the C API that
    would wedge ChunkBeamPipeline (round-3 advice)."""
    from tensorflowasr_amd.models import ctc_prefix_beam_decode
    ids, lens, sc, n = ctc_prefix_beam_decode(np.zeros((3, 0, 7), np.float32), None, 4, 0.99, 5)
    p = np.full((3, 2, 7), 1 / 7, np.float32)
    ids2, lens2, sc2, n2 = ctc_prefix_beam_decode(p, np.zeros(3, np.int32), 4, 0.99, 5)
    assert np.array_equal(n, n2) and np.array_equal(lens, lens2) and np.array_equal(sc, sc2)
    assert (ids == -1).all() and n.tolist() == [1, 1, 1] and sc[:, 0].tolist() == [0.0, 0.0, 0.0]


def test_prefix_beam_batch_threads_ragged_lengths():
    import json
    from oracle import ctc_beam_oracle as bo
    from tensorflowasr_amd.models import ctc_prefix_beam_decode
    rng = np.random.default_rng(5)
    B, T, V, beam = 5, 24, 9, 6
    z = rng.standard_normal((B, T, V)) * 2
    p = np.exp(z - z.max(-1, keepdims=True))
    p = (p / p.sum(-1, keepdims=True)).astype(np.float32)
    in_len = np.array([24, 1, 0, 13, 24], np.int32)
    a = ctc_prefix_beam_decode(p, in_len, beam, 0.95, 5, num_threads=3)
    b = ctc_prefix_beam_decode(p, in_len, beam, 0.95, 5, num_threads=1)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))                      # threading does not change results
    ids, lens, sc, n = a
    for u in range(B):
        ref = bo.ctc_beam_search(p[u, :in_len[u]], beam, 0.95, 5) if in_len[u] else [(np.float32(0.0), [])]
        assert n[u] == len(ref)
        for j, (s, path) in enumerate(ref):
            assert ids[u, j, :lens[u, j]].tolist() == path and abs(float(s) - sc[u, j]) < 1e-4
    assert n[2] == 1 and lens[2, 0] == 0 and sc[2, 0] == 0.0                    # empty input: the root prefix only
    # beam 1 with a peaked distribution reproduces greedy decoding
    fa = p.argmax(-1)
    onehot = np.full((B, T, V), 1e-6, np.float32)
    np.put_along_axis(onehot, fa[..., None], 1.0, axis=-1)
    ids1, lens1, _, _ = ctc_prefix_beam_decode(onehot, None, 1, 1.0, 40, num_threads=2)
    gid, glen = co.ctc_collapse(fa.astype(np.int32), [T] * B, V - 1)
    for u in range(B):
        assert ids1[u, 0, :lens1[u, 0]].tolist() == gid[u, :glen[u]].tolist()


def test_stateful_beam_decoder_matches_reference_class_bit_exact():
    """BeamDecoder.decode / reset (ctc_beam_search_decoder.cpp:217-405) fed in pieces: beam contents, order and
    float32 scores after every call equal the reference class compiled in place (tests/golden/make_golden.py)."""
    from tensorflowasr_amd.models import BeamDecoder
    kat = np.load(os.path.join(ROOT, "tests", "golden", "beam_stateful_kat.npz"))
    ncases = len([k for k in kat.files if k.endswith("_meta")])
    assert ncases == 4
    for ci in range(ncases):
        V, beam, ctn, npieces = [int(v) for v in kat["c%d_meta" % ci]]
        cp = float(kat["c%d_cp" % ci][0])
        pieces, probs = kat["c%d_pieces" % ci], kat["c%d_probs" % ci]
        dec = BeamDecoder(["%04x" % v for v in range(V)], beam, cp, ctn)
        call = 0
        for u in range(2):
            if u:
                dec.reset()
            o = 0
            for n in pieces:
                got = dec.decode_ids(probs[u, o:o + n], max_len=probs.shape[1])
                o += int(n)
                k = int(kat["c%d_n" % ci][call])
                assert len(got) == k
                for i in range(k):
                    ln = int(kat["c%d_lens" % ci][call][i])
                    assert got[i][1] == kat["c%d_ids" % ci][call][i][:ln].tolist(), (ci, call, i)
                    assert np.float32(got[i][0]) == kat["c%d_scores" % ci][call][i], (ci, call, i)
                call += 1
        # piecewise == whole
        dec.reset()
        whole = dec.decode_ids(probs[1], max_len=probs.shape[1])
        assert [w[1] for w in whole] == [g[1] for g in got]
        text = BeamDecoder(["%04x" % v for v in range(V)], beam, cp, ctn).decode(probs[1])
        assert text[0][1] == "".join("%04x" % t for t in whole[0][1])
    with pytest.raises(_lib.Mi355AsrError):
        BeamDecoder(["a"], 4)
    with pytest.raises(NotImplementedError):
        BeamDecoder(["a", "b"], 4, ext_scorer=object())


def test_prefix_beam_argument_errors():
    from tensorflowasr_amd.models import ctc_prefix_beam_decode
    with pytest.raises(_lib.Mi355AsrError):
        ctc_prefix_beam_decode(np.zeros((1, 3, 1), np.float32), None, 4)        # V < 2
    with pytest.raises(ValueError):
        ctc_prefix_beam_decode(np.zeros((1, 3, 4), np.float32), None, 4, is_logits=True)


REF_ONNX = "/root/reference/Inference/PythonInference/asr/models/offline/ctc_model.onnx"


@pytest.mark.skipif(not os.path.exists(REF_ONNX), reason="the reference tree is only present in the build container")
def test_ctc_decoder_weights_from_the_reference_onnx_export():
    """checkpoint.py (product code, structural tracing of the tf2onnx graph) recovers exactly the tensors that the
    oracle-side extraction pinned by executing the graph (tests/golden/ctc_decoder_weights.npz), and CTCDecoder accepts
    the .onnx path."""
    from tensorflowasr_amd import checkpoint
    w = checkpoint.ctc_decoder_weights_from_onnx(REF_ONNX, num_heads=4)
    gold = golden_ctc_weights()
    assert set(w) == set(gold)
    for k in gold:
        assert w[k].dtype == np.float32 and np.array_equal(w[k], gold[k]), k
    dec = CTCDecoder(num_classes=1332, dmodel=144, num_blocks=1, head_size=36, num_heads=4, kernel_size=32)
    assert {n: tuple(s) for n, s in dec._names_and_shapes()} == {k: v.shape for k, v in w.items()}
    nodes, inits = checkpoint.read_onnx(REF_ONNX)
    assert len(nodes) == 307 and len(inits) == 62


def test_keras_variable_names_map_to_abi_names():
    """checkpoint.keras_names_to_abi on names shaped like the reference's Keras variables: auto-numbered layers are
    resolved by their order inside a scope (the numbering seen in the exported graph: dense_53.., layer_normalization_65..)."""
    from tensorflowasr_amd import checkpoint
    blk = "ctc_decoder/decoder_conformer_block_0"
    names = ["ctc_decoder/dense_53/kernel:0", "ctc_decoder/dense_53/bias:0",
             blk + "/ff_module_1/layer_normalization_65/gamma:0", blk + "/ff_module_1/layer_normalization_65/beta:0",
             blk + "/ff_module_1/dense_54/kernel:0", blk + "/ff_module_1/dense_54/bias:0",
             blk + "/ff_module_1/dense_55/kernel:0", blk + "/ff_module_1/dense_55/bias:0",
             blk + "/mhsa_module/layer_normalization_66/gamma:0",
             blk + "/mhsa_module/multi_head_attention_13/query_kernel:0",
             blk + "/mhsa_module/multi_head_attention_13/projection_bias:0",
             blk + "/conv_module/layer_normalization_67/beta:0", blk + "/conv_module/pw_conv_1/kernel:0",
             blk + "/conv_module/dw_conv/depthwise_kernel:0", blk + "/conv_module/dw_conv/pointwise_kernel:0",
             blk + "/conv_module/batch_normalization_13/moving_variance:0", blk + "/conv_module/pw_conv_2/bias:0",
             blk + "/ff_module_2/dense_57/kernel:0", blk + "/ff_module_2/dense_56/kernel:0",
             blk + "/layer_normalization_69/gamma:0", "ctc_decoder/fully_connected/kernel:0", "Adam/iter:0"]
    m = checkpoint.keras_names_to_abi(names)
    b = "decoder_conformer_block_0"
    assert m["ctc_decoder/dense_53/kernel:0"] == "project/kernel"
    assert m[blk + "/ff_module_1/dense_54/kernel:0"] == b + "/ff_module_1/ffn1/kernel"
    assert m[blk + "/ff_module_1/dense_55/bias:0"] == b + "/ff_module_1/ffn2/bias"
    assert m[blk + "/ff_module_2/dense_56/kernel:0"] == b + "/ff_module_2/ffn1/kernel"
    assert m[blk + "/ff_module_2/dense_57/kernel:0"] == b + "/ff_module_2/ffn2/kernel"
    assert m[blk + "/mhsa_module/multi_head_attention_13/query_kernel:0"] == b + "/mhsa_module/mha/query_kernel"
    assert m[blk + "/conv_module/batch_normalization_13/moving_variance:0"] == b + "/conv_module/bn/moving_variance"
    assert m[blk + "/layer_normalization_69/gamma:0"] == b + "/ln/gamma"
    assert m["ctc_decoder/fully_connected/kernel:0"] == "fully_connected/kernel"
    assert "Adam/iter:0" not in m
    dec = CTCDecoder(num_classes=1332, dmodel=144, num_blocks=1, head_size=36, num_heads=4, kernel_size=32)
    assert set(m.values()) <= {n for n, _ in dec._names_and_shapes()}
    enc = ["conformer_encoder/conv_subsampling/conv2d_4/kernel:0", "conformer_encoder/conv_subsampling/conv2d_5/bias:0",
           "conformer_encoder/conv_subsampling/dense_9/kernel:0",
           "conformer_encoder/conformer_block_11/ff_module_2/dense_99/bias:0",
           "conformer_encoder/conformer_block_11/ff_module_2/dense_98/bias:0"]
    e = checkpoint.keras_names_to_abi(enc)
    assert e[enc[0]] == "conv_subsampling/conv1/kernel" and e[enc[1]] == "conv_subsampling/conv2/bias"
    assert e[enc[2]] == "conv_subsampling/linear/kernel"
    assert e[enc[3]] == "conformer_block_11/ff_module_2/ffn2/bias" and e[enc[4]] == "conformer_block_11/ff_module_2/ffn1/bias"


def _keras_style_names(scope, abi_names, start=7):
    """inverse of checkpoint.keras_names_to_abi for a whole model: C-ABI names (in weight_names() = construction order) ->
    names as Keras would print them, auto-numbered layers counted globally per kind from `start`."""
    kinds = {"ln": "layer_normalization", "bn": "batch_normalization", "mha": "multi_head_attention", "ffn1": "dense",
             "ffn2": "dense", "project": "dense", "linear": "dense", "conv1": "conv2d", "conv2": "conv2d",
             "inp_embedding": "embedding", "sep_conv": "separable_conv1d", "final": "conv1d", "conv5": "conv1d"}
    counter, seen, out = {}, {}, {}
    for abi in abi_names:
        parts = abi.split("/")
        keras = []
        for depth, p in enumerate(parts[:-1]):
            kind = kinds.get(p)
            if p.startswith("res_"):
                kind = "tf_residual_stack"
            elif re.fullmatch(r"conv_\d+", p) or (p == "conv1" and depth and parts[depth - 1].startswith("res_")):
                kind = "conv1d"
            if p == "wav_layer":
                keras += ["wave_pick_model", "sequential_3"]
                continue
            if p in ("conv1", "conv2") and parts[depth - 1] != "conv_subsampling" and not parts[depth - 1].startswith("res_"):
                kind = None
            if kind is None:
                keras.append(p)
                continue
            key = tuple(parts[:depth + 1])
            if key not in seen:
                n = counter.get(kind, start)
                counter[kind] = n + 1
                seen[key] = "%s_%d" % (kind, n) if n else kind
            keras.append(seen[key])
        out["/".join([scope] + keras + [parts[-1]]) + ":0"] = abi
    return out


def test_keras_names_cover_every_tensor_of_encoder_with_wav_info_and_translator():
    """the Translator's Embedding and the WavePickModel branch were not mapped, so loading the reference's
    `.h5` left them at their random _build() values without an error.  Whole-model name lists, written the way Keras
    auto-numbers layers, must map onto exactly Handle.weight_names()."""
    import re as _re  # noqa: F401
    from tensorflowasr_amd import checkpoint
    from tensorflowasr_amd.models import ConformerEncoder, Translator
    enc = ConformerEncoder(dmodel=144, num_blocks=2, add_wav_info=True, mel_layer_type="Melspectrogram")
    names = [n for n in enc._h.weight_names() if not n.startswith("mel_layer/")]     # DFT / mel constants: see keras_h5_to_abi
    k2a = _keras_style_names("conformer_encoder", names)
    assert len(k2a) == len(names)
    assert any("wave_pick_model/sequential_3/tf_residual_stack_8/conv1d_" in k for k in k2a)
    assert checkpoint.keras_names_to_abi(list(k2a) + ["Adam/iter:0"]) == k2a
    tr = Translator(inp_classes=60, tar_classes=80, dmodel=144, num_blocks=2)
    names = tr._h.weight_names()
    k2a = _keras_style_names("translator", names, start=2)
    assert "translator/embedding_2/embeddings:0" in k2a
    assert checkpoint.keras_names_to_abi(list(k2a)) == k2a
    assert set(k2a.values()) == set(names)


def test_keras_names_of_leaf_sublayers_under_the_attribute_scope_and_the_bare_gabor_kernel():
    """Round-5 advice: (1) `conformer_encoder/mel_layer/<LEAF sub-layer>/...` (what this repository's own tools write) must map
    again -- the Melspectrogram branch had started to drop every name deeper than two parts; (2) a bare `kernel:0` is the Gabor
    kernel only next to `leaf/...` variables, never in a Melspectrogram model's name list."""
    from tensorflowasr_amd import checkpoint
    from tensorflowasr_amd.models import ConformerEncoder
    enc = ConformerEncoder(dmodel=144, num_blocks=1, mel_layer_type="leaf")
    leaf_names = [n for n in enc._h.weight_names() if n.startswith("mel_layer/")]
    assert any("PCEN" in n for n in leaf_names) and any("learnable_pooling" in n for n in leaf_names) and len(leaf_names) == 9
    keras = {"conformer_encoder/%s:0" % n: n for n in leaf_names}
    assert checkpoint.keras_names_to_abi(list(keras)) == keras
    # the reference's own scoping (seen on the stand-in): leaf/... plus the bare kernel
    ref_style = {"kernel:0": "mel_layer/tfbanks_complex_conv/kernel"}
    ref_style.update({"leaf/%s:0" % n[len("mel_layer/"):]: n for n in leaf_names if "tfbanks_complex_conv" not in n})
    assert checkpoint.keras_names_to_abi(list(ref_style)) == ref_style
    # ... and no Gabor kernel appears in a Melspectrogram model because some variable is called `kernel:0`
    mel_model = ["conformer_encoder/melspectrogram/real_kernels:0", "conformer_encoder/melspectrogram/Variable:0", "kernel:0"]
    got = checkpoint.keras_names_to_abi(mel_model)
    assert "kernel:0" not in got and sorted(got.values()) == ["mel_layer/freq2mel", "mel_layer/real_kernels"]


def test_load_weights_from_a_file_checks_coverage(tmp_path):
    """a checkpoint whose names do not map, or that lacks tensors, must not load as a random-weight model"""
    from tensorflowasr_amd import _lib
    dec = CTCDecoder(num_classes=50, dmodel=144, num_blocks=1, head_size=36, num_heads=4, kernel_size=32)
    shapes = dict(dec._names_and_shapes())
    rng = np.random.default_rng(0)
    full = {n: rng.standard_normal(s).astype(np.float32) for n, s in shapes.items()}
    np.savez(tmp_path / "wrong_names.npz", **{"model/" + k: v for k, v in full.items()})
    with pytest.raises(_lib.Mi355AsrError, match="none of the"):
        dec.load_weights(str(tmp_path / "wrong_names.npz"))
    part = {k: v for k, v in full.items() if "ff_module_2" not in k}
    np.savez(tmp_path / "partial.npz", **part)
    with pytest.raises(_lib.Mi355AsrError, match="are not in this file"):
        dec.load_weights(str(tmp_path / "partial.npz"))
    with pytest.raises(Exception, match="HIP|GPU|ROCm|device"):   # coverage check passed; the upload needs a device and fails loudly
        dec.load_weights(str(tmp_path / "partial.npz"), allow_missing=True)
    assert "decoder_conformer_block_0/ff_module_1/ffn1/kernel" in dec.get_weights_dict()


def test_latest_checkpoint_picks_the_newest_step_of_any_format(tmp_path):
    """test_asr.py:95-114: `<sub-model>-ckpt/model_<step>.h5`, highest step wins; stray files are ignored"""
    from tensorflowasr_amd.checkpoint import latest_checkpoint
    d = tmp_path / "encoder-ckpt"
    d.mkdir()
    with pytest.raises(FileNotFoundError):
        latest_checkpoint(str(d))
    for f in ("model_5.h5", "model_40.h5", "model_7.npz", "notes.txt", "model_final.h5", "checkpoint", "model_100.tmp"):
        (d / f).write_bytes(b"")
    assert latest_checkpoint(str(d)).endswith("model_40.h5")
    (d / "model_40.npz").write_bytes(b"")
    assert latest_checkpoint(str(d)).endswith("model_40.npz")          # same step: npz before h5
    (d / "model_41.index").write_bytes(b"")
    (d / "model_41.data-00000-of-00001").write_bytes(b"")
    assert latest_checkpoint(str(d)) == str(d / "model_41")             # TensorFlow bundle: the prefix


def test_latest_tf_checkpoint_reads_the_state_file(tmp_path):
    """tf.train.latest_checkpoint (test_chunk_asr.py:41): the `checkpoint` state file wins; otherwise the highest *.index"""
    from tensorflowasr_amd.checkpoint import latest_tf_checkpoint
    d = tmp_path / "all-ckpt"
    d.mkdir()
    with pytest.raises(FileNotFoundError):
        latest_tf_checkpoint(str(d))
    for f in ("model_3.index", "model_3.data-00000-of-00001", "model_12.index", "model_12.data-00000-of-00001"):
        (d / f).write_bytes(b"")
    assert latest_tf_checkpoint(str(d)) == str(d / "model_12")
    (d / "checkpoint").write_text('model_checkpoint_path: "model_3"\nall_model_checkpoint_paths: "model_3"\n')
    assert latest_tf_checkpoint(str(d)) == str(d / "model_3")


def test_shipped_chunk_config_is_the_reference_schema():
    """tensorflowasr_amd/configs/chunk_conformerS.yml: the five sub-model sections with the reference's keys and values
    (asr/configs/chunk_conformerS.yml), accepted by models.ChunkConformer"""
    from tensorflowasr_amd.config import load_yaml
    from tensorflowasr_amd.models import ChunkConformer
    c = load_yaml(os.path.join(ROOT, "tensorflowasr_amd", "configs", "chunk_conformerS.yml"))
    mc = c["model_config"]
    assert mc["name"] == "ChunkConformer" and set(mc) == {"name", "ChunkConformerFront", "ChunkConformerEncoder", "ChunkCTCPicker",
                                                          "ChunkCTCDecoder", "ContextHelper"}
    assert mc["ChunkConformerEncoder"]["num_blocks"] == 15 and mc["ChunkCTCDecoder"]["win_back"] == 8
    assert mc["ChunkConformerFront"]["chunk_num"] == 16 and mc["ChunkCTCPicker"]["num_classes"] == 277
    m = ChunkConformer(c, 277, 9171)
    assert m.count_params() > 12_000_000 and m._stream_cfg()[:3] == (16, 160, 4)


def test_typed_weight_loading_converts_on_the_host():
    """mi355asr_load_weight_typed: fp16 / bf16 / fp64 checkpoints"""
    lib, rc, p = _create(num_classes=20, ctc_num_blocks=1, has_encoder=0, num_blocks=0)
    assert rc == 0
    x = np.linspace(-3, 3, 144 * 20).reshape(144, 20)
    dims = (ctypes.c_int64 * 2)(144, 20)
    h = x.astype(np.float16)
    b16 = (x.astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
    for dt, arr in ((1, h), (2, b16), (3, x.astype(np.float64)), (0, x.astype(np.float32))):
        arr = np.ascontiguousarray(arr)
        assert lib.mi355asr_load_weight_typed(p, b"fully_connected/kernel", arr.ctypes.data_as(ctypes.c_void_p), dt, 2, dims) == 0
    assert lib.mi355asr_load_weight_typed(p, b"fully_connected/kernel", h.ctypes.data_as(ctypes.c_void_p), 9, 2, dims) == -1
    sub = np.array([6.1e-5, 5.96e-8, -2e-7, 0.0, 65504.0, np.inf], np.float16)     # subnormal halves, max, inf
    d1 = (ctypes.c_int64 * 1)(6)
    assert lib.mi355asr_load_weight_typed(p, b"nope", sub.ctypes.data_as(ctypes.c_void_p), 1, 1, d1) == -3
    lib.mi355asr_destroy(p)


def test_weight_shape_query_matches_the_python_mirror():
    """mi355asr_weight_shape: what a non-Python caller uses to enumerate the tensors (examples/asr_session.cpp)"""
    lib, rc, p = _create(num_classes=1332, ctc_num_blocks=1)
    assert rc == 0
    from tensorflowasr_amd.models import ConformerCTC
    expect = {n: tuple(s) for n, s in ConformerCTC(1332, num_blocks=1)._names_and_shapes()}
    rank = ctypes.c_int32()
    dims = (ctypes.c_int64 * 8)()
    seen = {}
    for i in range(lib.mi355asr_num_weights(p)):
        assert lib.mi355asr_weight_shape(p, i, ctypes.byref(rank), dims, 8) == 0
        seen[lib.mi355asr_weight_name(p, i).decode()] = tuple(dims[k] for k in range(rank.value))
    assert seen == expect
    assert lib.mi355asr_weight_shape(p, 10 ** 6, ctypes.byref(rank), dims, 8) == -1
    assert lib.mi355asr_weight_shape(p, 0, ctypes.byref(rank), dims, 1) == -1      # conv / DFT kernels have rank 4
    lib.mi355asr_destroy(p)


def test_cpp_session_example_builds_and_fails_loudly_without_a_gpu():
    """examples/asr_session.cpp (the reference's C++ Session on the C ABI) compiles against include/mi355asr.h and links
    to libmi355asr.so; without a device it must exit non-zero with the HIP error, not fall back to anything."""
    import subprocess
    from tensorflowasr_amd import build as b
    exe = b.build_example(verbose=False)
    assert exe and os.path.exists(exe)
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    out = subprocess.run([exe, "1"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 1 and "error:" in out.stderr


def test_chunk_checkpoint_keys_map_onto_every_chunk_tensor():
    """checkpoint.chunk_checkpoint_keys_to_abi: object-graph attribute paths of the reference's ChunkConformer (fixed by
    its source) -> the names the chunk handle expects.  The keys are constructed here from the attribute names in
    chunk_conformer_blocks.py; every tensor of the handle must be reached exactly once."""
    from tensorflowasr_amd import checkpoint
    inv_root = {"front": "front", "encoder": "encoder", "picker": "phone_picker", "decoder": "decoder", "helper": "helper"}
    inv_blk = {"ff_module_1": "ffm1", "ff_module_2": "ffm2", "mhsa_module": "mhsam", "conv_module": "convm"}
    inv_mha = {"query": "_query_dense", "key": "_key_dense", "value": "_value_dense", "attention_output": "_output_dense"}
    inv_mel = {"real_kernels": "dft_real_kernels", "imag_kernels": "dft_imag_kernels", "freq2mel": "freq2mel"}
    abi_names = sorted(co.chunk_weights(dict(co.CHUNK_S), seed=0))
    keys = {}
    for n in abi_names:
        p = n.split("/")
        root = inv_root[p[0]]
        if p[0] == "front":
            k = "front/mel_layer/" + inv_mel[p[2]] if p[1] == "mel_layer" else "/".join(["front"] + p[1:])
        elif p[1] in ("project", "fully_connected"):
            k = "%s/%s/%s" % (root, "project" if p[1] == "project" else "fc", p[2])
        else:
            idx = p[1].rsplit("_", 1)[1]
            lst = "conformer_blocks" if p[0] == "encoder" else "decode_layers"
            if p[2] == "ln":
                tail = ["ln", p[3]]
            elif p[2] == "mhsa_module" and p[3] == "mha":
                tail = ["mhsam", "mha", inv_mha[p[4]], p[5]]
            else:
                tail = [inv_blk[p[2]]] + p[3:]
            k = "/".join([root, lst, idx] + tail)
        keys[k + "/.ATTRIBUTES/VARIABLE_VALUE"] = n
    extra = ["optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE", "encoder/conformer_blocks/0/ffm1/ffn1/kernel/.OPTIMIZER_SLOT/optimizer/m/.ATTRIBUTES/VARIABLE_VALUE",
             "helper/sample_helper/embeddings/.ATTRIBUTES/VARIABLE_VALUE", "_CHECKPOINTABLE_OBJECT_GRAPH"]
    m = checkpoint.chunk_checkpoint_keys_to_abi(list(keys) + extra)
    assert m == keys


@pytest.mark.parametrize("tag,dom,dom_kernel", [("r01h", "tail_ff2", "tail_ff2_ring_kernel"), ("r02a", "tail_ff1", "tail_ff1_ld_kernel"),
                                                ("r02b", "tail_ff1", "tail_ff1_ld_kernel"), ("r02c", "tail_ff1", "tail_ff1_ld_kernel"), ("r02d", "tail_ff1", "tail_ff1_ld_kernel"), ("r02e", "tail_ff1", "tail_ff1_ld_kernel")])
def test_profile_artifacts_and_kernel_categories(tag, dom, dom_kernel):
    """the committed rocprofv3 summary of a round's final build names every kernel of the step, tools/summarize_rocprof.py
    files each of them under a category that bench.py's kernel table knows, and the rocprof duration of the dominant
    kernel agrees with the HIP-event duration in the bench line"""
    import csv
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from summarize_rocprof import category
    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv"))))
    names = [r["Name"] for r in rows]
    cats = {category(n) for n in names}
    assert {"stft", "mel", "subconv", "sublinear", "ff1_qkv", "attention", "out_glu", "dwconv", "tail_ff2", "ctc_head",
            "collapse", dom} <= cats
    assert all("(" not in c for c in cats), sorted(c for c in cats if "(" in c)      # no kernel left uncategorised
    assert cats <= set(_lib.KERNEL_NAMES), cats - set(_lib.KERNEL_NAMES)
    bench = json.loads(open(os.path.join(ROOT, "profiles", tag + "_bench_n1.json")).read().strip().splitlines()[-1])
    assert bench["unit"] == "audio-frames/s" and bench["n_gpus"] == 1 and bench["dtype"] == "f32"
    assert set(bench["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert set(bench["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert bench["roofline"]["kernel"] == dom
    if tag != "r01h":
        assert set(bench["cpu_baseline"]) >= {"threads1", "all_cores", "published_tf2_1core"}
        assert set(bench["roofline"]) >= {"traffic_source", "hbm_frac", "algorithmic_bytes"} and bench["h2d_inclusive"]
    by = {r["Name"]: float(r["AverageNs"]) / 1e3 for r in rows}
    t = next(v for k, v in by.items() if dom_kernel in k)
    assert abs(t - bench["kernels"][dom]["avg_ms"] * 1e3) / t < 0.15      # rocprof and HIP events agree


def test_round4_profile_artifacts_bench_line_and_library_reported_schemes():
    """profiles/r04_*: the block is attention + ONE pair-pipelined launch (no out_glu, no dwconv category left), every kernel of
    the trace has a category, the bench line carries roofline / cpu_baseline / exact_products / latency_b1 and, per kernel, the
    operand scheme the LIBRARY reported; rocprofv3 and the HIP events agree on the dominant kernel."""
    import csv
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from summarize_rocprof import category
    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", "r04_kernel_stats.csv"))))
    cats = {category(r["Name"]) for r in rows}
    assert {"stft", "mel", "subconv", "sublinear", "ff1_qkv", "attention", "tail_ff1", "tail_ff2", "ctc_head", "collapse"} <= cats
    assert "out_glu" not in cats and "dwconv" not in cats
    assert cats <= set(_lib.KERNEL_NAMES), cats - set(_lib.KERNEL_NAMES)
    calls = {category(r["Name"]): int(r["Calls"]) for r in rows}
    steps = calls["subconv"]
    assert calls["attention"] == 14 * steps and calls["tail_ff1"] == 12 * steps and calls["tail_ff2"] == 2 * steps and calls["ff1_qkv"] == 2 * steps
    bench = json.loads(open(os.path.join(ROOT, "profiles", "r04_bench_n1.json")).read().strip().splitlines()[-1])
    assert bench["unit"] == "audio-frames/s" and bench["n_gpus"] == 1 and bench["dtype"] == "f32" and bench["ms_per_step"] < 2.1
    assert set(bench["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"} and bench["roofline"]["kernel"] == "tail_ff1"
    assert set(bench["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    ex = bench["exact_products"]
    assert ex["ms_per_step"] > bench["ms_per_step"] and ex["max_abs_logit_diff_vs_default"] < 1e-3
    assert set(ex["schemes"].values()) <= {"f32", "bf16x3"} and "bf16x3" in ex["schemes"].values()      # no two-term kernel in that leg
    assert bench["latency_b1"]["ms"] < 1.3
    sch = {k: v["scheme"] for k, v in bench["kernels"].items()}
    assert sch["tail_ff1"] == sch["attention"] == sch["subconv"] == sch["sublinear"] == sch["stft"] == sch["ctc_head"] == "f16x2"
    assert bench["config3"]["ms_per_step"] < 1.4 and bench["config5"]["ms_predict"] < 2.7
    t = next(float(r["AverageNs"]) / 1e3 for r in rows if category(r["Name"]) == "tail_ff1")
    assert abs(t - bench["kernels"]["tail_ff1"]["avg_ms"] * 1e3) / t < 0.15


def test_round4_final_profile_artifacts_35_launches_and_config3_below_a_millisecond():
    """profiles/r04z_* (the round's last build): the subsampling Dense and the CTC projection are no kernels of their own any more
    and neither is the class head (35 launches per step, 38 before: counted from the trace), config 3 runs below a
    millisecond on the dmodel-256 chain kernel, which its trace shows."""
    import csv
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from summarize_rocprof import category
    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", "r04z_kernel_stats.csv"))))
    calls = {}
    for r in rows:
        calls[category(r["Name"])] = calls.get(category(r["Name"]), 0) + int(r["Calls"])
    assert "sublinear" not in calls and "ctc_project" not in calls and "ctc_head" not in calls and "out_glu" not in calls and "dwconv" not in calls
    steps = calls["subconv"]
    per_step = {k: v // steps for k, v in calls.items() if k in _lib.KERNEL_NAMES}
    assert per_step["attention"] == 14 and per_step["tail_ff1"] == 12 and per_step["tail_ff2"] == 2 and per_step["ff1_qkv"] == 2
    assert sum(per_step.values()) == 35                # stft, utt_max, mel, subconv, 2 + 14 + 12 + 2 of the blocks, collapse
    bench = json.loads(open(os.path.join(ROOT, "profiles", "r04z_final_bench_n1.json")).read().strip().splitlines()[-1])
    assert bench["ms_per_step"] < 2.05 and bench["roofline"]["kernel"] == "tail_ff1" and bench["latency_b1"]["ms"] < 1.3
    assert "sublinear" not in bench["kernels"] and "ctc_project" not in bench["kernels"] and "ctc_head" not in bench["kernels"]
    assert bench["config3"]["ms_per_step"] < 0.95 and bench["config5"]["ms_predict"] < 2.7
    c3 = list(csv.DictReader(open(os.path.join(ROOT, "profiles", "r04z_config3_kernel_stats.csv"))))
    chain = [r for r in c3 if "chain256_bf16_kernel" in r["Name"]]
    assert len(chain) == 4 and all(float(r["AverageNs"]) < 60e3 for r in chain)          # FFModule / conv tail at 832 and at 16 640 rows
    assert not any("gemm_ring_kernel<1," in r["Name"] or "gemm_ring_kernel<5," in r["Name"] for r in c3)     # no per-layer FFN / conv-tail first layer left


def test_pair_pipelined_stream_generator_simulates_and_matches_the_committed_sources():
    """tools/gen_pp.py describes the pair-pipelined fragment stream of fused_pp.hip once; its simulator replays every unit
    (fragment reads into pool slots, counted lgkmcnt waits, MFMAs, ring-slot hand-overs) and the committed device code
    (pp_units.inc) and host packing table (pp_layout.inc) are exactly what it emits."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_pp", os.path.join(root, "tools", "gen_pp.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    units, leads = g.build_all()                       # raises on any slot / count / coverage error
    assert {u.name for u in units} == {"A", "AP", "F", "BP", "B", "S"}
    by = {u.name: u for u in units}
    # the two-term fp16 scheme: three products per fragment pair (the three-term bf16 stream had 114 / 60 / 54 MFMAs)
    assert g.TERMS == 2 and (by["F"].nm, by["A"].nm, by["B"].nm) == (57, 30, 27) and by["F"].nslabs == 2
    assert all(u.length % g.NPOOL == 0 and max(len(s) for s in u.slabs) <= g.SLOT for u in units)
    assert min(leads.values()) >= 5                    # no fragment is requested less than 5 MFMAs before its first use
    csrc = os.path.join(root, "tensorflowasr_amd", "csrc")
    assert open(os.path.join(csrc, "pp_units.inc")).read() == g.emit_units(units) + "\n"
    assert open(os.path.join(csrc, "pp_layout.inc")).read() == g.emit_layout(units)
    # a chain of P hidden pairs = units A, AP, (P - 2) x F, BP, B = 2 P ring slots; every MFMA of the FFN (18 pairs) once
    assert by["A"].nm + by["AP"].nm + 16 * by["F"].nm + by["BP"].nm + by["B"].nm == 1026


def test_device_beam_search_arithmetic_is_the_host_c_librarys(tmp_path):
    """csrc/refmath.h restates glibc's expf / logf / log (the functions that DEFINE the reference decoder's float scores:
    decoder_utils.h:41-49, ctc_beam_search_decoder.cpp:57-59) for the device search.  tools/refmath_check.cpp compiles the
    same header for the host and compares it with the installed libm on the ranges the search uses -- every 5th float here
    (all of [1, 2] for logf at stride 1 costs nothing more), every float with MI355ASR_REFMATH_FULL=1 (17 s on 8 cores;
    measured round 4: 0 differences in 1 099 694 081 + 8 388 609 + 1 065 353 217 arguments).  A different C library
    (other glibc, no FMA unit) fails here first, before the device search silently stops matching the host search."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "refmath_check")
    subprocess.check_call(["g++", "-O2", "-mfma", "-ffp-contract=off", "-std=c++17", "-pthread",
                           os.path.join(root, "tools", "refmath_check.cpp"), "-o", exe])
    stride = "1" if os.environ.get("MI355ASR_REFMATH_FULL") == "1" else "5"
    out = subprocess.run([exe, stride], capture_output=True, text=True, timeout=1200)
    print(out.stdout)
    lines = [ln.split() for ln in out.stdout.strip().splitlines()]
    assert len(lines) == 4 and all(ln[-2] == "differ" for ln in lines)
    assert out.returncode == 0 and all(int(ln[-1]) == 0 for ln in lines), out.stdout
    assert all(int(ln[-3]) > 1_000_000 for ln in lines)
    # the tables the header uses are the installed library's (tools/libm_tables.py regenerates them byte for byte)
    inc = os.path.join(root, "tensorflowasr_amd", "csrc", "refmath_tables.inc")
    if os.path.exists("/lib/x86_64-linux-gnu/libm.so.6"):
        import sys
        again = str(tmp_path / "tables.inc")
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "libm_tables.py"), "--out", again], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert open(inc).read() == open(again).read()


def test_host_fp16_rounding_of_the_two_term_weight_packs_equals_numpy():
    """api.hip's f16_rne / f16_to_float (the hi and lo terms of the two-term fp16 weight packs are rounded on the host):
    normal and subnormal values, overflow, exact ties between neighbouring halves -- bit-equal to NumPy's float16."""
    import ctypes
    from tensorflowasr_amd import _lib
    lib = _lib.lib()
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.standard_normal(50000).astype(np.float32) * np.float32(s) for s in (1e-8, 1e-6, 1e-4, 1e-2, 1, 100, 30000, 70000)]
                       + [np.array([0, -0.0, 65504, 65519.99, 65520, 6.1e-5, 6.09e-5, 5.96e-8, 2.98e-8, 2.99e-8, 1e-9, np.inf, -np.inf], np.float32)])
    h = np.arange(0, 0x7bff, dtype=np.uint16).view(np.float16).astype(np.float64)
    ties = ((h[:-1] + h[1:]) / 2).astype(np.float32)
    x = np.ascontiguousarray(np.concatenate([x, ties, -ties]))
    bits = np.zeros(x.size, np.uint16)
    back = np.zeros(x.size, np.float32)
    fn = lib.mi355asr_test_f16_rne
    fn.restype = None
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
    fn(x.ctypes.data, x.size, bits.ctypes.data, back.ctypes.data)
    with np.errstate(over="ignore"):
        ref = x.astype(np.float16)
    assert np.array_equal(bits, ref.view(np.uint16))
    assert np.array_equal(back, ref.astype(np.float32))


def test_two_term_fp16_scheme_is_as_close_to_fp64_as_the_six_product_bf16_scheme_numpy_model():
    """The operand schemes of DESIGN.md section 2 restated in NumPy on one FFN-sized layer (512 x 144 x 576, fp64 products and
    sums, so only the operand representation and the dropped term pairs show): three bf16 terms / six products, and two
    fp16 terms (round-to-nearest hi + lo of the operand times the power of two that puts its bound in [2^13, 2^14) -- the
    kernels' pp_pow2_scale) / three products.  Both land within 4e-7 of the fp64 result (a plain fp32 matmul: 2e-6, its
    accumulation); without the scales the lo terms of small weights go subnormal and the scheme is five times worse."""
    rng = np.random.default_rng(0)
    M, K, N = 512, 144, 576
    x = (rng.standard_normal((M, K)) * 1.3 + 0.1).astype(np.float32)
    lim = np.sqrt(6 / (K + N))
    W = rng.uniform(-lim, lim, (K, N)).astype(np.float32)
    ref = x.astype(np.float64) @ W.astype(np.float64)

    def bf16_terms(a):
        r, out = a.astype(np.float32).copy(), []
        for _ in range(3):
            hi = (r.view(np.uint32) & 0xFFFF0000).view(np.float32)
            out.append(hi.astype(np.float64))
            r = (r - hi).astype(np.float32)
        return out

    def pow2_scale(bound):                      # fused_pp.hip: pp_pow2_scale
        e = (np.float32(bound).view(np.uint32) >> 23) & 255
        return np.float32(2.0) ** np.clip(140 - e.astype(np.int64), -14, 15)

    def fp16_terms(a, scale):
        v = (a.astype(np.float32) * scale).astype(np.float32)
        hi = v.astype(np.float16)
        lo = (v - hi.astype(np.float32)).astype(np.float16)
        return hi.astype(np.float64), lo.astype(np.float64)

    xa, wa = bf16_terms(x), bf16_terms(W)
    six = sum(xa[i] @ wa[j] for i in range(3) for j in range(3) if i + j <= 2)
    sx = pow2_scale(np.abs(x).max(1, keepdims=True))            # one scale per row (token)
    sw = pow2_scale(np.float32(np.abs(W).max()))
    xh, xl = fp16_terms(x, sx)
    wh, wl = fp16_terms(W, sw)
    assert np.abs(xh).max() < 2 ** 14 and np.abs(wh).max() < 2 ** 14
    three = (xh @ wh + xh @ wl + xl @ wh) / (sx.astype(np.float64) * float(sw))
    xh1, xl1 = fp16_terms(x, np.float32(1))
    wh1, wl1 = fp16_terms(W, np.float32(1))
    unscaled = xh1 @ wh1 + xh1 @ wl1 + xl1 @ wh1
    e6, e3, e1 = (np.abs(v - ref).max() for v in (six, three, unscaled))
    e32 = np.abs((x @ W).astype(np.float64) - ref).max()
    assert e6 < 4e-7 and e3 < 4e-7 and e3 < 1.5 * e6 + 1e-8
    assert e32 > 3 * e3 and e1 > 3 * e3


# the environment switches the library reads (each once per process), what each selects, and the test that runs it.
# Round 3 ended with 41, many of them losers of finished experiments; round 4 deleted those together with their kernels.
SWITCHES = {
    # arithmetic: two fp16 terms (default where a bound or a row maximum is available) -> three exact bf16 terms -> fp32 MFMA
    "MI355ASR_PP": "0: the round-2 loader-wave kernels on three bf16 terms instead of the pair-pipelined two-term ones | test_opt_in_kernel_variants, bench.py exact_products",
    "MI355ASR_PP_OUTGLU": "0: three-term out_glu_ld_kernel | test_opt_in_kernel_variants, test_two_term_fp16_block_and_attention_under_adversarial_operand_bounds",
    "MI355ASR_PP_HEAD": "0: three-term head_ld_kernel | test_opt_in_kernel_variants",
    "MI355ASR_PP_SUBLINEAR": "0: three-term sublinear_split_ld_kernel | test_opt_in_kernel_variants",
    "MI355ASR_SUBCONV_TERMS": "3: three-term subsampling conv; 22: two-term also for caller-supplied features | test_two_term_fp16_subsampling_conv_...",
    "MI355ASR_ATTN_TERMS": "3: three-term attention_split_kernel | test_opt_in_kernel_variants, adversarial test",
    "MI355ASR_FFT_TERMS": "3: three-term STFT | test_opt_in_kernel_variants",
    "MI355ASR_LEAF_TERMS": "0: fp32 LEAF Gabor convolution | test_leaf_*",
    "MI355ASR_SUBCONV_F32": "1: fp32-MFMA subsampling conv | test_opt_in_kernel_variants",
    "MI355ASR_SUBLINEAR_SPLIT": "0: fp32 subsampling Dense; 2: split kernels for any row count | test_opt_in_kernel_variants",
    "MI355ASR_FF1QKV_RING": "0: fp32-MFMA ff1_qkv_kernel | test_opt_in_kernel_variants",
    "MI355ASR_TAILFF2_RING": "0: fp32-MFMA tail_ff2_kernel | test_opt_in_kernel_variants",
    "MI355ASR_OUTGLU_SPLIT": "0: fp32-MFMA out_glu_kernel | test_opt_in_kernel_variants",
    "MI355ASR_HEAD_RING": "0: fp32-MFMA class head | test_opt_in_kernel_variants",
    "MI355ASR_ATTN_SPLIT": "0: fp32-MFMA attention_lds_kernel | test_opt_in_kernel_variants",
    "MI355ASR_ATTN_LDS": "0: online-softmax attention_kernel (K / V from L2) | test_opt_in_kernel_variants",
    "MI355ASR_FFT_SPLIT": "0: fp32-MFMA Cooley-Tukey STFT | test_opt_in_kernel_variants",
    "MI355ASR_FFT": "0: dense DFT GEMM | test_opt_in_kernel_variants",
    "MI355ASR_MEL_BAND": "0: dense mel GEMM | test_opt_in_kernel_variants",
    # structure: what is folded into which launch
    "MI355ASR_FUSED": "0: one launch per layer (dmodel 144) | test_opt_in_kernel_variants",
    "MI355ASR_TAIL_FF1": "0: tail_ff2 and the next block's ff1_qkv as separate launches | test_opt_in_kernel_variants",
    "MI355ASR_PP_DW": "0: depthwise conv as its own launch | test_opt_in_kernel_variants",
    "MI355ASR_PP_OGF": "0: out-projection + GLU as its own launch | test_block_as_two_launches_equals_the_three_launch_path_bit_for_bit",
    "MI355ASR_CHAIN256": "0: bf16 mode, dmodel 256: one launch per dense layer instead of chain256_bf16_kernel | test_bf16_chain256_against_layer_at_a_time",
    "MI355ASR_ATTN64_SPLIT": "0: head size 64: the fp32-MFMA attention kernels instead of the two-term attention_split64_kernel | test_head_size_64_two_term_attention_against_the_fp32_mfma_kernels",
    "MI355ASR_ATTN_LONG": "0: more than 256 keys on the fp32-MFMA attention kernels instead of attention_split_long_kernel (key blocks with an online softmax on the two-term pipe) | test_long_utterances_attention_in_key_blocks_against_the_fp32_kernels_and_the_oracle",
    "MI355ASR_NS1_MAX_M": "n: dmodel 144: blocks of up to n rows (default 4096; 0: never) on the one-tile-per-workgroup kernels of fused_ns.hip (small batches: eight waves share a 16-token tile) instead of the pair-pipelined ones | test_small_batches_one_tile_per_workgroup_against_the_pair_pipelined_kernels_and_the_oracle",
    "MI355ASR_NS1_ATTN": "0: the attention of a small-batch block (one-tile-per-workgroup kernels) as its own launch instead of inside the out-projection launch | test_small_batches_one_tile_per_workgroup_against_the_pair_pipelined_kernels_and_the_oracle",
    "MI355ASR_STREAM256": "0: bf16 mode, dmodel 256, chunks of <= 16 rows: one launch per layer / module instead of the whole block stack in stream256_kernel | test_streaming_block_stack_in_one_launch_vs_layer_at_a_time_and_rounding_oracle",
    "MI355ASR_CHAIN256_RT": "1 / 2 / 4 / 5: row tiles per workgroup of chain256_bf16_kernel (default by row count) | test_bf16_chain256_against_layer_at_a_time",
    "MI355ASR_SUBCONV_RT": "1 / 2: row tiles per wave of the two-term subsampling conv (default: one while that gives no CU a second workgroup) | test_two_term_subsampling_conv_one_row_tile_per_wave_bit_identical",
    "MI355ASR_SUBCONV_C1M": "0: conv1 of the two-term subsampling conv on the VALU in fp32 instead of on the matrix pipe | test_subsampling_conv1_on_the_matrix_pipe_against_the_valu_evaluation",
    "MI355ASR_QKV_HEAD_MAJOR": "0: q / k / v as token-major [B T, 3 D] rows instead of head-major planes | test_head_major_qkv_bit_identical_to_token_major",
    "MI355ASR_ATTN_BAND_LDS": "0: band attention with K / V straight from L2 (no staged window) | test_band_attention_staged_window_bit_identical",
    "MI355ASR_PP_HEADF": "0: the class head as its own launch instead of in the CTC block's tail launch | test_layer_in_front_of_a_block_in_its_first_launch_bit_for_bit",
    "MI355ASR_PP_PRE": "0: subsampling Dense and CTC projection as their own launches | test_layer_in_front_of_a_block_in_its_first_launch_bit_for_bit",
    "MI355ASR_GEMM16": "1: layer-at-a-time gemm16 kernels for every row count | test_bf16 / gemm16 tests",
    "MI355ASR_GEMM_RING": "0: dmodel 256 / 512 without the slab-ring GEMMs | test_ring_gemm_at_16000_rows_...",
    "MI355ASR_SMALL_M": "rows up to which the layer-at-a-time kernels run (default 48) | test_fused_block_path_at_short_utterances",
    "MI355ASR_RING_MIN_M": "rows from which gemm_ring runs | ring tests",
    "MI355ASR_RING_RT": "forces the ring GEMM's row tiles per wave (tests) | ring tests",
    "MI355ASR_GEMM256": "0: dense layers of dmodel 256 in bf16 mode stay on the ring kernel at many rows (gemm256_bf16_kernel off) | config 3 tests",
    "MI355ASR_GEMM256_MIN_M": "rows from which gemm256_bf16_kernel takes a K = 256 layer (default 8192) | config 3 tests",
    "MI355ASR_RING_SLOTS": "forces the ring depth (tests) | ring tests",
    "MI355ASR_RING_CPW": "forces column chunks per workgroup (tests) | ring tests",
    "MI355ASR_PP_HEAD_RANGES": "forces the number of class ranges of the two-term class head (1: one workgroup per row tile) | head tests",
    "MI355ASR_RING_HEAD_RANGES": "forces the number of class ranges of the ring class head (1: one workgroup per row tile) | ring tests",
    "MI355ASR_TOPN_REG": "0: LDS top-n kernel instead of the register-resident one | test_prefix_beam_topn_*",
    "MI355ASR_BEAM_DEVICE": "0: prefix search on host threads | test_prefix_beam_device_path_matches_reference_kats",
    # diagnostics (timing-only kernels need -DMI355ASR_DIAG_KERNELS; results are wrong by construction)
    "MI355ASR_BEAM_PROF": "1: clock counters of the device search | tools/r03_beamprof.py",
    "MI355ASR_PP_DIAG": "timing-only variants of pp_block_kernel (diag build) | profiles/r03_pp_experiments.md",
    "MI355ASR_SUBCONV_DIAG": "timing-only variants of the subsampling kernel (diag build) | profiles/r03_pp_experiments.md",
    "MI355ASR_HEAD_NOSTORE": "class head without its logit stores (diag build) | profiles",
}


def test_environment_switches_are_the_documented_ones():
    """no switch without a line in SWITCHES (and in DESIGN.md), no line without a switch: the switchboard cannot grow silently"""
    csrc = os.path.join(ROOT, "tensorflowasr_amd", "csrc")
    found = set()
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h", ".inc")):
            found |= set(re.findall(r'(?:getenv|env_on|mi355_env)\("(MI355ASR_[A-Z0-9_]+)"', open(os.path.join(csrc, f)).read()))
    assert found == set(SWITCHES), (sorted(found - set(SWITCHES)), sorted(set(SWITCHES) - found))
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    missing = [k for k in SWITCHES if k not in design]
    assert not missing, missing

