"""GPU parity tests (run with `-m gpu` on an MI355X): every call goes through the C ABI of libmi355asr.so and is
compared with the CPU oracle / the committed golden fixtures.  Tolerance contract (BASELINE.json north_star):
fp32 outputs within 1e-3 absolute of the reference forward, CTC-greedy token ids identical; integer paths
(argmax of given logits, greedy collapse) bit-exact.  Observed errors on the first MI355X run are ~1e-5
(profiles/r01_stage_report_first_gpu_run.json)."""
import ctypes
import json
import os

import numpy as np
import pytest

from helpers import (GOLDEN, argmax_mismatch_report, assert_frames_and_ids, assert_own_argmax, chunk_config_dict, co, encoder_kwargs, golden_ctc_io,
                     golden_ctc_weights, maxdiff, small_cfg, waves)

pytestmark = pytest.mark.gpu
TOL = 1e-3


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


@pytest.fixture(scope="module")
def enc2(torch_cuda):
    from tensorflowasr_amd.models import ConformerEncoder
    cfg = small_cfg(2)
    w = co.encoder_weights(cfg, seed=0)
    e = ConformerEncoder(**encoder_kwargs(cfg))
    e.load_weights(w, by_name=False)
    return e, w, cfg


@pytest.fixture(scope="module")
def full_s(torch_cuda):
    """ConformerCTC(S), 13+1 blocks, synthetic encoder + synthetic CTC head (non-blank outputs)."""
    from tensorflowasr_amd.models import ConformerCTC
    cfg = dict(co.CONFORMER_S)
    w = co.encoder_weights(cfg, seed=0)
    w.update(co.ctc_decoder_weights(cfg, 1332, seed=1))
    m = ConformerCTC(1332, **{k: v for k, v in encoder_kwargs(cfg).items() if k != "mel_layer_type"})
    m.load_weights(w, by_name=False)
    return m, w, cfg


def test_native_library_is_loaded(torch_cuda):
    from tensorflowasr_amd import _lib
    assert os.path.basename(_lib.LIB_PATH) == "libmi355asr.so" and os.path.exists(_lib.LIB_PATH)
    maps = open("/proc/self/maps").read()
    _lib.lib()
    maps = open("/proc/self/maps").read()
    assert "libmi355asr.so" in maps


@pytest.mark.parametrize("L,scale", [(32000, 1.0), (67263, 0.05), (1000, 1.0), (16160, 1.0), (32000, 1e-3), (32000, 3e-5), (32000, 1e-7)])
def test_melspectrogram_parity(enc2, L, scale):
    """scales down to 1e-7: the two-term STFT normalises every operand column by the power of two of its own maximum, with no
    floor on how quiet the column may be (the dB features are relative to the utterance's own maximum: a quiet recording has
    the same features as a loud one until the power falls under the reference's amin)"""
    e, w, _ = enc2
    x = waves(2, L, 5) * np.float32(scale)
    ref = co.melspectrogram(x.astype(np.float64), w)
    got = e.melspectrogram(x).cpu().numpy()
    assert got.shape == ref.shape
    assert maxdiff(got, ref) < TOL


def test_melspectrogram_with_a_dense_trained_filterbank(torch_cuda):
    """backend.mel()'s triangular filters run through the banded kernel; a freq2mel whose filters have wide support (the
    layer is trainable: mel_layer_trainable) must take the dense GEMM -- and a banded one with a few wide filters too."""
    from tensorflowasr_amd.models import ConformerEncoder
    cfg = small_cfg(1)
    x = waves(2, 16000, 41)
    rng = np.random.default_rng(3)
    base = co.encoder_weights(cfg, seed=2)
    for kind in ("dense", "one_wide_filter", "banded"):
        w = dict(base)
        f2m = base["mel_layer/freq2mel"].copy()
        if kind == "dense":
            f2m = (f2m + 1e-3 * rng.standard_normal(f2m.shape)).astype(np.float32)
        elif kind == "one_wide_filter":
            f2m[100:300, 7] = 0.01
        w["mel_layer/freq2mel"] = f2m
        e = ConformerEncoder(**encoder_kwargs(cfg))
        e.load_weights(w, by_name=False)
        got = e.melspectrogram(x).cpu().numpy()
        ref = co.melspectrogram(x.astype(np.float64), w)
        assert maxdiff(got, ref) < TOL, kind


def test_melspectrogram_silence_and_padding_dependence(enc2):
    """dB max-normalisation makes features depend on the (zero-padded) utterance as a whole; all-zero input hits
    the amin floor everywhere (max-normalised to 0 dB)."""
    e, w, _ = enc2
    z = np.zeros((1, 8000), np.float32)
    got = e.melspectrogram(z).cpu().numpy()
    ref = co.melspectrogram(z.astype(np.float64), w)
    assert maxdiff(got, ref) < TOL and np.abs(got).max() < TOL
    x = waves(1, 16000, 2)
    xp = np.concatenate([x, np.zeros((1, 8000), np.float32)], 1)
    a = e.melspectrogram(x).cpu().numpy()
    b = e.melspectrogram(xp).cpu().numpy()
    refb = co.melspectrogram(xp.astype(np.float64), w)
    assert maxdiff(b, refb) < TOL
    assert maxdiff(a[:, :90], b[:, :90]) < TOL      # same max -> same interior frames


def test_stft_kernel_selection_and_dense_fallback(enc2, torch_cuda):
    """The DFT kernels are model variables: the analytic ones (backend.py:27-69) select the Cooley-Tukey kernel, a
    checkpoint that changed them (here: a Hamming-like window plus a detuned bin) must run the dense DFT GEMM with
    the kernels as loaded.  Both against the oracle's dense DFT."""
    from tensorflowasr_amd.models import ConformerEncoder
    e, w, cfg = enc2
    assert e._h.lib.mi355asr_stft_mode(e._h.ptr) == 1
    w2 = dict(w)
    n = np.arange(1024)
    ham = (0.54 - 0.46 * np.cos(2 * np.pi * n / 1024)) / np.maximum(0.5 - 0.5 * np.cos(2 * np.pi * n / 1024), 0.05)
    shp = w["mel_layer/real_kernels"].shape
    hcol = ham.astype(np.float32).reshape((-1,) + (1,) * (len(shp) - 1))
    re = w["mel_layer/real_kernels"] * hcol
    im = w["mel_layer/imag_kernels"] * hcol
    re[..., 37] *= 0.5
    w2["mel_layer/real_kernels"], w2["mel_layer/imag_kernels"] = re, im
    e2 = ConformerEncoder(**encoder_kwargs(cfg))
    e2.load_weights(w2, by_name=False)
    assert e2._h.lib.mi355asr_stft_mode(e2._h.ptr) == 0
    x = waves(2, 24000, 11)
    ref2 = co.melspectrogram(x.astype(np.float64), w2)
    got2 = e2.melspectrogram(x).cpu().numpy()
    assert maxdiff(got2, ref2) < TOL
    ref1 = co.melspectrogram(x.astype(np.float64), w)
    assert maxdiff(e.melspectrogram(x).cpu().numpy(), ref1) < TOL
    assert maxdiff(ref1, ref2) > 0.1          # the two kernel sets really give different features


def test_fft_and_dense_stft_agree(torch_cuda):
    """MI355ASR_FFT=0 forces the dense DFT GEMM on the analytic kernels: same features as the Cooley-Tukey path."""
    from tensorflowasr_amd.models import ConformerEncoder
    cfg = small_cfg(1)
    w = co.encoder_weights(cfg, seed=3)
    x = waves(3, 48000, 4)
    outs = []
    for flag, mode in (("0", 0), ("1", 1)):
        os.environ["MI355ASR_FFT"] = flag
        try:
            e = ConformerEncoder(**encoder_kwargs(cfg))
            e.load_weights(w, by_name=False)
        finally:
            os.environ.pop("MI355ASR_FFT")
        assert e._h.lib.mi355asr_stft_mode(e._h.ptr) == mode
        outs.append(e.melspectrogram(x).cpu().numpy())
    assert maxdiff(outs[0], outs[1]) < 5e-4      # each is ~2.4e-4 from the fp64 oracle on the dB scale


@pytest.mark.parametrize("F", [200, 50, 37, 3])
def test_conv_subsampling_parity(enc2, F):
    e, w, _ = enc2
    rng = np.random.default_rng(F)
    mel = (-80.0 * rng.random((3, F, 80))).astype(np.float32)
    ref = co.conv_subsampling(mel.astype(np.float64), w)
    got = e.conv_subsampling(mel).cpu().numpy()
    assert got.shape == ref.shape
    assert maxdiff(got, ref) < TOL


def test_two_term_subsampling_conv_one_row_tile_per_wave_bit_identical(torch_cuda):
    """subconv_split_ring_kernel<..., RTN = 1>: 128 positions per workgroup, one 16-position row tile per wave -- what the
    launcher picks while that gives no CU a second workgroup (one to four utterances per call: -2.4 % latency).  A position's sums
    run over the same steps in the same order whichever tile it sits in: BIT-IDENTICAL to MI355ASR_SUBCONV_RT=2, for dmodel
    144 / 256 / 512, utterance and streaming-chunk shapes (MI355ASR_SUBCONV_TERMS=22 sends the stage API's features through
    the two-term kernel)."""
    import subprocess
    import sys
    import tempfile
    code = r'''
import sys, numpy as np
sys.path.insert(0, "tests")
from helpers import co, encoder_kwargs, small_cfg
from tensorflowasr_amd.models import ConformerEncoder
out = {}
for name, base in (("S", co.CONFORMER_S), ("M", co.CONFORMER_M), ("L", co.CONFORMER_L)):
    cfg = small_cfg(1, base)
    w = co.encoder_weights(cfg, seed=61)
    e = ConformerEncoder(**encoder_kwargs(cfg))
    e.load_weights(w, by_name=False)
    rng = np.random.default_rng(62)
    for B, F in ((4, 123), (1, 1000), (24, 50)):
        mel = (-80.0 * rng.random((B, F, 80))).astype(np.float32)
        out["%s_%d_%d" % (name, B, F)] = e.conv_subsampling(mel).cpu().numpy()
np.savez(sys.argv[1], **out)
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for rt in ("1", "2"):
            f = os.path.join(td, rt + ".npz")
            r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, MI355ASR_SUBCONV_TERMS="22", MI355ASR_SUBCONV_RT=rt),
                               capture_output=True, text=True, timeout=900, cwd=root)
            assert r.returncode == 0, r.stderr[-3000:]
            res[rt] = dict(np.load(f))
    assert len(res["1"]) == 9
    for k in res["1"]:
        assert np.isfinite(res["1"][k]).all() and np.abs(res["1"][k]).max() > 0.1, k
        assert np.array_equal(res["1"][k], res["2"][k]), (k, float(np.abs(res["1"][k] - res["2"][k]).max()))


def test_two_term_fp16_subsampling_conv_against_the_three_term_kernel_and_the_oracle(torch_cuda):
    """subconv_split_ring_kernel<..., TM = 2>: conv1 values and conv2 kernel as hi + lo fp16 terms (power-of-two scales from
    the weights and the frontend's [-80, 0] dB range), three MFMAs per fragment pair instead of six.  The stage API feeds
    caller-supplied features, which take the three-term kernel; MI355ASR_SUBCONV_TERMS=22 (subprocess) sends them through
    the two-term one.  On features inside the bound both are equally far from the fp64 oracle (dmodel 144, 256, 512:
    one column chunk and chunks on grid.z), including rows at the edge of the bound and a batch of exact zeros."""
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, "tests")
from helpers import co, encoder_kwargs, maxdiff, small_cfg
from tensorflowasr_amd.models import ConformerEncoder
res = []
for base in (co.CONFORMER_S, co.CONFORMER_M, co.CONFORMER_L):
    cfg = small_cfg(1, base)
    w = co.encoder_weights(cfg, seed=61)
    e = ConformerEncoder(**encoder_kwargs(cfg))
    e.load_weights(w, by_name=False)
    rng = np.random.default_rng(62)
    mel = (-80.0 * rng.random((4, 123, 80))).astype(np.float32)
    mel[1, :, ::2] = -80.0                     # the bound itself
    mel[2] = 0.0
    mel[3] *= 1e-3                             # small features: the lo terms go subnormal
    ref = co.conv_subsampling(mel.astype(np.float64), w)
    got = e.conv_subsampling(mel).cpu().numpy()
    res += [maxdiff(got, ref), float(np.abs(ref).max())]
print("RESULT " + " ".join("%.4e" % v for v in res))
'''
    errs = {}
    for terms in ("3", "22"):
        env = dict(os.environ, MI355ASR_SUBCONV_TERMS=terms)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900,
                             cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT")]
        assert line, out.stderr[-2000:]
        errs[terms] = [float(v) for v in line[0].split()[1:]]
    print(errs)
    for i in range(0, 6, 2):
        e3, e2, scale = errs["3"][i], errs["22"][i], errs["3"][i + 1]
        assert e3 < TOL and e2 < TOL
        assert e2 <= 2.0 * e3 + 2e-7 * scale, (i, e3, e2, scale)   # the same distance from the oracle (fp32 accumulation noise)


def test_two_term_fp16_block_and_attention_under_adversarial_operand_bounds(torch_cuda):
    """The two-term kernels carry 2^-22 of a power-of-two BOUND, not of the value (round-3 verdict, Weak 4): where the bound is
    loose every factor of two costs a bit.  tools/adversarial_two_term.py builds weights that pull bound and value apart --
    every LayerNorm gamma x 30 with beta = 3, one ffn1 column x 50 (the hidden row's bound follows its L1 norm), gamma_i x 1000
    on a feature whose normalised input is ~0 (the static q / k / v bound counts 1000 |W_i| sqrt(143) the values never reach:
    bound / value >= 2^10) -- and runs the fused block (pp_block_kernel, pp_out_glu_kernel, attention_split_kernel<2>) against
    the fp64 oracle.  The same script under MI355ASR_PP=0 MI355ASR_PP_OUTGLU=0 MI355ASR_ATTN_TERMS=3 runs the three-term
    kernels (exact fp32 products).  The two-term result must be no further from the oracle than twice the exact-product
    kernels' distance (measured round 4: 8.1e-6 / 1.0e-5 / 7.0e-6 / 1.2e-4 / 1.7e-3 against 8.4e-6 / 1.3e-5 / 8.0e-6 / 9.1e-5 /
    1.3e-2: the combined case is ill-conditioned for ANY fp32 evaluation, the operand scheme is not what limits it)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for tag, extra in (("two", {}), ("three", {"MI355ASR_PP": "0", "MI355ASR_PP_OUTGLU": "0", "MI355ASR_ATTN_TERMS": "3"})):
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "adversarial_two_term.py")], env=dict(os.environ, **extra),
                             capture_output=True, text=True, timeout=900, cwd=root)
        rows = [ln.split() for ln in out.stdout.splitlines() if ln.startswith("CASE")]
        assert len(rows) == 5, out.stderr[-2000:]
        res[tag] = {r[1]: (float(r[2]), float(r[3])) for r in rows}
    print(res)
    for case in ("plain", "gamma30", "w1col50", "quiet1000", "all"):
        e2, scale = res["two"][case]
        e3, _ = res["three"][case]
        assert e2 <= 2.0 * e3 + 2e-7 * scale, (case, e2, e3)
        if case != "all":
            assert e2 < TOL and e3 < TOL, (case, e2, e3)


def test_subsampling_dense_two_term_stream_at_5000_rows(torch_cuda):
    """Dense(2880 -> 144) behind the subsampling convs runs from 4 096 rows on as pp_sublinear_kernel: two fp16 terms, the
    operand scale taken per (token, 144-wide chunk) from the chunk's own maximum (ReLU outputs have no static bound).  20
    utterances x 1 000 feature frames = 5 000 rows against the fp64 oracle: in-range features, one utterance of zeros (all-zero
    chunks), one of tiny features (1e-3 dB) and one whose conv output spans six orders of magnitude across the frequency axis."""
    from tensorflowasr_amd.models import ConformerEncoder
    cfg = small_cfg(1)
    w = co.encoder_weights(cfg, seed=71)
    e = ConformerEncoder(**encoder_kwargs(cfg))
    e.load_weights(w, by_name=False)
    rng = np.random.default_rng(72)
    mel = (-80.0 * rng.random((20, 1000, 80))).astype(np.float32)
    mel[3] = 0.0
    mel[7] *= 1e-3
    mel[11] *= np.logspace(-6, 0, 80, dtype=np.float32)[None, :]
    ref = co.conv_subsampling(mel.astype(np.float64), w)
    got = e.conv_subsampling(mel).cpu().numpy()
    assert got.shape == ref.shape == (20, 250, 144)
    err = maxdiff(got, ref)
    print("subsampling Dense at 5 000 rows: max|d| %.3g of max|ref| %.3g" % (err, np.abs(ref).max()))
    assert err < 2e-7 * np.abs(ref).max() + 1e-5


@pytest.mark.parametrize("B,T", [(2, 50), (3, 250), (1, 300), (2, 7), (1, 16), (1, 17), (1, 750), (2, 257), (1, 448), (1, 449), (3, 500)])
def test_conformer_block_parity(enc2, B, T):
    # T > 256 (round 6): attention_split_long_kernel -- key blocks of 224 with an online softmax: one key in the last block (257 = 224 +
    # 33 and 449 = 2 x 224 + 1), exactly full blocks (448), a ragged last query tile (500 = 31 tiles + 4 rows), 750 = four blocks
    e, w, _ = enc2
    rng = np.random.default_rng(B * 1000 + T)
    x = rng.standard_normal((B, T, 144)).astype(np.float32)
    ref = co.conformer_block(x.astype(np.float64), w, "conformer_block_1", 36)
    got = e.conformer_block(1, x).cpu().numpy()
    assert maxdiff(got, ref) < TOL


def test_block_as_two_launches_equals_the_three_launch_path_bit_for_bit(torch_cuda):
    """Round 4: out-projection + residual + LayerNorm + pw_conv_1 + GLU run in the prologue of the pair-pipelined tail kernel
    (pp_block_kernel<..., OGF>): six waves compute the 96 frames of the depthwise window, halo included, from the attention
    output.  Every row goes through the same units in the same order as in pp_out_glu_kernel, so the encoder output must be
    BIT-IDENTICAL to the build with MI355ASR_PP_OGF=0 (out_glu as its own launch) -- offline 'same' padding at T = 250 / 64 /
    65 / 127 / 200 (utterance lengths around the 64-frame tiles: partial last tiles, a halo tile entirely past the end) and
    the ChunkConformer's causal padding (two halo tiles in front)."""
    import subprocess
    import sys
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, "tests")
from helpers import chunk_config_dict, co, encoder_kwargs, small_cfg, waves
from tensorflowasr_amd.models import ChunkConformer, ConformerEncoder
out = {}
cfg = small_cfg(3)
w = co.encoder_weights(cfg, seed=3)
e = ConformerEncoder(**encoder_kwargs(cfg))
e.load_weights(w, by_name=False)
for B, T in ((9, 250), (13, 64), (13, 65), (7, 127), (5, 200)):
    x = np.random.default_rng(T).standard_normal((B, T, 144)).astype(np.float32)
    out["blk_%d" % T] = e.conformer_block(1, x).cpu().numpy()
x = waves(12, 80000, 3)
out["enc"] = e(x).cpu().numpy()
c5 = dict(co.CHUNK_S, enc_num_blocks=2, picker_num_blocks=1, helper_num_blocks=1, decoder_num_blocks=1)
w5 = co.chunk_weights(c5, seed=4)
m = ChunkConformer(chunk_config_dict(c5), c5["picker_num_classes"], c5["decoder_num_classes"])
m.load_weights(w5, by_name=False)
got = m.predict(waves(4, 160000, 9), stages=True)
out["chunk_enc"] = got["enc"].cpu().numpy()
out["chunk_text"] = got["text_logits"].cpu().numpy()
np.savez(sys.argv[1], **out)
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import tempfile
    res = {}
    with tempfile.TemporaryDirectory() as td:
        # (MI355ASR_NS1_MAX_M=0: this is a property of the pair-pipelined kernels; small shapes otherwise take the one-tile kernels of round 6)
        for tag, extra in (("two", {"MI355ASR_NS1_MAX_M": "0"}), ("three", {"MI355ASR_PP_OGF": "0", "MI355ASR_NS1_MAX_M": "0"})):
            f = os.path.join(td, tag + ".npz")
            r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=900, cwd=root)
            assert r.returncode == 0, r.stderr[-3000:]
            res[tag] = dict(np.load(f))
    for k in res["two"]:
        assert np.isfinite(res["two"][k]).all() and np.abs(res["two"][k]).max() > 0.1, k
        assert np.array_equal(res["two"][k], res["three"][k]), (k, float(np.abs(res["two"][k] - res["three"][k]).max()))


def test_folded_block_launches_on_grids_larger_than_the_chip_bit_for_bit(torch_cuda):
    """Round 5 (advisor, round-4 review): a folded launch reads x1 with the halo frames of its neighbours' tiles, so it must
    not store the block output into the buffer x1 lives in -- a workgroup scheduled after its neighbour finished would read y
    as x1.  One workgroup fits per CU (147 KB of LDS), so the hazard needs more than 256 workgroups and a launch that stores
    y: the last block of a chunk stack (B = 32 x 30 s: 12 tiles x 32 = 384 workgroups), a CTC decoder with two blocks
    (80 x 250 frames: 320 workgroups, the first block stores y for the second), and the encoder with MI355ASR_TAIL_FF1=0
    (every block stores y).  All of it BIT-IDENTICAL to MI355ASR_PP_OGF=0, where x1 is dead before y is written."""
    import subprocess
    import sys
    import tempfile
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, "tests")
from helpers import chunk_config_dict, co, encoder_kwargs, small_cfg, waves
from tensorflowasr_amd.models import ChunkConformer, ConformerCTC
out = {}
cfg = dict(small_cfg(2), ctcdecoder_num_blocks=2)
w = co.encoder_weights(cfg, seed=71)
w.update(co.ctc_decoder_weights(cfg, 200, seed=72))
m = ConformerCTC(200, ctcdecoder_num_blocks=2, **{k: v for k, v in encoder_kwargs(cfg).items() if k != "mel_layer_type"})
m.load_weights(w, by_name=False)
x = waves(80, 160000, 11)
enc = m.encode(x)
out["enc"] = enc.cpu().numpy()
out["logits"] = m.ctc_logits(enc).cpu().numpy()
xb = np.random.default_rng(5).standard_normal((80, 250, 144)).astype(np.float32)
out["blk"] = m.conformer_block(1, xb).cpu().numpy()
c5 = dict(co.CHUNK_S, enc_num_blocks=2, picker_num_blocks=1, helper_num_blocks=1, decoder_num_blocks=1)
w5 = co.chunk_weights(c5, seed=4)
mc = ChunkConformer(chunk_config_dict(c5), c5["picker_num_classes"], c5["decoder_num_classes"])
mc.load_weights(w5, by_name=False)
got = mc.predict(waves(32, 480000, 9), stages=True)
out["chunk_enc"] = got["enc"].cpu().numpy()
out["chunk_text"] = got["text_logits"].cpu().numpy()
np.savez(sys.argv[1], **out)
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for tag, extra in (("fold", {}), ("fold_noff1", {"MI355ASR_TAIL_FF1": "0"}), ("plain", {"MI355ASR_PP_OGF": "0"})):
            f = os.path.join(td, tag + ".npz")
            r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=1200, cwd=root)
            assert r.returncode == 0, r.stderr[-3000:]
            res[tag] = dict(np.load(f))
    for tag in ("fold", "fold_noff1"):
        for k in res["plain"]:
            assert np.isfinite(res[tag][k]).all() and np.abs(res[tag][k]).max() > 0.1, (tag, k)
            assert np.array_equal(res[tag][k], res["plain"][k]), (tag, k, float(np.abs(res[tag][k] - res["plain"][k]).max()))


def test_head_major_qkv_bit_identical_to_token_major(torch_cuda):
    """Round 5: between a pair-pipelined block kernel and attention_split_kernel<2> q / k / v travel as three head-major planes
    [B, H, T, 36] (a head's rows contiguous: the token-major rows put a head's 144 bytes at a 1728-byte stride = two 128-byte
    lines per row, 1.79 x the fetch, round-4 counters).  Only addresses change: encoder output, logits and ids BIT-IDENTICAL to
    MI355ASR_QKV_HEAD_MAJOR=0 -- flat token tiles that straddle utterances (first block's own launch: T = 250 is no multiple
    of 16), per-utterance tiles (the folded launches), T = 250 / 100 / 17, and a batch whose rows are not a multiple of 64."""
    import subprocess
    import sys
    import tempfile
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, "tests")
from helpers import co, encoder_kwargs, small_cfg, waves
from tensorflowasr_amd.models import ConformerCTC
cfg = small_cfg(3)
w = co.encoder_weights(cfg, seed=81)
w.update(co.ctc_decoder_weights(cfg, 200, seed=82))
m = ConformerCTC(200, **{k: v for k, v in encoder_kwargs(cfg).items() if k != "mel_layer_type"})
m.load_weights(w, by_name=False)
out = {}
for tag, B, L in (("a", 9, 160000), ("b", 21, 64000), ("c", 40, 10880)):
    x = waves(B, L, 3)
    enc = m.encode(x)
    out[tag + "_enc"] = enc.cpu().numpy()
    out[tag + "_logits"] = m.ctc_logits(enc).cpu().numpy()
    ids, lens = m.recognize(x)
    out[tag + "_ids"] = ids.cpu().numpy(); out[tag + "_lens"] = lens.cpu().numpy()
xb = np.random.default_rng(5).standard_normal((7, 250, 144)).astype(np.float32)
out["blk"] = m.conformer_block(1, xb).cpu().numpy()
np.savez(sys.argv[1], **out)
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for tag, extra in (("hm", {}), ("tm", {"MI355ASR_QKV_HEAD_MAJOR": "0"})):
            f = os.path.join(td, tag + ".npz")
            r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=900, cwd=root)
            assert r.returncode == 0, r.stderr[-3000:]
            res[tag] = dict(np.load(f))
    for k in res["tm"]:
        assert np.isfinite(res["hm"][k].astype(np.float64)).all(), k
        assert np.array_equal(res["hm"][k], res["tm"][k]), (k, float(np.abs(res["hm"][k].astype(np.float64) - res["tm"][k]).max()))
    assert np.abs(res["hm"]["a_enc"]).max() > 0.1


def test_subsampling_conv1_on_the_matrix_pipe_against_the_valu_evaluation(torch_cuda):
    """Round 5: in the two-term subsampling conv of dmodel 144 (and 256: the streaming configuration) conv1 itself runs on the matrix pipe -- the mel patch staged as fp16
    hi + lo planes, three MFMAs per (row tile, tap) -- instead of 36 fp32 multiply-adds per value on the VALU.  It is then a
    two-term product (2^-22 of its operand bounds) like conv2: against MI355ASR_SUBCONV_C1M=0 (exact fp32 conv1) and against the
    oracle on the encoder's own features (offline 'same' frontend: static mel bound) and on the chunk front (run-time bound per
    utterance, quiet and loud utterances in one batch): both builds within the usual distance of the oracle, C1M at most twice
    the VALU build's + 1e-6, and 2e-6 of the output's scale apart from each other."""
    import subprocess
    import sys
    import tempfile
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, "tests")
from helpers import chunk_config_dict, co, encoder_kwargs, small_cfg, waves
from tensorflowasr_amd.models import ChunkConformer, ConformerEncoder
out = {}
cfg = small_cfg(1)
w = co.encoder_weights(cfg, seed=91)
e = ConformerEncoder(**encoder_kwargs(cfg))
e.load_weights(w, by_name=False)
x = waves(6, 160000, 12)
x[1] *= np.float32(1e-3)
x[2, 40000:] = 0.0
out["sub"] = e(x).cpu().numpy()                      # one block behind the subsampling: its error travels with the features
cfgm = small_cfg(1, co.CONFORMER_M)                  # dmodel 256: two 128-channel chunks on grid.z, both evaluate conv1
em = ConformerEncoder(**encoder_kwargs(cfgm))
em.load_weights(co.encoder_weights(cfgm, seed=92), by_name=False)
out["subM"] = em(x).cpu().numpy()[:3]            # six utterances: two row tiles per wave (three alone: one, see launch_subconv_split)
out["subM1"] = em(x[:3]).cpu().numpy()
c5 = dict(co.CHUNK_S, enc_num_blocks=1, picker_num_blocks=1, helper_num_blocks=1, decoder_num_blocks=1)
w5 = co.chunk_weights(c5, seed=6)
mc = ChunkConformer(chunk_config_dict(c5), c5["picker_num_classes"], c5["decoder_num_classes"])
mc.load_weights(w5, by_name=False)
xc = waves(3, 48000, 75)
xc[0] *= np.float32(30.0); xc[1] *= np.float32(1e-3)
out["front"] = mc.predict(xc, stages=True)["front"].cpu().numpy()
np.savez(sys.argv[1], **out)
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for tag, extra in (("mm", {}), ("valu", {"MI355ASR_SUBCONV_C1M": "0"})):
            f = os.path.join(td, tag + ".npz")
            r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=900, cwd=root)
            assert r.returncode == 0, r.stderr[-3000:]
            res[tag] = dict(np.load(f))
    cfg = small_cfg(1)
    w = co.encoder_weights(cfg, seed=91)
    x = waves(6, 160000, 12)
    x[1] *= np.float32(1e-3)
    x[2, 40000:] = 0.0
    ref_enc = co.conformer_encoder(x[:3].astype(np.float64), w, cfg)
    cfgm = small_cfg(1, co.CONFORMER_M)
    ref_encm = co.conformer_encoder(x[:3].astype(np.float64), co.encoder_weights(cfgm, seed=92), cfgm)
    c5 = dict(co.CHUNK_S, enc_num_blocks=1, picker_num_blocks=1, helper_num_blocks=1, decoder_num_blocks=1)
    w5 = co.chunk_weights(c5, seed=6)
    xc = waves(3, 48000, 75)
    xc[0] *= np.float32(30.0)
    xc[1] *= np.float32(1e-3)
    ref_front = co.chunk_predict(xc.astype(np.float64), w5, c5)["front"]
    for key, ref, got in (("sub", ref_enc, lambda r: r["sub"][:3]), ("subM", ref_encm, lambda r: r["subM"]), ("subM1", ref_encm, lambda r: r["subM1"]),
                          ("front", ref_front, lambda r: r["front"])):
        scale = max(1.0, float(np.abs(ref).max()))
        e_mm, e_va = maxdiff(got(res["mm"]), ref), maxdiff(got(res["valu"]), ref)
        apart = maxdiff(got(res["mm"]), got(res["valu"]))
        print("%s: conv1 on the matrix pipe %.3g, on the VALU %.3g from the oracle; apart %.3g (scale %.3g)" % (key, e_mm, e_va, apart, scale))
        assert e_mm < TOL and e_va < TOL and e_mm <= 2 * e_va + 1e-6 * scale and apart < 2e-6 * scale + 1e-6


def test_layer_in_front_of_a_block_in_its_first_launch_bit_for_bit(torch_cuda):
    """Round 4: the subsampling Dense (conformer_blocks.py:102-106, K = 20 * 144) and the CTC decoder's projection
    (conformer_blocks.py:631) run in the prologue of the first block's ff_module_1 + qkv launch (pp_block_kernel<..., PRE>): the
    layer's two-term stream flows through the ring in front of the block's, x0 stays in registers.  Same units in the same
    order as pp_sublinear_kernel: encoder output, logits and token ids BIT-IDENTICAL to MI355ASR_PP_PRE=0 (layers as their
    own launches), at 5000 rows (row count not a multiple of 64: a partly idle last workgroup) and at 16 x 250 rows; the same for
    the ChunkConformer's predict (front Dense + the three stack projections); and both within the usual tolerance of the oracle."""
    import subprocess
    import sys
    import tempfile
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, "tests")
from helpers import co, encoder_kwargs, small_cfg, waves
from tensorflowasr_amd.models import ConformerCTC
cfg = small_cfg(2)
w = co.encoder_weights(cfg, seed=61)
w.update(co.ctc_decoder_weights(cfg, 300, seed=62))
m = ConformerCTC(300, **{k: v for k, v in encoder_kwargs(cfg).items() if k != "mel_layer_type"})
m.load_weights(w, by_name=False)
out = {}
for tag, B, L in (("a", 20, 160000), ("b", 17, 159000)):
    x = waves(B, L, 5)
    enc = m.encode(x)
    logits, amax = m.ctc_logits(enc, return_argmax=True)
    ids, lens = m.recognize(x)
    out[tag + "_enc"] = enc.cpu().numpy(); out[tag + "_logits"] = logits.cpu().numpy()
    out[tag + "_ids"] = ids.cpu().numpy(); out[tag + "_lens"] = lens.cpu().numpy()
# the ChunkConformer's offline predict: the front's Dense in the encoder's first block, the projections of the phone picker, the
# context helper and the text decoder in theirs (4 250 rows; picked frames: whatever the picker keeps)
from helpers import chunk_config_dict
from tensorflowasr_amd.models import ChunkConformer
c5 = dict(co.CHUNK_S, enc_num_blocks=2, picker_num_blocks=1, helper_num_blocks=1, decoder_num_blocks=1)
w5 = co.chunk_weights(c5, seed=4)
mc = ChunkConformer(chunk_config_dict(c5), c5["picker_num_classes"], c5["decoder_num_classes"])
mc.load_weights(w5, by_name=False)
got = mc.predict(waves(17, 160000, 9), stages=True)
out["chunk_enc"] = got["enc"].cpu().numpy()
out["chunk_text"] = got["text_logits"].cpu().numpy()
np.savez(sys.argv[1], **out)
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for tag, extra in (("folded", {}), ("own", {"MI355ASR_PP_PRE": "0"}), ("head_own", {"MI355ASR_PP_HEADF": "0"})):
            f = os.path.join(td, tag + ".npz")
            r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=900, cwd=root)
            assert r.returncode == 0, r.stderr[-3000:]
            res[tag] = dict(np.load(f))
    for k in res["folded"]:
        assert np.array_equal(res["folded"][k], res["own"][k]), (k, float(np.abs(res["folded"][k].astype(np.float64) - res["own"][k]).max()))
        # ... and the class head BEHIND the CTC decoder's last block, in that block's tail launch (MI355ASR_PP_HEADF=0: pp_head_kernel)
        assert np.array_equal(res["folded"][k], res["head_own"][k]), (k, float(np.abs(res["folded"][k].astype(np.float64) - res["head_own"][k]).max()))
    assert (res["folded"]["a_lens"] > 0).any()
    cfg = small_cfg(2)
    w = co.encoder_weights(cfg, seed=61)
    w.update(co.ctc_decoder_weights(cfg, 300, seed=62))
    x = waves(3, 160000, 5)
    enc_ref = co.conformer_encoder(x.astype(np.float64), w, cfg)
    assert maxdiff(res["folded"]["a_enc"][:3], enc_ref) < TOL
    assert maxdiff(res["folded"]["a_logits"][:3], co.ctc_decoder(enc_ref, w, cfg)) < TOL


@pytest.mark.parametrize("scale", [1e-4, 1.0, 1e3, 3e4])
def test_conformer_block_operand_scales_follow_the_input_magnitude(enc2, scale):
    """The two-term fp16 kernels scale every operand row by the power of two of its own largest magnitude (and hidden rows by
    a bound derived from it): a block input of any magnitude -- the fused path, 2 000 tokens -- stays within the tolerance of
    the fp64 oracle, relative to the size of the output (the block ends in a LayerNorm: O(1))."""
    e, w, _ = enc2
    rng = np.random.default_rng(int(np.log10(scale) * 10) + 50)
    x = (rng.standard_normal((8, 250, 144)) * scale).astype(np.float32)
    x[3, 100:120] = 0.0                                # rows of exact zeros
    x[5, :, 7] *= 50.0                                 # one dominant feature: the other operands of the row lose their lo terms
    ref = co.conformer_block(x.astype(np.float64), w, "conformer_block_1", 36)
    got = e.conformer_block(1, x).cpu().numpy()
    assert np.isfinite(got).all()
    assert maxdiff(got, ref) < TOL


def test_conformer_block_is_batch_size_invariant_per_path(enc2):
    """The block runs through one of three kernel families depending on the row count (layer-at-a-time for a handful of rows,
    round 6: one 16-token tile per workgroup up to 4 096 rows, the pair-pipelined kernels above): inside a family an utterance's
    result does not depend on what else is in the batch (bit-identical), across families it agrees to fp32 rounding."""
    e, w, _ = enc2
    rng = np.random.default_rng(11)
    x = rng.standard_normal((8, 250, 144)).astype(np.float32)
    big = np.tile(x, (34, 1, 1))                       # 272 x 250 = 68 000 tokens
    mid = np.tile(x, (4, 1, 1))                        # 32 x 250 = 8 000 tokens
    few = e.conformer_block(0, x).cpu().numpy()        # 2 000 tokens: one tile per workgroup
    got = e.conformer_block(0, big).cpu().numpy()
    gmid = e.conformer_block(0, mid).cpu().numpy()
    assert np.array_equal(got[:8], got[-8:]) and np.array_equal(gmid[:8], gmid[-8:])
    assert np.array_equal(got[:8], gmid[:8])
    two = e.conformer_block(0, np.tile(x, (2, 1, 1))).cpu().numpy()        # 4 000 tokens: the same family as `few`
    assert np.array_equal(two[:8], few) and np.array_equal(two[8:], few)
    assert maxdiff(got[:8], few) < 2e-5
    lat = e.conformer_block(0, x[:3]).cpu().numpy()    # 750 tokens
    assert np.array_equal(e.conformer_block(0, x[:2]).cpu().numpy(), lat[:2])
    assert np.array_equal(e.conformer_block(0, x[:1]).cpu().numpy(), lat[:1])
    assert np.array_equal(lat, few[:3])
    assert maxdiff(got[:3], lat) < 2e-5
    ref = co.conformer_block(x[:1].astype(np.float64), w, "conformer_block_0", 36)
    assert maxdiff(got[:1], ref) < TOL and maxdiff(lat[:1], ref) < TOL


def test_encoder_parity_two_blocks(enc2):
    e, w, cfg = enc2
    x = waves(2, 32000)
    ref = co.conformer_encoder(x.astype(np.float64), w, cfg)
    got = e(x[..., None]).cpu().numpy()               # reference call shape [B, L, 1]
    assert got.shape == (2, 50, 144)
    assert maxdiff(got, ref) < TOL


def test_ctc_decoder_against_reference_exported_graph(torch_cuda):
    """Trained weights + logits from the reference's own ctc_model.onnx (tests/golden)."""
    from tensorflowasr_amd.models import CTCDecoder, ctc_greedy_decode
    io = golden_ctc_io()
    dec = CTCDecoder(1332, dmodel=144, num_blocks=1, head_size=36, num_heads=4, kernel_size=32)
    dec.load_weights(golden_ctc_weights(), by_name=False)
    la, aa = dec(io["x_a"], return_argmax=True)
    la, aa = la.cpu().numpy(), aa.cpu().numpy()
    assert maxdiff(la, io["logits_a"]) < TOL
    assert (la.argmax(-1) == io["logits_a"].argmax(-1)).all()
    assert np.array_equal(aa, la.argmax(-1))                       # in-kernel argmax == argmax of its own logits
    lb, ab = dec(io["x_b"], return_argmax=True)
    assert np.array_equal(ab.cpu().numpy(), io["argmax_b"])        # 26 % non-blank frames
    assert maxdiff(lb.cpu().numpy()[:, ::8], io["logits_b_every8"]) < TOL
    ids, lens = ctc_greedy_decode(ab, None, 1331)
    rid, rlen = co.ctc_collapse(io["argmax_b"], [250], 1331)
    assert np.array_equal(ids.cpu().numpy(), rid) and np.array_equal(lens.cpu().numpy(), rlen)
    assert rlen[0] > 10


def test_head_argmax_tie_breaks_to_lowest_index(torch_cuda):
    """All-zero fully_connected kernel + constant bias -> every class ties -> id 0 (strict '<', first max)."""
    from tensorflowasr_amd.models import CTCDecoder
    w = golden_ctc_weights()
    w["fully_connected/kernel"] = np.zeros_like(w["fully_connected/kernel"])
    w["fully_connected/bias"] = np.full_like(w["fully_connected/bias"], 0.25)
    dec = CTCDecoder(1332, dmodel=144, num_blocks=1, head_size=36, num_heads=4, kernel_size=32)
    dec.load_weights(w, by_name=False)
    lg, am = dec(np.random.default_rng(0).standard_normal((1, 20, 144)).astype(np.float32), return_argmax=True)
    assert (am.cpu().numpy() == 0).all() and (lg.cpu().numpy() == 0.25).all()
    w["fully_connected/bias"][1331] = 0.5                           # last valid class wins; padding columns never do
    w["fully_connected/bias"][7] = 0.5
    dec.load_weights(w, by_name=False)
    _, am = dec(np.zeros((1, 5, 144), np.float32), return_argmax=True)
    assert (am.cpu().numpy() == 7).all()


def test_frame_argmax_on_a_sub_ulp_tie_keeps_the_larger_logit(torch_cuda):
    """the documented deviation (tests/test_oracle.py): on two logits one ulp apart the device arg-max
    returns the class of the larger logit, where the reference's fp32 log(softmax + 1e-7) ties and returns the lower index"""
    from tensorflowasr_amd.models import frame_argmax
    from test_oracle import _tf_greedy_argmax_fp32, sub_ulp_tie_row
    row = sub_ulp_tie_row(1332)
    got = frame_argmax(torch_cuda.from_numpy(np.tile(row, (3, 4, 1))).cuda()).cpu().numpy()
    assert (got == 5).all() and _tf_greedy_argmax_fp32(row) == 2


def test_greedy_kats_bit_exact(torch_cuda):
    from tensorflowasr_amd.models import ctc_greedy_decode
    kats = json.load(open(os.path.join(GOLDEN, "greedy_kat.json")))
    for k in kats:
        fa = np.array(k["probs"], np.float32).argmax(-1) if "probs" in k else np.array(k["frame_argmax"])
        ids, n = ctc_greedy_decode(fa[None].astype(np.int32), None, k["blank"], device="cuda:0")
        ids, n = ids.cpu().numpy(), int(n.cpu().numpy()[0])
        assert ids[0, :n].tolist() == k["expect"]
        assert (ids[0, n:] == -1).all()


def test_greedy_ragged_lengths_and_edges(torch_cuda):
    from tensorflowasr_amd.models import ctc_greedy_decode
    rng = np.random.default_rng(3)
    B, T, blank = 9, 333, 6
    fa = rng.integers(0, 7, (B, T)).astype(np.int32)
    fa[4] = blank                                    # all blank
    fa[5] = 2                                        # one long run
    in_len = np.array([333, 0, 1, 64, 333, 333, 65, 128, 200], np.int32)
    ids, lens = ctc_greedy_decode(fa, in_len, blank, device="cuda:0")
    rid, rlen = co.ctc_collapse(fa, in_len, blank)
    assert np.array_equal(ids.cpu().numpy(), rid) and np.array_equal(lens.cpu().numpy(), rlen)
    assert rlen[1] == 0 and rlen[4] == 0 and rlen[5] == 1


def test_recognize_full_S_model_ids_identical(full_s):
    m, w, cfg = full_s
    x = waves(2, 48000)
    ids, lens = m.recognize(x)
    ids, lens = ids.cpu().numpy().copy(), lens.cpu().numpy().copy()
    enc_gpu = m.encode(x).cpu().numpy()
    lg_gpu, am_gpu = m.ctc_logits(enc_gpu, return_argmax=True)
    lg_gpu, am_gpu = lg_gpu.cpu().numpy(), am_gpu.cpu().numpy()
    enc_ref = co.conformer_encoder(x.astype(np.float64), w, cfg)
    lg_ref = co.ctc_decoder(enc_ref, w, cfg)
    assert maxdiff(enc_gpu, enc_ref) < TOL
    err = maxdiff(lg_gpu, lg_ref)
    assert err < TOL
    # integer path is bit-exact given the GPU's own logits
    assert_own_argmax(am_gpu, lg_gpu)
    gid, glen = co.ctc_collapse(am_gpu, [lg_gpu.shape[1]] * 2, 1331)
    assert np.array_equal(ids, gid) and np.array_equal(lens, glen)
    # and the oracle's ids, unconditionally (helpers.assert_frames_and_ids: never skips)
    assert_frames_and_ids(lg_gpu, am_gpu, ids, lens, lg_ref, [lg_ref.shape[1]] * 2, 1331)
    assert lens.min() >= 1                              # synthetic head: non-blank tokens present


def test_recognize_respects_input_length(full_s):
    m, _, _ = full_s
    x = waves(3, 32000, 20)
    ids_full, lens_full = m.recognize(x)
    ids_full = ids_full.cpu().numpy().copy()
    in_len = np.array([50, 10, 0], np.int32)            # am_dataloader.py:153-155: in_len = L // 640
    ids, lens = m.recognize(x, in_len)
    ids, lens = ids.cpu().numpy(), lens.cpu().numpy()
    am = m.ctc_logits(m.encode(x), return_argmax=True)[1].cpu().numpy()
    rid, rlen = co.ctc_collapse(am, in_len, 1331)
    assert np.array_equal(ids, rid) and np.array_equal(lens, rlen)
    assert np.array_equal(ids[0], ids_full[0]) and lens[2] == 0


def test_streaming_block_conformer_parity(torch_cuda):
    from tensorflowasr_amd.models import StreamingConformerEncoder
    cfg = small_cfg(2, co.STREAMING_S)
    w = co.encoder_weights(cfg, seed=2)
    e = StreamingConformerEncoder(**encoder_kwargs(cfg))
    e.add_chunk_size(8000, 80, 640)
    e.load_weights(w, by_name=False)
    x = waves(2, 24000, 9)
    ref = co.streaming_conformer_encoder(x.astype(np.float64), w, cfg, 8000)
    got = e(x).cpu().numpy()
    assert got.shape == (2, 39, 256)
    assert maxdiff(got, ref) < TOL
    e.set_inference_func()
    one = e.inference(x[:1, :8000, None]).cpu().numpy()   # test_asr.py:121-128: one block at a time
    assert maxdiff(one, ref[:1, :13]) < TOL
    with pytest.raises(Exception):
        e(x[:, :12000])


def test_full_size_batch64_properties(full_s, torch_cuda):
    """BASELINE configs[1] shape (64 x 10 s): determinism, batch invariance, permutation equivariance, and the
    oracle on one full-length utterance."""
    torch = torch_cuda
    m, w, cfg = full_s
    x = waves(8, 160000, 100)
    xb = torch.from_numpy(np.tile(x, (8, 1))).cuda()
    ids1, lens1 = m.recognize(xb)
    ids1, lens1 = ids1.cpu().numpy().copy(), lens1.cpu().numpy().copy()
    ids2, lens2 = m.recognize(xb)
    assert np.array_equal(ids1, ids2.cpu().numpy()) and np.array_equal(lens1, lens2.cpu().numpy())   # deterministic
    assert ids1.shape == (64, 250)
    for r in range(1, 8):                                                     # replicated rows agree bit for bit
        assert np.array_equal(ids1[:8], ids1[8 * r:8 * r + 8])
    enc_b = m.encode(xb).cpu().numpy()
    enc_1 = m.encode(x[3:4]).cpu().numpy()
    # utterances are independent; a single utterance runs the layer-at-a-time kernels (< 4 096 rows), the batch of 64
    # the fused ones: same function, different summation order
    assert maxdiff(enc_b[3], enc_1[0]) < 1e-4
    assert np.array_equal(m.encode(x[2:5]).cpu().numpy()[1], enc_1[0])        # bit-identical inside a kernel family
    perm = np.random.default_rng(0).permutation(8)
    ids8 = m.recognize(x)[0].cpu().numpy().copy()                             # (recognize() reuses its output buffers)
    ids_p, lens_p = m.recognize(x[perm])
    assert np.array_equal(ids_p.cpu().numpy(), ids8[perm])
    assert np.array_equal(ids8, ids1[:8])                                     # and the decoded ids agree across families
    assert np.isfinite(enc_b).all()
    assert ((ids1 >= -1) & (ids1 < 1331)).all()
    assert all((ids1[b, lens1[b]:] == -1).all() and (ids1[b, :lens1[b]] >= 0).all() for b in range(64))
    enc_ref = co.conformer_encoder(x[:1].astype(np.float64), w, cfg)
    assert maxdiff(enc_b[:1], enc_ref) < TOL


def test_workspace_too_small_is_reported(enc2, torch_cuda):
    torch = torch_cuda
    from tensorflowasr_amd import _lib
    e, _, _ = enc2
    h = e._h
    x = torch.zeros((1, 16000), device="cuda")
    out = torch.empty((1, 25, 144), device="cuda")
    ws = torch.empty(1024, dtype=torch.uint8, device="cuda")
    rc = h.lib.mi355asr_encoder_forward(h.ptr, ctypes.c_void_p(x.data_ptr()), 1, 16000, ctypes.c_void_p(out.data_ptr()),
                                        ctypes.c_void_p(ws.data_ptr()), 1024, None)
    assert rc == -4 and b"workspace too small" in h.lib.mi355asr_last_error()


def test_prefix_beam_device_path_matches_reference_kats(torch_cuda):
    """GPU top-n selection kernel + host search vs the KATs of the reference's own decoder (cases with pruning)."""
    torch = torch_cuda
    from tensorflowasr_amd.models import ctc_prefix_beam_decode
    k = np.load(os.path.join(GOLDEN, "beam_kat.npz"))
    meta = json.loads(str(k["meta"]))
    done = 0
    for i, m in enumerate(meta):
        if not m["cutoff_prob"] < 1.0:
            continue
        p = torch.from_numpy(k["probs_%d" % i][None]).cuda()
        ids, lens, sc, n = ctc_prefix_beam_decode(p, None, m["beam"], m["cutoff_prob"], m["cutoff_top_n"])
        assert n[0] == m["n"]
        assert np.array_equal(lens[0, :m["n"]], k["lens_%d" % i])
        assert np.array_equal(ids[0, :m["n"]], k["ids_%d" % i])
        assert np.array_equal(sc[0, :m["n"]].astype(np.float64), k["scores_%d" % i])
        done += 1
    assert done >= 5


def test_prefix_beam_from_logits_batch(torch_cuda):
    """is_logits path: fused softmax in the selection kernel; compared with the host path on fp32 softmax."""
    torch = torch_cuda
    from tensorflowasr_amd.models import ctc_prefix_beam_decode
    rng = np.random.default_rng(8)
    B, T, V, beam = 6, 120, 9160, 8
    z = (rng.standard_normal((B, T, V)) * 4).astype(np.float32)
    z[..., -1] += 3
    in_len = np.array([120, 77, 1, 120, 5, 64], np.int32)
    zt = torch.from_numpy(z).cuda()
    a = ctc_prefix_beam_decode(zt, in_len, beam, 0.999, 40, is_logits=True)
    p = torch.softmax(zt, -1).cpu().numpy()
    b = ctc_prefix_beam_decode(p, in_len, beam, 0.999, 40)
    assert np.array_equal(a[3], b[3]) and np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])
    assert np.abs(a[2] - b[2]).max() < 1e-3          # v_exp_f32 softmax vs torch softmax
    with pytest.raises(Exception):
        ctc_prefix_beam_decode(zt, in_len, beam, 1.0, 40, is_logits=True)     # un-pruned mode is host-only


def test_prefix_beam_topn_ties_and_zero_rows(torch_cuda):
    """The selection kernel on rows full of ties: the host orders equal probabilities by class (stable_sort), so must the
    kernel -- rows of quantised probabilities, rows with fewer non-zero classes than cutoff_top_n, a row of equal values,
    V below one wave.  Beam 1: the search itself is then free of unspecified tie-breaks."""
    torch = torch_cuda
    from tensorflowasr_amd.models import ctc_prefix_beam_decode
    rng = np.random.default_rng(77)
    for V in (37, 200, 4233):
        B, T = 3, 40
        p = rng.random((B, T, V)).astype(np.float32) ** 8
        p[0] = np.round(p[0] * 16) / 16                      # many ties, many exact zeros
        p[1, :, 5:] *= (rng.random((T, V - 5)) < 0.02)       # fewer than 40 non-zero classes
        p[2, ::2] = 1.0                                      # all equal
        p /= np.maximum(p.sum(-1, keepdims=True), 1e-30)
        for top_n in (40, 7):
            d = ctc_prefix_beam_decode(torch.from_numpy(p).cuda(), None, 1, 0.9999, top_n)
            h = ctc_prefix_beam_decode(p, None, 1, 0.9999, top_n)
            assert np.array_equal(d[3], h[3]) and np.array_equal(d[1], h[1])
            assert np.array_equal(d[0], h[0]) and np.array_equal(d[2], h[2])


def test_prefix_beam_topn_when_few_lanes_hold_all_the_large_classes(torch_cuda):
    """The selection kernels' threshold T0 is the cutoff_top_n-th largest LANE maximum (classes are dealt to 64 lanes
    round-robin).  Rows whose large values all sit in classes = l mod 64 for a few l put hundreds of classes above T0:
    more than the kernels' candidate list holds, so the rounds path (arg-max, knock-out) has to produce the same list --
    in the register-resident kernel (V <= 9 216) and in the LDS kernel (V above)."""
    torch = torch_cuda
    from tensorflowasr_amd.models import ctc_prefix_beam_decode
    rng = np.random.default_rng(78)
    for V in (1332, 9160, 9300):
        B, T = 2, 12
        p = (rng.random((B, T, V)).astype(np.float32) * 1e-4 + 1e-6)
        cls = np.arange(V)
        hot = (cls % 64) < 20                                  # 20 lanes own every large class: > 256 of them
        p[:, :, hot] = rng.random((B, T, int(hot.sum()))).astype(np.float32) + 0.5
        p[1] = np.where(hot[None, :], np.round(p[1] * 8) / 8, p[1])   # and ties among them
        p /= p.sum(-1, keepdims=True)
        for top_n in (40, 25):
            d = ctc_prefix_beam_decode(torch.from_numpy(p).cuda(), None, 1, 0.9999, top_n)
            h = ctc_prefix_beam_decode(p, None, 1, 0.9999, top_n)
            assert np.array_equal(d[3], h[3]) and np.array_equal(d[1], h[1])
            assert np.array_equal(d[0], h[0]) and np.array_equal(d[2], h[2])


def _host_libm(tmp_path):
    """the HOST C library's expf / logf / log, batched (NumPy's own float32 exp / log are SIMD routines with other roundings)"""
    import subprocess
    src = tmp_path / "libm_batch.c"
    src.write_text(r"""
#include <math.h>
#include <float.h>
void batch(int kind, const float* in, void* out, int n) {
  for (int i = 0; i < n; ++i) {
    volatile float x = in[i];
    if (kind == 0) ((float*)out)[i] = expf(x);
    else if (kind == 1) ((float*)out)[i] = logf(x);
    else if (kind == 2) ((double*)out)[i] = log((double)x + (double)FLT_MIN);
    else { volatile float y = in[n + i]; float m = x > y ? x : y;            /* decoder_utils.h:41-49, T = float */
           ((float*)out)[i] = (x <= -FLT_MAX) ? y : (y <= -FLT_MAX) ? x : logf(expf(x - m) + expf(y - m)) + m; }
  }
}
""")
    so = str(tmp_path / "libm_batch.so")
    subprocess.check_call(["gcc", "-O1", "-fno-builtin", "-shared", "-fPIC", str(src), "-o", so, "-lm"])
    h = ctypes.CDLL(so)
    h.batch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    return h.batch


def test_device_score_arithmetic_equals_the_host_c_library(torch_cuda, tmp_path):
    """The float scores of the reference's prefix search are whatever ITS C library returns for expf / logf / log
    (decoder_utils.h:41-49, ctc_beam_search_decoder.cpp:57-59).  The device search evaluates csrc/refmath.h; here 6 x 10^6
    arguments per function go through the device's routines (mi355asr_beam_math_eval) and through the libm of THIS box:
    bit for bit, including the arguments where glibc's logf is not the correctly rounded value."""
    torch = torch_cuda
    from tensorflowasr_amd import _lib
    lib = _lib.lib()
    batch = _host_libm(tmp_path)
    rng = np.random.default_rng(4)
    n = 6_000_000

    def both(kind, x, dt):
        xd = torch.from_numpy(x).cuda()
        nn = x.size // 2 if kind == 3 else x.size
        out = torch.empty(nn, dtype=torch.float64 if dt == np.float64 else torch.float32, device="cuda")
        _lib.check(lib.mi355asr_beam_math_eval(kind, ctypes.c_void_p(xd.data_ptr()), ctypes.c_void_p(out.data_ptr()), nn, None))
        torch.cuda.synchronize()
        ref = np.empty(nn, dt)
        batch(kind, x.ctypes.data, ref.ctypes.data, nn)
        return out.cpu().numpy(), ref

    # expf on [-17.5, 0]: uniform, and log-uniform towards 0
    x = np.concatenate([rng.uniform(-17.5, 0, n // 2), -np.exp(rng.uniform(np.log(1e-8), np.log(17.5), n // 2))]).astype(np.float32)
    d, h = both(0, x, np.float32)
    assert np.array_equal(d.view(np.uint32), h.view(np.uint32))
    assert (h != np.exp(x.astype(np.float64)).astype(np.float32)).sum() > 100     # ... where libm is NOT the rounded exact value
    # logf on [1, 2]: EVERY float of the binade (2^23 + 1 arguments)
    x = np.arange(0x3f800000, 0x40000001, dtype=np.uint32).view(np.float32)
    d, h = both(1, x, np.float32)
    assert np.array_equal(d.view(np.uint32), h.view(np.uint32))
    assert (h != np.log(x.astype(np.float64)).astype(np.float32)).sum() > 50_000
    # log(p + FLT_MIN) in double for float probabilities: the whole dynamic range and the neighbourhood of 1 (its own branch)
    x = np.concatenate([np.exp(rng.uniform(np.log(1.2e-38), 0, n // 2)), rng.uniform(0, 1, n // 4), 1 - np.exp(rng.uniform(-16, -2, n // 4)),
                        [0.0, 1.0, 0.9375, 0.93749994, 1.1754944e-38]]).astype(np.float32)
    d, h = both(2, x, np.float64)
    assert np.array_equal(d.view(np.uint64), h.view(np.uint64))
    # log_sum_exp of two scores as the search forms them
    a = rng.uniform(-5000, 0, n).astype(np.float32)
    b = (a + rng.choice([-1, 1], n) * np.exp(rng.uniform(np.log(1e-4), np.log(30), n))).astype(np.float32)
    a[:1000] = -np.finfo(np.float32).max
    b[500:1500] = -np.finfo(np.float32).max
    d, h = both(3, np.concatenate([a, b]), np.float32)
    assert np.array_equal(d.view(np.uint32), h.view(np.uint32))


def test_prefix_beam_device_search_long_inputs_vs_reference_decoder_and_host(torch_cuda):
    """400-600 frames (tests/golden/beam_long_kat.npz, the reference's own decoder): the device search reproduces the
    reference's ranked scores bit for bit and equals the host search in EVERYTHING -- ids, lengths, scores -- although half
    of the scores are shared by several hypotheses (both searches define the order the reference leaves to nth_element the
    same way: beam.hip Search::better)."""
    torch = torch_cuda
    from test_host import _long_kat_check
    from tensorflowasr_amd.models import ctc_prefix_beam_decode
    dev = _long_kat_check(lambda p, beam, cp, tn: ctc_prefix_beam_decode(torch.from_numpy(p).cuda(), None, beam, cp, tn))
    host = _long_kat_check(lambda p, beam, cp, tn: ctc_prefix_beam_decode(p, None, beam, cp, tn, num_threads=1))
    for d, h in zip(dev, host):
        for a, b in zip(d, h):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("beam", [1, 4, 10, 14, 15, 40, 100])
def test_prefix_beam_device_search_equals_host_search(torch_cuda, beam):
    """The device search (beam_device.hip: one-key-per-thread path up to beam 14, radix path above, and the radix
    fallback of the former on near-ties) against the host search (beam.hip, pinned to the reference decoder's KATs) on the
    same probabilities: peaked rows (the cumulative cut keeps a handful of classes), flat rows (all 40 candidates,
    closely spaced scores), blank-dominated rows; ragged lengths including 0 and 1."""
    torch = torch_cuda
    from tensorflowasr_amd.models import ctc_prefix_beam_decode
    rng = np.random.default_rng(100 + beam)
    B, T, V = 6, 90, 300
    z = rng.standard_normal((B, T, V)).astype(np.float32)
    z[0] *= 6.0
    z[1] *= 0.2
    z[2] *= 3.0
    z[2, :, -1] += 8.0
    z[3] *= rng.uniform(0.1, 6.0, (T, 1)).astype(np.float32)
    z[4] *= 2.0
    z[5] *= 4.0
    z[5, ::3, -1] += 6.0
    z[5, 1::3] = z[5, ::3][: z[5, 1::3].shape[0]] + 0.01 * z[5, 1::3]          # repeated characters
    p = torch.softmax(torch.from_numpy(z), -1).numpy()
    in_len = np.array([T, T, 61, T, 1, 0], np.int32)
    for cutoff_prob, top_n in ((0.99, 40), (0.9999, 25)):
        d = ctc_prefix_beam_decode(torch.from_numpy(p).cuda(), in_len, beam, cutoff_prob, top_n)
        h = ctc_prefix_beam_decode(p, in_len, beam, cutoff_prob, top_n)
        assert np.array_equal(d[3], h[3])
        assert np.array_equal(d[2], h[2])                  # the ranked scores, bit for bit
        # Hypotheses with the same float score and last character: the reference leaves their order to std::nth_element;
        # both searches define it the same way (beam.hip Search::better), so everything is equal, ties included
        assert np.array_equal(d[1], h[1]) and np.array_equal(d[0], h[0])


def test_plain_spectrogram_frontend_encoder_parity(torch_cuda):
    """mel_layer_type = anything but 'Melspectrogram' / 'leaf' builds the plain Spectrogram layer in the reference
    (conformer_blocks.py:318-323): 513 dB bins straight into the subsampling convs (F2 = 129, Dense 129 d -> d)."""
    from tensorflowasr_amd.models import ConformerEncoder
    cfg = dict(small_cfg(1), mel_layer_type="Spectrogram")
    w = co.encoder_weights(cfg, seed=9)
    assert "mel_layer/freq2mel" not in w and w["conv_subsampling/linear/kernel"].shape[0] == 129 * cfg["dmodel"]
    kw = dict(encoder_kwargs(cfg), mel_layer_type="Spectrogram")
    e = ConformerEncoder(**kw)
    assert set(e._expected_shapes()) == set(e._h.weight_names())
    e.load_weights(w, by_name=False)
    x = waves(2, 16000, 31)
    ref, inter = co.conformer_encoder(x.astype(np.float64), w, cfg, return_intermediates=True)
    assert maxdiff(e.melspectrogram(x).cpu().numpy(), inter["mel"]) < 1e-3 * 80      # dB values in [-80, 0]
    assert maxdiff(e(x).cpu().numpy(), ref) < TOL


# ---- row a15: ChunkConformer offline predict (oracle pinned by the reference's own code: tests/test_tf_goldens.py) ---------
def _chunk_model(cfg, w):
    from tensorflowasr_amd.models import ChunkConformer
    m = ChunkConformer(chunk_config_dict(cfg), cfg["picker_num_classes"], cfg["decoder_num_classes"])
    m.load_weights(w, by_name=False)
    return m


@pytest.mark.parametrize("L", [24000, 50000])
def test_chunk_conformer_predict_stage_parity(torch_cuda, L):
    cfg = dict(co.CHUNK_S, enc_num_blocks=2, decoder_num_classes=300)
    w = co.chunk_weights(cfg, seed=3)
    x = waves(3, L, 40)
    w["picker/fully_connected/bias"][-1] = _pick_bias_for_ragged_counts(cfg, w, x)
    ref = co.chunk_predict(x.astype(np.float64), w, cfg)
    m = _chunk_model(cfg, w)
    got = m.predict(x, stages=True)
    for k in ("front", "enc", "picker_logits", "picker_hidden"):
        assert maxdiff(got[k].cpu().numpy(), ref[k]) < TOL, k
    bad = argmax_mismatch_report(got["picker_logits"].cpu().numpy(), ref["picker_logits"])
    assert not bad, bad                                   # feature_pick is driven by this argmax
    assert np.array_equal(got["counts"], ref["counts"])
    assert 0 < ref["counts"].min() < ref["counts"].max() <= ref["front"].shape[1]     # ragged, non-trivial
    for k in ("picked", "helper", "text_logits"):
        assert got[k].shape == ref[k].shape, k
        assert maxdiff(got[k].cpu().numpy(), ref[k]) < TOL, k
    assert np.array_equal(got["text_argmax"].cpu().numpy(), got["text_logits"].cpu().numpy().argmax(-1))
    logits, counts = m.predict(x)
    assert np.array_equal(logits.cpu().numpy(), got["text_logits"].cpu().numpy())      # deterministic


@pytest.mark.parametrize("amp", [1.0, 1e-3, 40.0])
def test_chunk_front_two_term_conv_scales_by_the_run_time_maximum(torch_cuda, amp):
    """The valid ChunkConformer frontend has no dB normalisation: its log10 features have no static bound, so the two-term
    subsampling conv (round 4) takes its operand scale from each utterance's largest |mel|, left by the banded mel kernel as a float
    bit pattern (atomicMax).  Waveforms of three amplitudes -- log10 power shifts by -6 / +3.2 -- one silent utterance in
    the batch (its features sit at log10(amin), far from the maximum the scale follows): the front output against the oracle."""
    cfg = dict(co.CHUNK_S, enc_num_blocks=1, decoder_num_classes=300)
    w = co.chunk_weights(cfg, seed=6)
    x = (waves(4, 48000, 70) * np.float32(amp)).astype(np.float32)
    x[2] = 0.0
    ref = co.chunk_predict(x.astype(np.float64), w, cfg)
    got = _chunk_model(cfg, w).predict(x, stages=True)
    e = maxdiff(got["front"].cpu().numpy(), ref["front"])
    print("chunk front, amplitude %g: max|d| %.3g of max|ref| %.3g" % (amp, e, np.abs(ref["front"]).max()))
    assert e < TOL * max(1.0, float(np.abs(ref["front"]).max()) / 50.0)


def test_chunk_front_is_the_same_for_an_utterance_whatever_shares_its_batch(torch_cuda):
    """Round 5 (advisor): the operand scale of the two-term subsampling conv in the chunk front is taken per UTTERANCE (one
    atomicMax word per utterance), so an utterance's features do not depend on its neighbours: a quiet utterance (amplitude
    1e-3) batched with a loud one (amplitude 30) gives bit for bit what it gives alone, and both stay within the usual
    distance of the oracle."""
    cfg = dict(co.CHUNK_S, enc_num_blocks=1, decoder_num_classes=300)
    w = co.chunk_weights(cfg, seed=6)
    m = _chunk_model(cfg, w)
    x = waves(3, 48000, 75)
    x[0] *= np.float32(30.0)
    x[1] *= np.float32(1e-3)
    both = m.predict(x, stages=True)["front"].cpu().numpy()
    ref = co.chunk_predict(x.astype(np.float64), w, cfg)["front"]
    for b in range(3):
        alone = m.predict(x[b:b + 1], stages=True)["front"].cpu().numpy()
        assert np.array_equal(alone[0], both[b]), (b, float(np.abs(alone[0] - both[b]).max()))
        assert maxdiff(both[b], ref[b]) < TOL * max(1.0, float(np.abs(ref[b]).max()) / 50.0), b


def test_chunk_band_attention_matches_keras_mask_semantics(torch_cuda):
    """win_back > 0 (text decoder, 36/8) at a length that is not a multiple of 16 and shorter than the window."""
    cfg = dict(co.CHUNK_S, enc_num_blocks=1, decoder_num_classes=64, enc_win_front=5, enc_win_back=3,
               picker_win_front=2, picker_win_back=0, helper_win_front=36, helper_win_back=0)
    w = co.chunk_weights(cfg, seed=9, picker_blank_bias=-50.0)      # keep every frame
    x = waves(2, 8000, 3)
    ref = co.chunk_predict(x.astype(np.float64), w, cfg)
    got = _chunk_model(cfg, w).predict(x, stages=True)
    assert ref["front"].shape[1] == 12
    for k in ("front", "enc", "picker_logits", "picked", "helper", "text_logits"):
        assert maxdiff(got[k].cpu().numpy(), ref[k]) < TOL, k


def test_chunk_conformer_full_S_config_10s(torch_cuda):
    """chunk_conformerS.yml dimensions (15 + 1 + 2 + 1 blocks, 277 / 9171 classes) on one 10 s utterance, and the
    prefix beam search on the text logits (BASELINE config 5 path end to end)."""
    from tensorflowasr_amd.models import ctc_prefix_beam_decode
    cfg = dict(co.CHUNK_S)
    w = co.chunk_weights(cfg, seed=5)
    x = waves(1, 160000, 77)
    w["picker/fully_connected/bias"][-1] = _pick_bias_for_ragged_counts(cfg, w, x)
    ref = co.chunk_predict(x.astype(np.float64), w, cfg)
    m = _chunk_model(cfg, w)
    got = m.predict(x, stages=True)
    assert ref["front"].shape[1] == 250
    assert maxdiff(got["enc"].cpu().numpy(), ref["enc"]) < TOL
    assert np.array_equal(got["counts"], ref["counts"])
    err = maxdiff(got["text_logits"].cpu().numpy(), ref["text_logits"])
    assert err < TOL
    # beam search from the GPU logits (fused softmax + top-n) vs the host path on an fp32 softmax of the same logits
    import torch
    a = ctc_prefix_beam_decode(got["text_logits"], got["counts"], 8, 0.999, 40, is_logits=True)
    b = ctc_prefix_beam_decode(torch.softmax(got["text_logits"], -1).cpu().numpy(), got["counts"], 8, 0.999, 40)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


# ---------------------------------------------------------------------------------------------------------
# Translator
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,U,T,blocks", [(2, 7, 30, 1), (3, 40, 250, 2), (1, 1, 13, 1), (2, 23, 300, 2)])
def test_translator_parity(torch_cuda, B, U, T, blocks):
    """Embedding + RBlocks (cross-attention over the encoder output, PE on the query) + Dense head against the
    oracle.  (3, 40, 250): the LDS attention kernel with Tq != Tk; (2, 23, 300): the online-softmax kernel;
    (1, 1, 13): a single token."""
    from tensorflowasr_amd.models import Translator
    cfg = dict(co.CONFORMER_S, translator_num_blocks=blocks, translator_kernel_size=32, translator_fc_factor=0.5)
    inp, tar = 1332, 517
    w = co.translator_weights(cfg, inp, tar, seed=11 + blocks)
    tr = Translator(inp_classes=inp, tar_classes=tar, dmodel=144, num_blocks=blocks, head_size=36, num_heads=4,
                    kernel_size=32, fc_factor=0.5)
    tr.load_weights(w, by_name=False)
    rng = np.random.default_rng(B * 1000 + U)
    ids = rng.integers(0, inp, (B, U)).astype(np.int32)
    ids[0, U // 2:] = 0                                   # 0-padding after the decoded tokens (test_asr.py:199)
    enc = rng.standard_normal((B, T, 144)).astype(np.float32)
    ref = co.translator(ids, enc.astype(np.float64), w, cfg)
    got, amax = tr([ids, enc], training=False, return_argmax=True)
    got = got.cpu().numpy()
    assert got.shape == ref.shape
    assert maxdiff(got, ref) < TOL
    bad = [b for b in argmax_mismatch_report(got, ref) if b[1] > 1e-3]   # ties closer than the tolerance may flip
    assert not bad, bad
    assert (amax.cpu().numpy() == got.argmax(-1)).all()
    tr.set_inference_func()
    assert maxdiff(tr.inference(ids, enc).cpu().numpy(), got) == 0.0


def test_translator_head_on_the_two_term_stream_from_2048_rows(torch_cuda):
    """Round 5: with 2048 token rows or more (a batch of offline_stt calls: 64 utterances x ~90 phone tokens) the Translator's
    Dense(144 -> tar_classes) runs where the CTC decoder's and the ChunkConformer's class heads run -- pp_head_kernel's two-term
    stream.  32 x 80 = 2560 rows, 700 classes (five column groups, the last one partly filled): logits, the in-kernel argmax
    and the fp32-MFMA head of the same rows in two halves (1280 rows each: below the threshold) against each other and the oracle."""
    from tensorflowasr_amd.models import Translator
    cfg = dict(co.CONFORMER_S, translator_num_blocks=1, translator_kernel_size=32, translator_fc_factor=0.5)
    inp, tar, B, U, T = 300, 700, 32, 80, 120
    w = co.translator_weights(cfg, inp, tar, seed=31)
    tr = Translator(inp_classes=inp, tar_classes=tar, dmodel=144, num_blocks=1, head_size=36, num_heads=4, kernel_size=32, fc_factor=0.5)
    tr.load_weights(w, by_name=False)
    rng = np.random.default_rng(77)
    ids = rng.integers(0, inp, (B, U)).astype(np.int32)
    enc = rng.standard_normal((B, T, 144)).astype(np.float32)
    got, amax = tr([ids, enc], return_argmax=True)
    got, amax = got.cpu().numpy(), amax.cpu().numpy()
    assert np.array_equal(amax, got.argmax(-1))
    halves = np.concatenate([tr([ids[:16], enc[:16]]).cpu().numpy(), tr([ids[16:], enc[16:]]).cpu().numpy()])
    ref = co.translator(ids[:4], enc[:4].astype(np.float64), w, cfg)
    e2, e1 = maxdiff(got[:4], ref), maxdiff(halves[:4], ref)
    print("translator head, 2560 rows: two-term stream %.3g, fp32 MFMA kernel %.3g from the oracle; apart %.3g" % (e2, e1, maxdiff(got, halves)))
    assert e2 < TOL and e1 < TOL and e2 < 2 * e1 + 2e-6 and maxdiff(got, halves) < 1e-4
    # what offline_stt asks for: the text ids only, no logits written -- on both head kernels
    none, only = tr([ids, enc], return_argmax=True, return_logits=False)
    assert none is None and np.array_equal(only.cpu().numpy(), amax)
    none, only = tr([ids[:16], enc[:16]], return_argmax=True, return_logits=False)
    assert none is None and np.array_equal(only.cpu().numpy(), halves[:16].argmax(-1))


def test_translator_head_split_over_class_ranges(torch_cuda):
    """Round 6: few rows and many classes -- 64 x 80 = 5 120 rows (80 row workgroups) and 2 500 classes (18 column groups) -- leave
    most of the chip idle, so the column groups of pp_head_kernel are split over several workgroups per row tile (here three
    ranges) and head_combine_kernel picks each row's winner from the per-range (maximum, class) pairs: the ids must equal the
    arg-max of the logits the same kernel writes (lowest class among equal maxima), with and without the logits, and the logits
    must sit at the usual distance from the oracle."""
    from tensorflowasr_amd.models import Translator
    cfg = dict(co.CONFORMER_S, translator_num_blocks=1, translator_kernel_size=32, translator_fc_factor=0.5)
    inp, tar, B, U, T = 300, 2500, 64, 80, 60
    w = co.translator_weights(cfg, inp, tar, seed=33)
    w["fully_connected/kernel"][:, 1700] = w["fully_connected/kernel"][:, 40]          # two classes in different ranges tie on every row
    w["fully_connected/bias"][1700] = w["fully_connected/bias"][40]
    tr = Translator(inp_classes=inp, tar_classes=tar, dmodel=144, num_blocks=1, head_size=36, num_heads=4, kernel_size=32, fc_factor=0.5)
    tr.load_weights(w, by_name=False)
    rng = np.random.default_rng(78)
    ids = rng.integers(0, inp, (B, U)).astype(np.int32)
    enc = rng.standard_normal((B, T, 144)).astype(np.float32)
    got, amax = tr([ids, enc], return_argmax=True)
    got, amax = got.cpu().numpy(), amax.cpu().numpy()
    assert np.array_equal(amax, got.argmax(-1)) and not (amax == 1700).any()
    none, only = tr([ids, enc], return_argmax=True, return_logits=False)
    assert none is None and np.array_equal(only.cpu().numpy(), amax)
    ref = co.translator(ids[:2], enc[:2].astype(np.float64), w, cfg)
    assert maxdiff(got[:2], ref) < TOL


def test_translator_rejects_bad_shapes_and_clamps_ids(torch_cuda):
    from tensorflowasr_amd.models import Translator
    cfg = dict(co.CONFORMER_S, translator_num_blocks=1, translator_kernel_size=32, translator_fc_factor=0.5)
    w = co.translator_weights(cfg, 20, 30, seed=1)
    tr = Translator(inp_classes=20, tar_classes=30, num_blocks=1)
    tr.load_weights(w, by_name=False)
    enc = np.random.default_rng(0).standard_normal((1, 40, 144)).astype(np.float32)
    with pytest.raises(ValueError):
        tr([np.zeros((2, 3), np.int32), enc])
    a = tr([np.array([[25, -3, 4]], np.int32), enc]).cpu().numpy()     # out-of-range ids clamp to [0, 19]
    b = tr([np.array([[19, 0, 4]], np.int32), enc]).cpu().numpy()
    assert maxdiff(a, b) == 0.0
    assert tr([np.zeros((1, 0), np.int32), enc]).shape == (1, 0, 30)


def _write_wav(path, x, sr=16000):
    import wave
    with wave.open(str(path), "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(sr)
        f.writeframes((np.clip(x, -1, 1) * 32767).astype("<i2").tobytes())


def _asr_config(tmp_path, streaming, model_yaml):
    from tensorflowasr_amd.config import load_yaml
    (tmp_path / "phones.txt").write_text("\n".join(["<S>", "</S>", "[SPACE]", "[UNK]"] + ["p%d" % i for i in range(56)]) + "\n")
    (tmp_path / "chars.txt").write_text("\n".join(["<S>", "</S>", "[SPACE]", "[UNK]"] + [chr(0x4e00 + i) for i in range(96)]) + "\n")
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tensorflowasr_amd", "configs")
    cfg = load_yaml(os.path.join(here, "am_data_streaming.yml" if streaming else "am_data.yml"))
    cfg.update(load_yaml(os.path.join(here, model_yaml)))
    cfg["inp_config"]["vocabulary"] = str(tmp_path / "phones.txt")
    cfg["tar_config"]["vocabulary"] = str(tmp_path / "chars.txt")
    cfg["running_config"]["outdir"] = str(tmp_path / "logs")
    return cfg


def test_asr_offline_stt_end_to_end(torch_cuda, tmp_path):
    """test_asr.py:186-219 with checkpoints on disk: wav file -> phones, text; every stage against the oracle."""
    from tensorflowasr_amd.asr import ASR
    cfg = _asr_config(tmp_path, False, "conformerS.yml")
    cfg["model_config"]["num_blocks"] = 2                      # a short encoder keeps the oracle fast
    mc = dict(co.CONFORMER_S, num_blocks=2, translator_num_blocks=2, translator_kernel_size=32, translator_fc_factor=0.5)
    V_in, V_out = 61, 101                                      # vocabulary + blank
    we = co.encoder_weights(mc, seed=21)
    wc = co.ctc_decoder_weights(mc, V_in, seed=22)
    wt = co.translator_weights(mc, V_in, V_out, seed=23)
    x = co.synth_wave(3, length=40000)
    # a random network's frames are nearly identical (one phone for the whole utterance); centring the CTC head on
    # this utterance makes the argmax follow the per-frame deviations -> a 54-token phone sequence
    xq = (np.clip(x, -1, 1) * 32767).astype("<i2").astype(np.float64) / 32768.0
    enc0 = co.conformer_encoder(xq[None], we, mc)
    wc["fully_connected/bias"] = np.zeros(V_in, np.float32)
    wc["fully_connected/bias"] = (-co.ctc_decoder(enc0, wc, mc).mean(axis=(0, 1))).astype(np.float32)
    for sub, w in (("encoder", we), ("ctc_decoder", wc), ("translator", wt)):
        d = tmp_path / "logs" / (sub + "-ckpt")
        d.mkdir(parents=True)
        np.savez(d / "model_5.npz", **{k: v * 0 for k, v in w.items()})        # an older step: must be ignored
        np.savez(d / "model_40.npz", **w)
    _write_wav(tmp_path / "utt.wav", x)
    asr = ASR(cfg)
    assert asr.phone_featurizer.num_classes == V_in and asr.text_featurizer.num_classes == V_out
    phones, text = asr.stt(str(tmp_path / "utt.wav"))
    # oracle pipeline on the same PCM16-quantised samples
    wav = asr.speech_featurizer.load_wav(str(tmp_path / "utt.wav")).astype(np.float64)[None]
    enc = co.conformer_encoder(wav, we, mc)
    ids, lens = co.ctc_greedy(co.ctc_decoder(enc, wc, mc), np.array([enc.shape[1]]), blank=V_in - 1)
    row = np.clip(ids[0][:lens[0]], 0, None)
    tl = co.translator(row[None], enc, wt, mc)
    exp_ph = [int(n) for n in row if n != 0]
    exp_tx = []
    for n in tl[0].argmax(-1):
        if n != 0:
            exp_tx.append(int(n))
        if n == 1:
            break
    assert len(exp_ph) > 3
    assert phones == " ".join(asr.phone_featurizer.iextract(exp_ph))
    assert text == "".join(asr.text_featurizer.iextract(exp_tx))


def test_asr_stream_stt_runs_block_conformer(torch_cuda, tmp_path):
    """test_asr.py:116-164: 0.5 s blocks through the StreamingConformerEncoder, global CTC + Translator per block."""
    from tensorflowasr_amd.asr import ASR
    cfg = _asr_config(tmp_path, True, "Streaming_ConformerS.yml")
    asr = ASR(cfg, load_checkpoint=False)                      # Keras-default random init (seeded)
    x = co.synth_wave(4, length=20000)                         # 2.5 blocks: the tail block is zero-padded
    _write_wav(tmp_path / "s.wav", x)
    phones, text = asr.stt(str(tmp_path / "s.wav"))
    assert isinstance(phones, str) and isinstance(text, str)
    # same ids as running the pieces by hand on the zero-padded signal
    wav = asr.speech_featurizer.load_wav(str(tmp_path / "s.wav"))
    wav = np.pad(wav, (0, 24000 - len(wav)))[None, :, None]
    enc = asr.encoder(wav)
    assert enc.shape[1] == 3 * 13
    ids, _ = asr._phone_ids(enc)
    assert phones == " ".join(asr.phone_featurizer.iextract([int(n) for n in ids[0].cpu().numpy() if n != 0]))


def test_am_tester_metrics_match_oracle_pipeline(torch_cuda, tmp_path):
    """am_tester.py:34-89 over one batch of the list-file loader: S/I/D counts, SER and CER equal the ones the
    oracle pipeline gives with the same error-rate arithmetic."""
    from test_host import _eval_fixture
    from tensorflowasr_amd.eval import AMTester, EvalList, wer
    cfg = _eval_fixture(tmp_path, False)
    cfg["model_config"]["num_blocks"] = 2
    cfg["running_config"]["outdir"] = str(tmp_path / "logs")
    t = AMTester(cfg, load_checkpoint=False)
    ds = EvalList(cfg, t.speech_featurizer, t.phone_featurizer, t.text_featurizer, batch_size=3)
    batch = ds.eval_data_generator()
    x, in_len, ph, _, txt = batch
    mc = dict(co.CONFORMER_S, num_blocks=2, translator_num_blocks=2, translator_kernel_size=32, translator_fc_factor=0.5)
    we, wc, wt = t.encoder.get_weights_dict(), t.ctc_model.get_weights_dict(), t.translator.get_weights_dict()
    for k in ("mel_layer/real_kernels", "mel_layer/imag_kernels"):
        we[k] = we[k].reshape(1024, 513)                       # Keras variable shape [n_dft,1,1,nb] -> oracle's 2-D
    enc = co.conformer_encoder(x[..., 0].astype(np.float64), we, mc)
    wc["fully_connected/bias"] = (-co.ctc_decoder(enc, wc, mc).mean(axis=(0, 1))).astype(np.float32)   # varied argmax
    t.ctc_model.load_weights(wc, by_name=False)
    ids, lens = co.ctc_greedy(co.ctc_decoder(enc, wc, mc), in_len, blank=t.phone_featurizer.blank)
    dense = np.clip(ids[:, :max(int(lens.max()), 1)], 0, None)
    tr = co.translator(dense, enc, wt, mc).argmax(-1)
    n = [0, 0, 0, 0]
    ser = []
    for hyp, ref in zip(dense, ph):
        i, j = [int(v) for v in hyp if v != 0], [int(v) for v in ref if v != 0]
        _, s_, d_, i_ = wer(j, i)
        n[0] += len(j); n[1] += s_; n[2] += i_; n[3] += d_
        ser.append(0 if i == j else 1)
    m = [0, 0, 0, 0]
    for hyp, ref in zip(tr, txt):
        hyp = [int(v) for v in hyp]
        if 1 in hyp:
            hyp = hyp[:hyp.index(1)]
        i, j = [v for v in hyp if v not in (0, 1)], [int(v) for v in ref if v not in (0, 1)]
        _, s_, d_, i_ = wer(j, i)
        m[0] += len(j); m[1] += s_; m[2] += i_; m[3] += d_
    assert lens.min() >= 5                                     # the hypotheses are real sequences, not a constant
    t.set_datasets([batch])
    t.set_all_steps(1)
    r = t.run()
    assert r["phone_s_i_d"] == "%d_%d_%d" % tuple(n[1:]) and r["trans_s_i_d"] == "%d_%d_%d" % tuple(m[1:])
    assert abs(r["phone_cer"] - sum(n[1:]) / (n[0] + 1e-6)) < 1e-12 and abs(r["txt_cer"] - sum(m[1:]) / (m[0] + 1e-6)) < 1e-12
    assert r["phone_ser"] == np.mean(ser) and r["steps"] == 1


# ---------------------------------------------------------------------------------------------------------
# ChunkConformer streaming with explicit caches (chunk_conformer_blocks.py:799-866)
# ---------------------------------------------------------------------------------------------------------
from helpers import pick_bias_for_ragged_counts as _pick_bias_for_ragged_counts, stream_oracle as _stream_oracle  # noqa: E402


@pytest.mark.parametrize("samples,nchunks", [(2560, 30), (5120, 9)])
def test_chunk_conformer_streaming_matches_oracle_and_offline(torch_cuda, samples, nchunks):
    """picker_stream_predict / feature_pick / decoder_stream_predict fed chunk by chunk, as test_chunk_asr.py:60-83
    drives them: every valid output against the oracle's restatement of the stream_call chain, the final caches
    against the oracle's, and -- the point of the cache design -- against the OFFLINE predict of the whole signal."""
    cfg = dict(co.CHUNK_S, enc_num_blocks=2, picker_num_classes=30, decoder_num_classes=40)
    w = co.chunk_weights(cfg, seed=3)
    x = waves(1, samples * nchunks, 5)
    w["picker/fully_connected/bias"][-1] = _pick_bias_for_ragged_counts(cfg, w, x)
    ref_ph, ref_hid, ref_txt, ref_unv, rpc, rdc, rsteps = _stream_oracle(x.astype(np.float64), w, cfg, nchunks, samples)
    m = _chunk_model(cfg, w)
    caches, caches2 = m.init_picker_caches(1), m.init_decoder_caches(1)
    ph, txt, steps, unv = [], [], [], None
    for i in range(nchunks):
        vp, _, vh, caches = m.picker_stream_predict(x[:, i * samples:(i + 1) * samples, None], caches)
        if vp.shape[1] == 0:
            continue
        ph.append(vp)
        f, _ = m.feature_pick(vh, vp)
        if f.shape[1] != 0:
            vt, unv, caches2 = m.decoder_stream_predict(f, caches2)
            txt.append(vt)
            steps.append((i, vt.shape[1]))
    ph = torch_cuda.cat(ph, 1).cpu().numpy()
    txt = torch_cuda.cat(txt, 1).cpu().numpy()
    assert steps == rsteps                                           # same picks at every step
    assert ph.shape == ref_ph.shape and maxdiff(ph, ref_ph) < TOL
    assert txt.shape == ref_txt.shape and maxdiff(txt, ref_txt) < TOL
    assert maxdiff(unv.cpu().numpy(), ref_unv) < TOL
    # caches: window lengths and contents
    assert caches[0].shape == (1, 2560, 1) and caches[1].shape == (1, 4, 80, 1)
    assert caches[2].shape == (2, 1, 36, 144) and caches[3].shape == (2, 1, 32, 144)
    assert maxdiff(caches[2][1, 0].cpu().numpy(), rpc["enc_mha"][1][0]) < TOL
    assert maxdiff(caches[3][0, 0].cpu().numpy(), rpc["enc_cnn"][0][0]) < TOL
    assert maxdiff(caches[1][0, :, :, 0].cpu().numpy(), rpc["front_sub"][0]) < TOL
    assert caches2[4].shape[1] == rdc["dec_inp"].shape[1] == 8        # decoder win_back rows wait for right context
    assert maxdiff(caches2[2][0, 0].cpu().numpy(), rdc["decoder_mha"][0][0]) < TOL
    if samples != 2560:
        return      # feeding more than chunk_num * hop samples per call drops mel frames (:452), by the reference's design
    # streaming == offline on the same signal
    off = m.predict(x, stages=True)
    assert maxdiff(ph, off["picker_logits"].cpu().numpy()) < TOL
    n = txt.shape[1]
    assert maxdiff(txt, off["text_logits"].cpu().numpy()[:, :n]) < TOL
    assert n == off["text_logits"].shape[1] - 8


def test_feature_pick_entry_point_matches_oracle(torch_cuda):
    """mi355asr_feature_pick_count / _gather (argmax, compaction, gather on the device) against the oracle's
    feature_pick: ragged counts, an utterance with nothing kept, max_T padding, a class count that is not a multiple of 4."""
    cfg = dict(co.CHUNK_S, enc_num_blocks=1)
    m = _chunk_model(cfg, co.chunk_weights(cfg, seed=1))
    rng = np.random.default_rng(4)
    B, T, d, V = 4, 75, 144, cfg["picker_num_classes"]
    assert V % 4 != 0
    hid = rng.standard_normal((B, T, d)).astype(np.float32)
    ctc = rng.standard_normal((B, T, V)).astype(np.float32)
    ctc[0, :, -1] += 1.5                                   # mostly blank
    ctc[2, :, -1] += 50.0                                  # all blank: nothing kept
    ctc[3, :, -1] -= 50.0                                  # never blank: everything kept
    ctc[1, 10, 5] = ctc[1, 10, 9] = 40.0                   # a tie: the first maximum wins (class 5, not blank)
    rf, counts = co.feature_pick(hid.astype(np.float64), ctc.astype(np.float64), V - 1)
    rc = np.zeros((B, T, V), np.float32)                   # the same compaction applied to the ctc rows
    for b in range(B):
        keep = ctc[b].argmax(-1) != V - 1
        rc[b, :keep.sum()] = ctc[b][keep]
    assert counts.tolist() == [int((ctc[b].argmax(-1) != V - 1).sum()) for b in range(B)] and counts[2] == 0 and counts[3] == T
    f, c = m.feature_pick(hid, ctc)
    assert f.shape == rf.shape == (B, T, d) and c.shape == rc.shape
    assert np.array_equal(f.cpu().numpy(), rf.astype(np.float32)) and np.array_equal(c.cpu().numpy(), rc)
    f2, c2 = m.feature_pick(hid[:3], ctc[:3], max_T=90)             # max_T above the batch maximum: zero padded to it
    n = int(counts[:3].max())
    assert f2.shape[1] == c2.shape[1] == 90 > n and np.array_equal(f2.cpu().numpy()[:, :n], rf[:3, :n].astype(np.float32))
    assert not f2.cpu().numpy()[2].any() and not f2.cpu().numpy()[:, n:].any()
    f3, _ = m.feature_pick(hid[:3], ctc[:3], max_T=5)                # below it: the batch maximum wins
    assert f3.shape[1] == n
    f0, c0 = m.feature_pick(hid[2:3], ctc[2:3])
    assert f0.shape == (1, 0, d) and c0.shape == (1, 0, V)


def _chunk_asr_config(tmp_path, cfg):
    from tensorflowasr_amd.config import load_yaml
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tensorflowasr_amd", "configs")
    c = load_yaml(os.path.join(here, "am_data.yml"))
    c.update(load_yaml(os.path.join(here, "chunk_conformerS.yml")))
    c["model_config"]["ChunkConformerEncoder"]["num_blocks"] = cfg["enc_num_blocks"]
    (tmp_path / "phones.txt").write_text("\n".join(["<S>", "</S>", "[SPACE]", "[UNK]"] + ["p%d" % i for i in range(26)]) + "\n")
    (tmp_path / "chars.txt").write_text("\n".join(["<S>", "</S>", "[SPACE]", "[UNK]"] + [chr(0x4e00 + i) for i in range(36)]) + "\n")
    c["inp_config"]["vocabulary"] = str(tmp_path / "phones.txt")
    c["tar_config"]["vocabulary"] = str(tmp_path / "chars.txt")
    c["running_config"]["outdir"] = str(tmp_path / "logs")
    return c


def test_chunk_asr_stream_call_and_tester(torch_cuda, tmp_path):
    """test_chunk_asr.py:47-139 and chunk_tester.py:34-63 on the shipped chunk_conformerS.yml (2 encoder blocks): the
    streaming loop's final phones are the greedy decode of the offline picker's kept frames, the offline text is the
    greedy decode of predict(); the tester's S/I/D and error rates equal a hand computation from the same logits."""
    from tensorflowasr_amd.chunk_asr import ChunkAMTester, ChunkASR
    from tensorflowasr_amd.eval import wer
    cfg = dict(co.CHUNK_S, enc_num_blocks=2, picker_num_classes=31, decoder_num_classes=41)
    conf = _chunk_asr_config(tmp_path, cfg)
    asr = ChunkASR(conf, load_checkpoint=False)
    assert asr.phone_featurizer.num_classes == 31 and asr.text_featurizer.num_classes == 41 and asr.wav_buf_length == 2560
    w = co.chunk_weights(cfg, seed=3)
    x = co.synth_wave(5, length=2560 * 24)
    _write_wav(tmp_path / "u.wav", x)
    xq = asr.load_wav(str(tmp_path / "u.wav"))
    w["picker/fully_connected/bias"][-1] = _pick_bias_for_ragged_counts(cfg, w, xq[None].astype(np.float32))
    asr.runner.load_weights(w, by_name=False)
    res = asr.stream_call(str(tmp_path / "u.wav"))
    assert len(res["streaming"]) > 5 and res["streaming"][-1][0] == 24 * 0.16
    times = [t for t, _, _ in res["streaming"]]
    assert times == sorted(times)
    off = asr.runner.predict(xq[None], stages=True)
    keep = off["picker_logits"].cpu().numpy()[0].argmax(-1) != 30
    pl = off["picker_logits"].cpu().numpy()[:, keep]
    rid, rlen = co.ctc_greedy(pl, [pl.shape[1]], 30)
    phones = " ".join(asr.phone_featurizer.iextract([int(n) for n in rid[0, :rlen[0]] if n > 0]))
    assert res["streaming"][-1][1] == phones and len(phones) > 0
    tl = off["text_logits"].cpu().numpy()
    tid, tlen = co.ctc_greedy(tl, [tl.shape[1]], 40)
    assert res["offline"] == "".join(asr.text_featurizer.iextract([int(n) for n in tid[0, :tlen[0]] if n > 0]))
    # tester over one batch of two utterances
    t = ChunkAMTester(conf, load_checkpoint=False)
    t.runner.load_weights(w, by_name=False)
    xb = np.stack([xq[:2560 * 12], xq[2560 * 12:]]).astype(np.float32)[..., None]
    labels = np.array([[4, 9, 12, 0], [7, 7, 30, 5]], np.int32)
    lg, _ = t.runner.predict(xb)
    hid, hlen = co.ctc_greedy(lg.cpu().numpy(), [lg.shape[1]] * 2, 40)
    n = [0, 0, 0, 0]
    ser = []
    for hyp, ref in zip(np.clip(hid[:, :max(int(hlen.max()), 1)], 0, None), labels):
        i, j = [int(v) for v in hyp if v != 0], [int(v) for v in ref if v != 0]
        _, s_, d_, i_ = wer(j, i)
        n[0] += len(j); n[1] += s_; n[2] += i_; n[3] += d_
        ser.append(0 if i == j else 1)
    t.set_datasets([(xb, np.array([48, 48], "int32"), None, None, labels, None)])
    r = t.run()
    assert r["s_i_d"] == "%d_%d_%d" % tuple(n[1:]) and r["steps"] == 1
    assert abs(r["cer"] - sum(n[1:]) / (n[0] + 1e-6)) < 1e-12 and r["ser"] == np.mean(ser)


# ---------------------------------------------------------------------------------------------------------
# bf16 MFMA mode (mi355asr_config.gemm_dtype = 1; BASELINE config 3)
# ---------------------------------------------------------------------------------------------------------
def _bf16_case(base, blocks, L, chunk, seed):
    from tensorflowasr_amd.models import ConformerCTC
    cfg = small_cfg(blocks, base)
    w = co.encoder_weights(cfg, seed=seed)
    w.update(co.ctc_decoder_weights(cfg, 300, seed=seed + 1))
    kw = {k: v for k, v in encoder_kwargs(cfg, chunk).items() if k != "mel_layer_type"}
    x = waves(3, L, 17)
    out = {}
    for dt in ("float32", "bfloat16"):
        m = ConformerCTC(300, gemm_dtype=dt, **kw)
        m.load_weights(w, by_name=False)
        enc = m.encode(x)
        logits, amax = m.ctc_logits(enc, return_argmax=True)
        out[dt] = (enc.cpu().numpy(), logits.cpu().numpy(), amax.cpu().numpy())
    return cfg, w, x, out


@pytest.mark.parametrize("base,blocks,L,chunk", [(co.STREAMING_S, 2, 24000, 8000), (co.CONFORMER_S, 2, 32000, 0)])
def test_bf16_gemm_mode_against_rounding_oracle_and_fp32(torch_cuda, base, blocks, L, chunk):
    """bf16 inputs / fp32 accumulation for the dense layers (d = 256 streaming blocks and d = 144 offline):
    (1) within 2e-3 of the oracle run with both GEMM operands rounded to bf16 -- this pins the implementation;
    (2) against the fp32 path: the deviation the bf16 mode costs ("tolerance relaxed; report max abs diff
    and ids agreement"), bounded loosely."""
    cfg, w, x, out = _bf16_case(base, blocks, L, chunk, seed=31)
    co.GEMM_ROUND_BF16 = True
    try:
        if chunk:
            enc_ref = co.streaming_conformer_encoder(x.astype(np.float64), w, cfg, chunk)
        else:
            enc_ref = co.conformer_encoder(x.astype(np.float64), w, cfg)
        logits_ref = co.ctc_decoder(enc_ref, w, cfg)
    finally:
        co.GEMM_ROUND_BF16 = False
    enc16, logits16, amax16 = out["bfloat16"]
    enc32, logits32, amax32 = out["float32"]
    assert enc16.shape == enc_ref.shape
    # End to end the two are not bit-comparable: the fp32 frontend differs from the fp64 oracle's by ~1e-5 relative,
    # which moves a fraction of a percent of the GEMM operands across a bf16 rounding boundary (one bf16 ulp = 2^-8
    # relative each).  Loose bound here; the per-block test below pins the arithmetic.
    e_enc, e_log = np.abs(enc16 - enc_ref), np.abs(logits16 - logits_ref)
    print("bf16 vs rounding oracle: encoder max %.3g mean %.3g ; logits max %.3g mean %.3g"
          % (e_enc.max(), e_enc.mean(), e_log.max(), e_log.mean()))
    assert e_enc.max() < 2e-2 and e_enc.mean() < 2e-3
    assert e_log.max() < 4e-2 and e_log.mean() < 3e-3
    assert (amax16 == logits16.argmax(-1)).all()
    d_enc, d_log = maxdiff(enc16, enc32), maxdiff(logits16, logits32)
    agree = float((amax16 == amax32).mean())
    print("bf16 vs fp32: encoder max|d|=%.3g logits max|d|=%.3g argmax agreement=%.4f" % (d_enc, d_log, agree))
    assert 1e-4 < d_enc < 0.2 and d_log < 0.3 and agree > 0.95


@pytest.mark.parametrize("base,T", [(co.STREAMING_S, 45), (co.CONFORMER_S, 45), (co.CONFORMER_S, 250)])
def test_bf16_block_matches_rounding_oracle(torch_cuda, base, T):
    """One ConformerBlock in bf16 mode on an exact fp32 input: every dense layer sees the same operands as the oracle
    with both GEMM operands rounded to bf16, so only the (rare) fp32-vs-fp64 tie flips remain.  (CONFORMER_S, 250): dmodel 144 in
    bf16 mode at a length where the two-term attention kernel runs with the static q / k / v bounds -- computed with the 1.01
    margin that covers the bf16 rounding of both factors (round-5 advice)."""
    from tensorflowasr_amd.models import ConformerCTC
    cfg = small_cfg(1, base)
    w = co.encoder_weights(cfg, seed=5)
    w.update(co.ctc_decoder_weights(cfg, 100, seed=6))
    m = ConformerCTC(100, gemm_dtype="bfloat16", **{k: v for k, v in encoder_kwargs(cfg).items() if k != "mel_layer_type"})
    m.load_weights(w, by_name=False)
    x = np.random.default_rng(2).standard_normal((3, T, cfg["dmodel"])).astype(np.float32)
    got = m.conformer_block(0, x).cpu().numpy()
    co.GEMM_ROUND_BF16 = True
    try:
        ref = co.conformer_block(x.astype(np.float64), w, "conformer_block_0", cfg["head_size"], cfg["fc_factor"])
    finally:
        co.GEMM_ROUND_BF16 = False
    exact = co.conformer_block(x.astype(np.float64), w, "conformer_block_0", cfg["head_size"], cfg["fc_factor"])
    e = np.abs(got - ref)
    print("bf16 block vs rounding oracle: max %.3g mean %.3g (vs exact: max %.3g)" % (e.max(), e.mean(), np.abs(got - exact).max()))
    assert e.max() < 6e-3 and e.mean() < 3e-4      # tie flips: ~1e-4 mean (see the test above)
    assert np.abs(got - exact).max() > 10 * e.mean()               # it really is the bf16 path


@pytest.mark.parametrize("base,dt,B,T", [(co.CONFORMER_S, "float32", 9, 250), (co.STREAMING_S, "bfloat16", 8, 260), (co.STREAMING_S, "float32", 2, 40)])
def test_ctc_decoder_without_logits_returns_the_same_frame_argmax(torch_cuda, base, dt, B, T):
    """CTCDecoder(..., return_logits=False): mi355asr_ctc_forward with logits = NULL (the class head keeps its running argmax
    only: what the streaming "global CTC" + greedy decode of BASELINE config 3 consumes) gives the argmax of the logits the
    same call writes otherwise -- on the fused dmodel-144 head, the ring head of dmodel 256 and the layer-at-a-time one."""
    from tensorflowasr_amd.models import CTCDecoder
    cfg = small_cfg(1, base)
    w = co.ctc_decoder_weights(cfg, 300, seed=12)
    m = CTCDecoder(num_classes=300, dmodel=cfg["dmodel"], num_blocks=cfg["ctcdecoder_num_blocks"], head_size=cfg["head_size"],
                   num_heads=cfg["num_heads"], kernel_size=cfg.get("ctcdecoder_kernel_size", 32), fc_factor=cfg["fc_factor"], gemm_dtype=dt)
    m.load_weights(w, by_name=False)
    x = np.random.default_rng(T).standard_normal((B, T, cfg["dmodel"])).astype(np.float32)
    logits, amax = m(x, return_argmax=True)
    none, amax2 = m(x, return_argmax=True, return_logits=False)
    assert none is None
    assert np.array_equal(amax.cpu().numpy(), amax2.cpu().numpy())
    assert np.array_equal(amax.cpu().numpy(), logits.cpu().numpy().argmax(-1))
    with pytest.raises(ValueError):
        m(x, return_logits=False)


def test_band_attention_staged_window_bit_identical(torch_cuda):
    """Round 4: band attention (chunk_conformer_blocks.py:158-176; win_front 36, win_back 0 / 8) stages the K / V window of a
    workgroup's 64 queries in LDS once (16-byte loads) instead of every wave reading its fragments from L2 (4-byte loads down
    V's columns: the memory pipe's instruction rate was half of the kernel).  Same MFMAs on the same operands in the same
    order: the ChunkConformer's predict must be BIT-IDENTICAL to MI355ASR_ATTN_BAND_LDS=0 -- utterances of 250, 47 and 13
    frames (shorter than a window, shorter than a query tile) and 1 000 frames (sixteen workgroups per head)."""
    import subprocess
    import sys
    import tempfile
    code = r'''
import sys, numpy as np
sys.path.insert(0, "tests")
from helpers import chunk_config_dict, co, waves
from tensorflowasr_amd.models import ChunkConformer
c5 = dict(co.CHUNK_S, enc_num_blocks=2, picker_num_blocks=1, helper_num_blocks=1, decoder_num_blocks=1)
w5 = co.chunk_weights(c5, seed=4)
m = ChunkConformer(chunk_config_dict(c5), c5["picker_num_classes"], c5["decoder_num_classes"])
m.load_weights(w5, by_name=False)
out = {}
for B, L in ((5, 160000), (3, 30000), (2, 8320), (2, 640000)):
    got = m.predict(waves(B, L, 9), stages=True)
    out["enc_%d" % L] = got["enc"].cpu().numpy()
    out["text_%d" % L] = got["text_logits"].cpu().numpy()
np.savez(sys.argv[1], **out)
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for tag, extra in (("staged", {}), ("l2", {"MI355ASR_ATTN_BAND_LDS": "0"})):
            f = os.path.join(td, tag + ".npz")
            r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=900, cwd=root)
            assert r.returncode == 0, r.stderr[-3000:]
            res[tag] = dict(np.load(f))
    for k in res["staged"]:
        assert np.isfinite(res["staged"][k]).all(), k
        assert np.array_equal(res["staged"][k], res["l2"][k]), (k, float(np.abs(res["staged"][k] - res["l2"][k]).max()))
    assert np.abs(res["staged"]["enc_160000"]).max() > 0.1


def test_bf16_chain256_against_layer_at_a_time(torch_cuda):
    """Round 4: in bf16 mode a dmodel-256 FFModule / ConvModule tail is ONE launch (chain256_bf16_kernel: hidden activation in
    LDS as bf16 operand fragments) instead of two gemm16 / gemm_ring launches with the fp32 hidden activation in HBM.  Same
    operands (both GEMMs' inputs rounded to nearest-even bf16), same fp32 accumulation order along K; only the trailing
    LayerNorm sums its row in a different order and the first GEMM walks K in one piece -- so a block's output must agree with
    MI355ASR_CHAIN256=0 up to the rare bf16 flips of hidden values whose fp32 sums differ in the last bit, at row counts that
    take one row tile per workgroup (45, 832 rows), two (8 208 rows = 513 tiles: an odd tile count, and 8 195 rows: a
    partial last tile) and five (round 6: 16 640 rows), and stay within the rounding oracle's tolerance."""
    import subprocess
    import sys
    import tempfile
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, "tests")
from helpers import co, encoder_kwargs, small_cfg
from tensorflowasr_amd.models import ConformerCTC
cfg = small_cfg(1, co.STREAMING_S)
w = co.encoder_weights(cfg, seed=5)
w.update(co.ctc_decoder_weights(cfg, 100, seed=6))
m = ConformerCTC(100, gemm_dtype="bfloat16", **{k: v for k, v in encoder_kwargs(cfg).items() if k != "mel_layer_type"})
m.load_weights(w, by_name=False)
out = {}
for B, T in ((1, 45), (64, 13), (27, 304), (5, 1639), (64, 260)):
    x = np.random.default_rng(B * T).standard_normal((B, T, cfg["dmodel"])).astype(np.float32)
    out["blk_%d_%d" % (B, T)] = m.conformer_block(0, x).cpu().numpy()
np.savez(sys.argv[1], **out)
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for tag, extra in (("chain", {}), ("layers", {"MI355ASR_CHAIN256": "0"}), ("rt1", {"MI355ASR_CHAIN256_RT": "1"}), ("rt4", {"MI355ASR_CHAIN256_RT": "4"}),
                           ("rt5", {"MI355ASR_CHAIN256_RT": "5"})):
            f = os.path.join(td, tag + ".npz")
            r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=900, cwd=root)
            assert r.returncode == 0, r.stderr[-3000:]
            res[tag] = dict(np.load(f))
    for k in res["chain"]:
        a, b = res["chain"][k], res["layers"][k]
        assert np.isfinite(a).all() and np.abs(a).max() > 0.1, k
        d = np.abs(a.astype(np.float64) - b)
        print(k, "chain vs layer-at-a-time: max %.3g mean %.3g" % (d.max(), d.mean()))
        # a bf16 operand of the SECOND GEMM still flips when the first GEMM's fp32 sum lands within an ulp of a rounding boundary
        # (gemm16 splits K over four waves, the ring kernels walk it in 32-wide steps: other summation orders): one bf16 ulp of
        # a hidden value each, measured mean 3e-5 -- a tenth of the distance either has from the rounding oracle
        assert d.mean() < 1e-4 and d.max() < 6e-3, k
    # one, two, four or five row tiles per workgroup (four: the hidden dimension in two phases, five: in four -- what 64 x 260 rows
    # take by default, 1 040 tiles being 260 workgroups of four = a second round over 256 CUs): the same summation orders, bit for bit
    for k in res["chain"]:
        assert np.array_equal(res["chain"][k], res["rt1"][k]) and np.array_equal(res["chain"][k], res["rt4"][k]), k
        assert np.array_equal(res["chain"][k], res["rt5"][k]), k
    cfg = small_cfg(1, co.STREAMING_S)
    w = co.encoder_weights(cfg, seed=5)
    x = np.random.default_rng(45).standard_normal((1, 45, cfg["dmodel"])).astype(np.float32)
    co.GEMM_ROUND_BF16 = True
    try:
        ref = co.conformer_block(x.astype(np.float64), w, "conformer_block_0", cfg["head_size"], cfg["fc_factor"])
    finally:
        co.GEMM_ROUND_BF16 = False
    e = np.abs(res["chain"]["blk_1_45"] - ref)
    assert e.max() < 6e-3 and e.mean() < 3e-4


def test_bf16_gemm256_rows_resident_against_the_ring_kernels(torch_cuda):
    """Round 6: in bf16 mode the K = 256 dense layers of a dmodel-256 block (qkv, attention out, pw_conv_1 + GLU) and the CTC projection
    run from 8 192 rows on gemm256_bf16_kernel -- a workgroup's row tiles normalised and converted once, resident in LDS, the column
    tiles split over its waves -- instead of the slab-ring kernel (MI355ASR_GEMM256=0).  Same operands (nearest-even bf16), same fp32
    accumulation along K in 32-wide steps; the prologue LayerNorm reduces its row in another order, so a hidden value on a rounding
    boundary may flip: the CTC decoder over 64 x 260 rows (five row tiles per workgroup), 40 x 250 (four) and 33 x 251 = 8 283 (a partial last tile in a partial last workgroup) must agree
    with the ring build like chain256 does with the layer-at-a-time build, and sit at the rounding oracle's distance."""
    import subprocess
    import sys
    import tempfile
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, "tests")
from helpers import co
from tensorflowasr_amd.models import CTCDecoder
cfg = dict(co.STREAMING_S)
w = co.ctc_decoder_weights(cfg, 300, seed=8)
ctc = CTCDecoder(num_classes=300, dmodel=256, num_blocks=1, head_size=64, num_heads=4, kernel_size=32, fc_factor=0.5, gemm_dtype="bfloat16")
ctc.load_weights(w, by_name=False)
out = {}
for B, T in ((64, 260), (40, 250), (33, 251)):
    h = np.random.default_rng(B).standard_normal((B, T, 256)).astype(np.float32)
    out["lg_%d_%d" % (B, T)] = ctc(torch.from_numpy(h).cuda()).cpu().numpy()
np.savez(sys.argv[1], **out)
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for tag, extra in (("rows", {}), ("ring", {"MI355ASR_GEMM256": "0"})):
            f = os.path.join(td, tag + ".npz")
            r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=900, cwd=root)
            assert r.returncode == 0, r.stderr[-3000:]
            res[tag] = dict(np.load(f))
    for k in res["rows"]:
        a, b = res["rows"][k], res["ring"][k]
        d = np.abs(a.astype(np.float64) - b)
        print(k, "rows-resident vs ring: max %.3g mean %.3g" % (d.max(), d.mean()))
        # (measured: mean 9e-5, max 6e-3 -- a sixth of either build's distance from the rounding oracle, 5.7e-4 / 6.5e-3)
        assert np.isfinite(a).all() and d.max() > 0.0 and d.mean() < 2e-4 and d.max() < 1.2e-2, k
    cfg = dict(co.STREAMING_S)
    w = co.ctc_decoder_weights(cfg, 300, seed=8)
    h = np.random.default_rng(64).standard_normal((64, 260, 256)).astype(np.float32)[:2]
    co.GEMM_ROUND_BF16 = True
    try:
        ref = co.ctc_decoder(h.astype(np.float64), w, cfg)
    finally:
        co.GEMM_ROUND_BF16 = False
    e = np.abs(res["rows"]["lg_64_260"][:2] - ref)
    print("rows-resident vs rounding oracle: max %.3g mean %.3g" % (e.max(), e.mean()))
    assert e.max() < 4e-2 and e.mean() < 3e-3


# ---------------------------------------------------------------------------------------------------------
# ConformerM / ConformerL (asr/configs/conformerM.yml, conformerL.yml)
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("base,L", [(co.CONFORMER_L, 24000), (co.CONFORMER_M, 32000), (co.CONFORMER_L, 3000)])
def test_conformer_m_and_l_parity(torch_cuda, base, L):
    """dmodel 256 (chained fp32 kernels + column-split subsampling conv) and dmodel 512 = 8 heads x 64 (layer-at-a-time
    fp32 GEMM kernels): waveform -> encoder -> CTC logits -> greedy ids against the oracle."""
    from tensorflowasr_amd.models import ConformerCTC
    cfg = small_cfg(2, base)
    w = co.encoder_weights(cfg, seed=41)
    w.update(co.ctc_decoder_weights(cfg, 200, seed=42))
    m = ConformerCTC(200, **{k: v for k, v in encoder_kwargs(cfg).items() if k != "mel_layer_type"})
    m.load_weights(w, by_name=False)
    x = waves(2, L, 23)
    enc_ref = co.conformer_encoder(x.astype(np.float64), w, cfg)
    logits_ref = co.ctc_decoder(enc_ref, w, cfg)
    enc = m.encode(x)
    logits, amax = m.ctc_logits(enc, return_argmax=True)
    assert maxdiff(enc.cpu().numpy(), enc_ref) < TOL
    assert maxdiff(logits.cpu().numpy(), logits_ref) < TOL
    bad = [b for b in argmax_mismatch_report(logits.cpu().numpy(), logits_ref) if b[1] > 1e-3]
    assert not bad, bad
    ids, lens = m.recognize(x)
    rid, rlen = co.ctc_greedy(logits.cpu().numpy(), [logits.shape[1]] * 2, 199)
    assert (ids.cpu().numpy() == rid).all() and (lens.cpu().numpy() == rlen).all()


@pytest.mark.parametrize("T", [260, 272, 250, 273, 288, 289, 40])
def test_head_size_64_block_at_streaming_ctc_lengths(torch_cuda, T):
    """One dmodel-256 block (4 heads x 64) on 260 frames -- the "global CTC" history of BASELINE config 3 (20 chunks x 13
    frames) -- and around the limits of the attention kernels: round 5's two-term attention_split64_kernel up to 288 keys (33 ..
    288: one to three query tiles per wave, the last key tile partly filled), beyond it the L2-streaming fp32 kernel."""
    from tensorflowasr_amd.models import ConformerCTC
    cfg = small_cfg(1, co.CONFORMER_M)
    w = co.encoder_weights(cfg, seed=15)
    w.update(co.ctc_decoder_weights(cfg, 100, seed=16))
    m = ConformerCTC(100, **{k: v for k, v in encoder_kwargs(cfg).items() if k != "mel_layer_type"})
    m.load_weights(w, by_name=False)
    x = np.random.default_rng(T).standard_normal((2, T, cfg["dmodel"])).astype(np.float32)
    got = m.conformer_block(0, x).cpu().numpy()
    ref = co.conformer_block(x.astype(np.float64), w, "conformer_block_0", cfg["head_size"], cfg["fc_factor"])
    assert maxdiff(got, ref) < TOL


def test_head_size_64_two_term_attention_against_the_fp32_mfma_kernels(torch_cuda):
    """Round 5 (attention_split64.hip): head size 64 on the fp16 matrix pipe with two-term operands (static bounds of q / k / v from the
    LayerNorm and the projections, as for head size 36) against the fp32-MFMA kernels it replaces (MI355ASR_ATTN64_SPLIT=0 in a
    subprocess): a dmodel-256 block at 260 / 250 / 64 frames and a two-block ConformerM encoder from the waveform -- both builds
    within the usual distance of the fp64 oracle, the two-term kernel at most twice the fp32 kernels' + 2e-7 of the output's scale."""
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, "tests")
from helpers import co, encoder_kwargs, maxdiff, small_cfg, waves
from tensorflowasr_amd.models import ConformerCTC
res = []
cfg = small_cfg(2, co.CONFORMER_M)
w = co.encoder_weights(cfg, seed=15)
w.update(co.ctc_decoder_weights(cfg, 100, seed=16))
m = ConformerCTC(100, **{k: v for k, v in encoder_kwargs(cfg).items() if k != "mel_layer_type"})
m.load_weights(w, by_name=False)
for T in (260, 250, 64):
    x = (3.0 * np.random.default_rng(T).standard_normal((3, T, cfg["dmodel"]))).astype(np.float32)
    ref = co.conformer_block(x.astype(np.float64), w, "conformer_block_1", cfg["head_size"], cfg["fc_factor"])
    res += [maxdiff(m.conformer_block(1, x).cpu().numpy(), ref), float(np.abs(ref).max())]
wav = waves(2, 64000, 77)
ref = co.conformer_encoder(wav.astype(np.float64), w, cfg)
res += [maxdiff(m.encode(wav).cpu().numpy(), ref), float(np.abs(ref).max())]
print("RESULT " + " ".join("%.4e" % v for v in res))
'''
    errs = {}
    for tag, extra in (("two", {}), ("f32", {"MI355ASR_ATTN64_SPLIT": "0"})):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=900,
                             cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT")]
        assert line, (tag, out.stderr[-2000:])
        errs[tag] = [float(v) for v in line[0].split()[1:]]
    print(errs)
    for i in range(0, 8, 2):
        e2, e1, scale = errs["two"][i], errs["f32"][i], errs["f32"][i + 1]
        assert e2 < TOL and e1 < TOL
        assert e2 <= 2.0 * e1 + 2e-7 * scale, (i, e2, e1, scale)


def test_ring_gemm_path_of_m_and_l_in_a_subprocess(torch_cuda):
    """gemm_ring.hip (dense layers of dmodel 256 / 512 on the split-bf16 pipe, taken from 1500 rows on) forced for a small
    batch (MI355ASR_RING_MIN_M=1): encoder and CTC logits against the oracle for ConformerM and ConformerL, with one and
    with two row tiles per wave (partial tiles in both), the two-slot ring of the one-tile shape, several column chunks per workgroup, and the fp32 kernels it replaces (MI355ASR_GEMM_RING=0)."""
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, "tests")
from helpers import co, encoder_kwargs, maxdiff, small_cfg, waves
from tensorflowasr_amd.models import ConformerCTC
res = []
for base, L in ((co.CONFORMER_M, 32000), (co.CONFORMER_L, 24000)):
    cfg = small_cfg(2, base)
    w = co.encoder_weights(cfg, seed=41)
    w.update(co.ctc_decoder_weights(cfg, 200, seed=42))
    m = ConformerCTC(200, **{k: v for k, v in encoder_kwargs(cfg).items() if k != "mel_layer_type"})
    m.load_weights(w, by_name=False)
    x = waves(3, L, 23)
    enc_ref = co.conformer_encoder(x.astype(np.float64), w, cfg)
    logits_ref = co.ctc_decoder(enc_ref, w, cfg)
    enc = m.encode(x)
    logits, amax = m.ctc_logits(enc, return_argmax=True)
    assert (amax.cpu().numpy() == logits.cpu().numpy().argmax(-1)).all()      # the head's fused arg-max (first maximum)
    ids, lens = m.recognize(x)                                                  # logits not materialised on this path
    rid, rlen = co.ctc_greedy(logits.cpu().numpy(), [logits.shape[1]] * 3, 199)
    assert (ids.cpu().numpy() == rid).all() and (lens.cpu().numpy() == rlen).all()
    res += [maxdiff(enc.cpu().numpy(), enc_ref), maxdiff(logits.cpu().numpy(), logits_ref)]
print("RESULT " + " ".join("%.3e" % v for v in res))
'''
    for extra in ({"MI355ASR_RING_MIN_M": "1", "MI355ASR_RING_RT": "1"}, {"MI355ASR_RING_MIN_M": "1", "MI355ASR_RING_RT": "2"},
                  {"MI355ASR_RING_MIN_M": "1", "MI355ASR_RING_SLOTS": "2"},
                  {"MI355ASR_RING_MIN_M": "1", "MI355ASR_RING_RT": "1", "MI355ASR_RING_CPW": "2"},
                  {"MI355ASR_RING_MIN_M": "1", "MI355ASR_RING_RT": "2", "MI355ASR_RING_CPW": "8"}, {"MI355ASR_GEMM_RING": "0"}):
        env = dict(os.environ, **extra)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900,
                             cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT")]
        assert line, out.stderr[-2000:]
        errs = [float(v) for v in line[0].split()[1:]]
        print(extra, errs)
        assert max(errs) < TOL, (extra, errs)


def test_expected_rows_hint_keeps_a_small_batch_handle_off_the_ring_packs(torch_cuda):
    """mi355asr_set_expected_rows (before finalisation): a dmodel-256 handle that announces 300 rows packs no slab rings
    and runs the fused fp32 chains -- same outputs within the oracle tolerance; the hint is refused once the weights are
    finalised."""
    from tensorflowasr_amd import _lib
    from tensorflowasr_amd.models import ConformerCTC
    cfg = small_cfg(2, co.CONFORMER_M)
    w = co.encoder_weights(cfg, seed=43)
    w.update(co.ctc_decoder_weights(cfg, 120, seed=44))
    m = ConformerCTC(120, **{k: v for k, v in encoder_kwargs(cfg).items() if k != "mel_layer_type"})
    m._h.set_expected_rows(300)
    m.load_weights(w, by_name=False)
    x = waves(2, 24000, 29)
    enc_ref = co.conformer_encoder(x.astype(np.float64), w, cfg)
    enc = m.encode(x)
    assert maxdiff(enc.cpu().numpy(), enc_ref) < TOL
    assert maxdiff(m.ctc_logits(enc).cpu().numpy(), co.ctc_decoder(enc_ref, w, cfg)) < TOL
    with pytest.raises(_lib.Mi355AsrError):
        m._h.set_expected_rows(5000)


def test_ring_gemm_bf16_mode_in_a_subprocess(torch_cuda):
    """bf16 mode (BASELINE config 3) through gemm_ring.hip's one-term ring, forced for a small batch: one ConformerBlock
    of the streaming configuration (dmodel 256) on an exact fp32 input against the oracle with both GEMM operands rounded
    to bf16 -- the same bounds as test_bf16_block_matches_rounding_oracle, which runs the per-wave bf16 kernels."""
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, "tests")
from helpers import co, encoder_kwargs, small_cfg
from tensorflowasr_amd.models import ConformerCTC
cfg = small_cfg(1, co.STREAMING_S)
w = co.encoder_weights(cfg, seed=5)
w.update(co.ctc_decoder_weights(cfg, 100, seed=6))
m = ConformerCTC(100, gemm_dtype="bfloat16", **{k: v for k, v in encoder_kwargs(cfg).items() if k != "mel_layer_type"})
m.load_weights(w, by_name=False)
x = np.random.default_rng(2).standard_normal((5, 77, cfg["dmodel"])).astype(np.float32)
got = m.conformer_block(0, x).cpu().numpy()
co.GEMM_ROUND_BF16 = True
ref = co.conformer_block(x.astype(np.float64), w, "conformer_block_0", cfg["head_size"], cfg["fc_factor"])
co.GEMM_ROUND_BF16 = False
exact = co.conformer_block(x.astype(np.float64), w, "conformer_block_0", cfg["head_size"], cfg["fc_factor"])
e = np.abs(got - ref)
print("RESULT %.3e %.3e %.3e" % (e.max(), e.mean(), np.abs(got - exact).max()))
'''
    for extra in ({"MI355ASR_RING_MIN_M": "1", "MI355ASR_RING_RT": "1"}, {"MI355ASR_RING_MIN_M": "1", "MI355ASR_RING_RT": "2"},
                  {"MI355ASR_RING_MIN_M": "1", "MI355ASR_RING_SLOTS": "2"}):
        env = dict(os.environ, **extra)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900,
                             cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT")]
        assert line, out.stderr[-2000:]
        emax, emean, vs_exact = (float(v) for v in line[0].split()[1:])
        print(extra, emax, emean, vs_exact)
        assert emax < 6e-3 and emean < 3e-4 and vs_exact > 10 * emean, (extra, emax, emean, vs_exact)


def test_ring_class_head_split_over_class_ranges(torch_cuda):
    """Round 6: the class head of dmodel 256 / 512 (gemm_ring_kernel<E16_HEAD>) needs every class of a row for its arg-max and ran as
    ceil(M / 128) workgroups -- 130 on 256 CUs at config 3's 16 640 rows.  With scratch for per-range (maximum, class) pairs the
    launcher gives a row tile's 128-class chunks to several workgroups and launch_head_combine takes the best range (lowest class on
    equal maxima, as within a range).  CTCDecoder (1 block, dmodel 256, 1332 classes = 11 chunks) over 64 x 260 rows in bf16 mode and
    in the three-term fp32 mode: arg-max and logits with 2 / 3 / 4 ranges and the launcher's own choice bit-identical to one range,
    with and without the logits written; duplicated class columns (ties across ranges) resolve to the lower class."""
    import subprocess
    import sys
    import tempfile
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, "tests")
from helpers import co
from tensorflowasr_amd.models import CTCDecoder
mode, out = sys.argv[1], sys.argv[2]
cfg = dict(co.STREAMING_S)
V = 1332
w = co.ctc_decoder_weights(cfg, V, seed=54)
k = w["fully_connected/kernel"]
k[:, 1200] = k[:, 7]; w["fully_connected/bias"][1200] = w["fully_connected/bias"][7]          # a tie between chunks 0 and 9
ctc = CTCDecoder(num_classes=V, dmodel=256, num_blocks=1, head_size=64, num_heads=4, kernel_size=32, fc_factor=0.5,
                 **({"gemm_dtype": "bfloat16"} if mode == "bf16" else {}))
ctc.load_weights(w, by_name=False)
h = torch.from_numpy(np.random.default_rng(3).standard_normal((64, 260, 256)).astype(np.float32)).cuda()
lg, am = ctc(h, return_argmax=True)
_, am2 = ctc(h, return_argmax=True, return_logits=False)
lgn = lg.cpu().numpy()
np.savez(out, logits=lgn[::7], amax=am.cpu().numpy(), amax_nolog=am2.cpu().numpy(), amax_of_logits=np.argmax(lgn, axis=-1).astype(np.int32))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        for mode in ("bf16", "f32"):
            res = {}
            for nr in ("1", "2", "3", "4", "", "rows"):
                # (bf16 mode: from 8 192 rows the head runs on gemm256_bf16_kernel -- "rows" -- unless MI355ASR_GEMM256=0)
                env = dict(os.environ, **({"MI355ASR_RING_HEAD_RANGES": nr} if nr not in ("", "rows") else {}), **({} if nr == "rows" else {"MI355ASR_GEMM256": "0"}))
                f = os.path.join(td, "%s_%s.npz" % (mode, nr or "auto"))
                r = subprocess.run([sys.executable, "-c", code, mode, f], env=env, capture_output=True, text=True, timeout=900, cwd=root)
                assert r.returncode == 0, r.stderr[-3000:]
                res[nr] = np.load(f)
            one = res["1"]
            assert np.array_equal(one["amax"], one["amax_of_logits"])
            assert (one["amax"] == 7).any() and not (one["amax"] == 1200).any()          # the duplicated column never wins over its lower twin
            for nr, r in res.items():
                if nr == "rows" and mode == "bf16":
                    # the whole decoder ran on the rows-resident kernels: other LayerNorm reduction orders in front of the head
                    assert np.array_equal(r["amax"], r["amax_of_logits"]) and np.array_equal(r["amax_nolog"], r["amax"]), (mode, nr)
                    assert not (r["amax"] == 1200).any() and np.abs(r["logits"] - one["logits"]).mean() < 2e-4      # (measured ~9e-5)
                    continue
                assert np.array_equal(r["amax"], one["amax"]) and np.array_equal(r["amax_nolog"], one["amax"]), (mode, nr)
                assert np.array_equal(r["logits"], one["logits"]), (mode, nr)


def test_streaming_block_stack_in_one_launch_vs_layer_at_a_time_and_rounding_oracle(torch_cuda):
    """Round 5 (stream256.hip): in bf16 mode the streaming encoder's whole block stack -- 4 ConformerBlocks, dmodel 256, chunks of
    13 rows -- is ONE launch, one workgroup per chunk (weights streamed from L2, the residual rows in registers, attention and
    depthwise conv from LDS).  Against the oracle with both GEMM operands rounded to bf16 it must meet the bounds the layer-at-a-time
    path is held to (same operands, another summation order along K: a hidden value on a rounding boundary may fall the other
    way); the same model with MI355ASR_STREAM256=0 is run beside it (both in subprocesses: the switch is read once), and the
    profile counters say which kernels ran.  Chunks of 8000 samples (13 rows), 4000 (7 rows: more padding rows in the tile), and the two
    ends of the kernel's range (round-5 advice): 10240 (16 rows, no padding row) and 640 (ONE row, fifteen padding rows)."""
    import subprocess
    import sys
    code = r'''
import sys, ctypes, numpy as np
sys.path.insert(0, "tests")
from helpers import co, waves
from tensorflowasr_amd import _lib
from tensorflowasr_amd.models import StreamingConformerEncoder
cfg = dict(co.STREAMING_S)
w = co.encoder_weights(cfg, seed=61)
lib = _lib.lib()
for chunk, nchunks in ((8000, 20), (4000, 6), (10240, 5), (640, 9)):
    enc = StreamingConformerEncoder(dmodel=256, reduction_factor=4, num_blocks=4, head_size=64, num_heads=4, kernel_size=5, fc_factor=0.5,
                                    sample_rate=16000, n_mels=80, stride_ms=10, mel_layer_type="Melspectrogram", gemm_dtype="bfloat16")
    enc.add_chunk_size(chunk, 80, 640)
    enc.load_weights(w, by_name=False)
    x = waves(nchunks, chunk, 500)
    _lib.check(lib.mi355asr_profile_enable(enc._h.ptr, 1))
    got = enc(x).cpu().numpy()
    nk = len(_lib.KERNEL_NAMES)
    ms, cnt = (ctypes.c_double * nk)(), (ctypes.c_int64 * nk)()
    _lib.check(lib.mi355asr_profile_read(enc._h.ptr, ms, cnt, nk, 1))
    launches = {n: int(cnt[i]) for i, n in enumerate(_lib.KERNEL_NAMES) if cnt[i]}
    co.GEMM_ROUND_BF16 = True
    ref = co.streaming_conformer_encoder(x.astype(np.float64), w, cfg, chunk)
    co.GEMM_ROUND_BF16 = False
    e = np.abs(got - ref)
    np.save(sys.argv[1] + "_%d.npy" % chunk, got)
    print("RESULT %d %d %.3e %.3e %d %d" % (chunk, got.shape[1], e.max(), e.mean(), launches.get("enc_stack", 0), launches.get("ffn", 0)))
'''
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        res = {}
        for tag, extra in (("stack", {}), ("layers", {"MI355ASR_STREAM256": "0"})):
            out = subprocess.run([sys.executable, "-c", code, os.path.join(td, tag)], env=dict(os.environ, **extra), capture_output=True, text=True,
                                 timeout=900, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            lines = [ln.split()[1:] for ln in out.stdout.splitlines() if ln.startswith("RESULT")]
            assert len(lines) == 4, out.stderr[-3000:]
            for chunk, rows, emax, emean, n_stack, n_ffn in lines:
                res[tag, int(chunk)] = (int(rows), float(emax), float(emean), int(n_stack), int(n_ffn))
        for chunk, rows in ((8000, 13), (4000, 7), (10240, 16), (640, 1)):       # 16: no padding row in the tile; 1: fifteen of them
            s_rows, s_max, s_mean, s_stack, s_ffn = res["stack", chunk]
            l_rows, l_max, l_mean, l_stack, l_ffn = res["layers", chunk]
            assert s_rows == l_rows == rows
            assert (s_stack, s_ffn) == (1, 0) and (l_stack, l_ffn) == (0, 8), (res["stack", chunk], res["layers", chunk])
            a, b = np.load(os.path.join(td, "stack_%d.npy" % chunk)), np.load(os.path.join(td, "layers_%d.npy" % chunk))
            d = np.abs(a - b)
            print("chunk %d (%d rows): one launch vs rounding oracle max %.3g mean %.3g; layer at a time %.3g / %.3g; apart max %.3g mean %.3g"
                  % (chunk, rows, s_max, s_mean, l_max, l_mean, d.max(), d.mean()))
            assert s_max < 2e-2 and s_mean < 2e-3 and l_max < 2e-2 and l_mean < 2e-3
            assert s_mean < 2 * l_mean + 1e-5 and d.max() < 4e-2 and d.mean() < 2e-3


def test_long_utterances_attention_in_key_blocks_against_the_fp32_kernels_and_the_oracle(torch_cuda):
    """Round 6 (attention_split.hip: attention_split_long_kernel): utterances of more than 256 encoder frames keep the two-term
    attention -- key blocks of 224, online softmax -- instead of falling to the fp32-MFMA kernels (multihead_attention.py:151-188 has no
    length limit).  Encoder (2 blocks) on 3 x 20 s (T = 500: three key blocks, a ragged last query tile) and 2 x 30 s (T = 750, second
    utterance scaled by 1e-3) against the fp64 oracle; the same in a build with MI355ASR_ATTN_LONG=0 (the old route) and with
    MI355ASR_ATTN_TERMS=3 (three exact bf16 terms): all within the contract, the two routes within 2e-5 of each other."""
    import subprocess
    import sys
    import tempfile
    code = r'''
import sys, numpy as np
sys.path.insert(0, "tests")
from helpers import co, encoder_kwargs, small_cfg, waves
from tensorflowasr_amd.models import ConformerEncoder
cfg = small_cfg(2)
w = co.encoder_weights(cfg, seed=8)
e = ConformerEncoder(**encoder_kwargs(cfg)); e.load_weights(w, by_name=False)
out = {}
for tag, B, L in (("t500", 3, 320000), ("t750", 2, 480000)):
    x = waves(B, L, 900)
    if tag == "t750": x[1] *= np.float32(1e-3)
    got = e(x).cpu().numpy()
    ref = co.conformer_encoder(x[:2].astype(np.float64), w, cfg)
    out[tag] = got
    print("RESULT %s %d %.3e" % (tag, got.shape[1], np.abs(got[:2] - ref).max()))
np.savez(sys.argv[1], **out)
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        res = {}
        for tag, extra in (("long", {}), ("fp32", {"MI355ASR_ATTN_LONG": "0"}), ("three", {"MI355ASR_ATTN_TERMS": "3"})):
            r = subprocess.run([sys.executable, "-c", code, os.path.join(td, tag + ".npz")], env=dict(os.environ, **extra), capture_output=True,
                               text=True, timeout=900, cwd=root)
            lines = [ln.split()[1:] for ln in r.stdout.splitlines() if ln.startswith("RESULT")]
            assert len(lines) == 2, r.stderr[-3000:]
            assert [int(l[1]) for l in lines] == [500, 750]
            errs = [float(l[2]) for l in lines]
            assert all(e < TOL for e in errs), (tag, errs)
            res[tag] = (errs, np.load(os.path.join(td, tag + ".npz")))
        for k in ("t500", "t750"):
            d1 = float(np.abs(res["long"][1][k] - res["fp32"][1][k]).max())
            d3 = float(np.abs(res["long"][1][k] - res["three"][1][k]).max())
            print("%s: key-block kernel vs fp32-MFMA route %.3g, vs three-term %.3g; from the oracle %s / %s / %s"
                  % (k, d1, d3, res["long"][0], res["fp32"][0], res["three"][0]))
            assert 0.0 < d1 < 2e-5 and d3 < 2e-5


def test_small_batches_one_tile_per_workgroup_against_the_pair_pipelined_kernels_and_the_oracle(torch_cuda):
    """Round 6 (fused_ns.hip, ns1_* kernels; up to MI355ASR_NS1_MAX_M rows, default 4096): one utterance per call is test_asr.py's
    pattern (test_asr.py:186-219).  A workgroup of eight waves owns ONE 16-token tile, the hidden dimension and the column tiles of
    the plain layers are split over the waves, the block runs as attention + out-projection / GLU + (depthwise conv, tail, next
    ff_module_1 + qkv).  Same arithmetic as the pair-pipelined kernels, another summation order: encoder output, logits and greedy ids
    of 1 x 10 s, 3 x 3.7 s (a ragged last tile per utterance: 92 frames), 2 x 2.8 s and 12 x 10 s against the build with MI355ASR_NS1_MAX_M=0
    and against the fp64 oracle; the profile counters say which kernels ran (three launches of the tail category per block pair)."""
    import subprocess
    import sys
    import tempfile
    code = r'''
import sys, ctypes, numpy as np, torch
sys.path.insert(0, "tests")
from helpers import co, encoder_kwargs, small_cfg, waves, golden_ctc_weights
from tensorflowasr_amd import _lib
from tensorflowasr_amd.models import ConformerCTC
cfg = small_cfg(3)
w = co.encoder_weights(cfg, seed=0); w.update(golden_ctc_weights())
m = ConformerCTC(1332, **{k: v for k, v in encoder_kwargs(cfg).items() if k != "mel_layer_type"})
m.load_weights(w, by_name=False)
out = {}
for tag, B, L in (("b1", 1, 160000), ("b3", 3, 59000), ("b2", 2, 45000), ("b12", 12, 160000)):     # b12: 3 000 rows -- the class head folds into the CTC block's pair-pipelined tail from 2 048 rows, the encoder's blocks stay on the one-tile kernels
    x = waves(B, L, 40)
    enc = m.encode(x); lg = m.ctc_logits(enc)
    ids, lens = m.recognize(x)
    out[tag + "_enc"] = enc.cpu().numpy(); out[tag + "_lg"] = lg.cpu().numpy(); out[tag + "_ids"] = ids.cpu().numpy()
    ref = co.conformer_encoder(x[:1].astype(np.float64), w, cfg)
    lref = co.ctc_decoder(ref, w, cfg)
    print("RESULT %s %.3e %.3e" % (tag, np.abs(out[tag + "_enc"][:1] - ref).max(), np.abs(out[tag + "_lg"][:1] - lref).max()))
np.savez(sys.argv[1], **out)
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        res = {}
        for tag, extra in (("ns1", {}), ("ns1_attn_own", {"MI355ASR_NS1_ATTN": "0"}), ("pp", {"MI355ASR_NS1_MAX_M": "0"})):
            r = subprocess.run([sys.executable, "-c", code, os.path.join(td, tag + ".npz")], env=dict(os.environ, **extra), capture_output=True,
                               text=True, timeout=900, cwd=root)
            lines = [ln.split()[1:] for ln in r.stdout.splitlines() if ln.startswith("RESULT")]
            assert len(lines) == 4, r.stderr[-3000:]
            assert all(float(l[1]) < TOL and float(l[2]) < TOL for l in lines), (tag, lines)
            res[tag] = (lines, np.load(os.path.join(td, tag + ".npz")))
        for other in ("pp", "ns1_attn_own"):          # (the attention inside the out-projection launch sums its key tiles in another order)
            for k in res["ns1"][1].files:
                a, b = res["ns1"][1][k], res[other][1][k]
                if k.endswith("_ids"):
                    assert np.array_equal(a, b), (other, k)
                else:
                    apart = float(np.abs(a - b).max())
                    assert 0.0 < apart < 1e-4, (other, k, apart)
        print("one tile per workgroup vs pair-pipelined: oracle distances %s / %s" % (res["ns1"][0], res["pp"][0]))


@pytest.mark.parametrize("dm,H,hs,k,B,L", [(144, 3, 48, 7, 2, 16000), (144, 6, 24, 16, 2, 16000), (256, 8, 32, 9, 2, 16000), (144, 12, 12, 3, 1, 32000),
                                           (144, 2, 72, 31, 40, 80000), (512, 4, 128, 4, 1, 16000)])
def test_constructor_surface_beyond_the_shipped_yamls(torch_cuda, dm, H, hs, k, B, L):
    """Round 6 (review: "a user config that does not fit gets an error, not a slow path"): conformer_blocks.py:278-294 takes any head
    size and kernel size.  Head sizes 12 / 16 / 24 / 32 / 48 / 72 / 128 run on the online-softmax attention kernel, any kernel size on
    dwconv_any_kernel (even sizes: Keras 'same' pads (k - 1) // 2 in front), under the fused block kernels where dmodel is 144
    ((144, 2, 72, 31) at 40 x 5 s = 5 000 rows: the pair-pipelined kernels, below: one tile per workgroup).  Encoder (1 block) + CTC
    decoder against the oracle."""
    from tensorflowasr_amd.models import ConformerCTC
    cfg = dict(co.CONFORMER_S, dmodel=dm, num_heads=H, head_size=hs, kernel_size=k, num_blocks=1, ctcdecoder_kernel_size=k)
    w = co.encoder_weights(cfg, seed=1)
    w.update(co.ctc_decoder_weights(cfg, 70, seed=2))
    m = ConformerCTC(70, dmodel=dm, num_blocks=1, head_size=hs, num_heads=H, kernel_size=k, ctcdecoder_kernel_size=k)
    m.load_weights(w, by_name=False)
    x = waves(B, L, 3)
    enc = m.encode(x)
    ref = co.conformer_encoder(x[:2].astype(np.float64), w, cfg)
    assert maxdiff(enc.cpu().numpy()[:2], ref) < TOL
    assert maxdiff(m.ctc_logits(enc).cpu().numpy()[:2], co.ctc_decoder(ref, w, cfg)) < TOL


def test_constructor_rejects_what_no_kernel_covers(torch_cuda):
    from tensorflowasr_amd import _lib
    from tensorflowasr_amd.models import ConformerEncoder
    with pytest.raises(_lib.Mi355AsrError, match="head_size"):
        ConformerEncoder(dmodel=144, num_blocks=1, head_size=18, num_heads=8)
    with pytest.raises(_lib.Mi355AsrError, match="reduction_factor"):
        ConformerEncoder(dmodel=144, num_blocks=1, reduction_factor=3)     # conformer_blocks.py:75 asserts an even factor


@pytest.mark.parametrize("rf,dm,H,hs,B,L", [(2, 144, 4, 36, 2, 8000), (6, 144, 4, 36, 3, 32000), (8, 144, 4, 36, 2, 48000), (2, 256, 4, 64, 1, 8000),
                                            (8, 512, 8, 64, 2, 32000), (6, 144, 4, 36, 1, 16000 - 77)])
def test_reduction_factors_other_than_4(torch_cuda, rf, dm, H, hs, B, L):
    """conformer_blocks.py:76-80: conv1's time stride is reduction_factor // 2 (the frequency stride stays 2).  Factors 2, 6 and 8 (strides
    1, 3, 4) run on subconv_split_kernel<4, ST1> (mel window 2 ST1 + 3 rows) -- every shipped config has 4 and the faster kernels.
    Encoder (1 block) + CTC decoder against the oracle; the frame count follows ceil(ceil(F / st1) / 2)."""
    from tensorflowasr_amd.models import ConformerCTC
    cfg = dict(co.CONFORMER_S, dmodel=dm, num_heads=H, head_size=hs, num_blocks=1, reduction_factor=rf)
    w = co.encoder_weights(cfg, seed=5)
    w.update(co.ctc_decoder_weights(cfg, 70, seed=6))
    m = ConformerCTC(70, dmodel=dm, num_blocks=1, head_size=hs, num_heads=H, reduction_factor=rf)
    m.load_weights(w, by_name=False)
    x = waves(B, L, 4)
    enc = m.encode(x)
    ref = co.conformer_encoder(x.astype(np.float64), w, cfg)
    F = -(-L // 160)
    assert ref.shape[1] == -(-(-(-F // (rf // 2))) // 2) and tuple(enc.shape) == ref.shape
    assert maxdiff(enc.cpu().numpy(), ref) < TOL
    assert maxdiff(m.ctc_logits(enc).cpu().numpy(), co.ctc_decoder(ref, w, cfg)) < TOL


def test_translator_dmodel_512(torch_cuda):
    """conformerL.yml Translator (dmodel 512, 8 heads x 64): cross-attention through the layer-at-a-time GEMM path."""
    from tensorflowasr_amd.models import Translator
    cfg = dict(co.CONFORMER_L, translator_num_blocks=1, translator_kernel_size=32, translator_fc_factor=0.5)
    w = co.translator_weights(cfg, 60, 75, seed=9)
    tr = Translator(inp_classes=60, tar_classes=75, dmodel=512, num_blocks=1, head_size=64, num_heads=8, kernel_size=32)
    tr.load_weights(w, by_name=False)
    rng = np.random.default_rng(3)
    ids = rng.integers(0, 60, (2, 19)).astype(np.int32)
    enc = rng.standard_normal((2, 77, 512)).astype(np.float32)
    ref = co.translator(ids, enc.astype(np.float64), w, cfg)
    got, amax = tr([ids, enc], return_argmax=True)
    assert maxdiff(got.cpu().numpy(), ref) < TOL
    assert (amax.cpu().numpy() == got.cpu().numpy().argmax(-1)).all()


# ---------------------------------------------------------------------------------------------------------
# LEAF frontend (mel_layer_type 'leaf')
# ---------------------------------------------------------------------------------------------------------
def _leaf_weights(cfg, seed, trained=True):
    w = co.encoder_weights(cfg, seed=seed)
    for k in [k for k in w if k.startswith("mel_layer/")]:
        del w[k]
    lw = co.leaf_default_weights()
    if trained:                                         # move every learnable off its initial value
        rng = np.random.default_rng(seed)
        lw["mel_layer/tfbanks_preemp/kernel"] = np.array([-0.93, 1.02], np.float32).reshape(2, 1, 1)
        k = lw["mel_layer/tfbanks_complex_conv/kernel"]
        k[:, 0] *= rng.uniform(0.97, 1.03, 80).astype(np.float32)
        k[:, 1] *= rng.uniform(0.8, 1.2, 80).astype(np.float32)
        lw["mel_layer/learnable_pooling/kernel"] = rng.uniform(0.25, 0.6, (1, 1, 80, 1)).astype(np.float32)
        lw["mel_layer/PCEN/alpha"] = rng.uniform(0.8, 1.05, 80).astype(np.float32)          # > 1 is clipped
        lw["mel_layer/PCEN/delta"] = rng.uniform(1.0, 3.0, 80).astype(np.float32)
        lw["mel_layer/PCEN/root"] = rng.uniform(0.9, 3.0, 80).astype(np.float32)            # < 1 is clipped
        lw["mel_layer/PCEN/EMA/smooth"] = rng.uniform(0.02, 0.08, 80).astype(np.float32)
        lw["mel_layer/tfbanks_instancenorm/gamma"] = rng.uniform(0.5, 1.5, 80).astype(np.float32)
        lw["mel_layer/tfbanks_instancenorm/beta"] = (0.2 * rng.standard_normal(80)).astype(np.float32)
    w.update(lw)
    return w


@pytest.mark.parametrize("terms", ["3", "2", "0"])
@pytest.mark.parametrize("L,trained", [(16000, True), (24160, False), (8100, True), (1000, True), (512 * 9 + 3, True)])
def test_leaf_frontend_parity(torch_cuda, monkeypatch, L, trained, terms):
    """Gabor conv + squared modulus + Gaussian pooling + PCEN + instance norm against the oracle; lengths that are
    and are not multiples of the hop (the SAME padding of the pooling depends on L).  terms: the Gabor conv with fp32
    operands split into 3 (default) or 2 bf16 terms, or the fp32-MFMA kernel (0)."""
    from tensorflowasr_amd.models import ConformerEncoder
    monkeypatch.setenv("MI355ASR_LEAF_TERMS", terms)
    cfg = small_cfg(1)
    w = _leaf_weights(cfg, 7, trained)
    e = ConformerEncoder(**dict(encoder_kwargs(cfg), mel_layer_type="leaf"))
    e.load_weights(w, by_name=False)
    assert e._h.lib.mi355asr_stft_mode(e._h.ptr) == -1
    x = waves(2, L, 60)
    ref = co.leaf_frontend(x.astype(np.float64), w)
    got = e.melspectrogram(x).cpu().numpy()
    assert got.shape == ref.shape == (2, -(-L // 160), 80)
    print("leaf terms=%s L=%d trained=%s: max |gpu - oracle| = %.3g" % (terms, L, trained, maxdiff(got, ref)))
    assert maxdiff(got, ref) < (TOL if terms != "3" else 2e-4)
    enc_ref = co.conformer_block(co.conv_subsampling(ref, w), w, "conformer_block_0", cfg["head_size"], cfg["fc_factor"])
    assert maxdiff(e(x).cpu().numpy(), enc_ref) < TOL


def test_leaf_streaming_blocks_and_full_batch(torch_cuda):
    """LEAF under the Block Conformer (every 0.5 s block is its own utterance for the frontend) and at B = 8 x 10 s."""
    from tensorflowasr_amd.models import ConformerEncoder, StreamingConformerEncoder
    cfg = small_cfg(1, co.STREAMING_S)
    w = _leaf_weights(cfg, 8)
    st = StreamingConformerEncoder(**dict(encoder_kwargs(cfg), mel_layer_type="leaf"))
    st.add_chunk_size(8000, 80, 640)
    st.load_weights(w, by_name=False)
    x = waves(2, 16000, 70)
    mel = st.melspectrogram(x).cpu().numpy()                     # [B * 2 blocks, 50, 80]
    ref = co.leaf_frontend(x.reshape(4, 8000).astype(np.float64), w)
    assert maxdiff(mel, ref) < TOL
    cfg2 = small_cfg(1)
    w2 = _leaf_weights(cfg2, 9)
    e = ConformerEncoder(**dict(encoder_kwargs(cfg2), mel_layer_type="leaf"))
    e.load_weights(w2, by_name=False)
    xb = waves(8, 160000, 80)
    got = e.melspectrogram(xb).cpu().numpy()
    assert np.isfinite(got).all() and got.shape == (8, 1000, 80)
    assert maxdiff(got[:1], co.leaf_frontend(xb[:1].astype(np.float64), w2)) < TOL
    assert np.abs(got.mean(axis=1) - w2["mel_layer/tfbanks_instancenorm/beta"]).max() < 1e-3   # instance norm: mean = beta


# ---------------------------------------------------------------------------------------------------------
# add_wav_info: WavePickModel branch added to the subsampled features (wav_model.py:108-146)
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("base,L,B", [(None, 16000, 3), (None, 640 * 7, 1), ("M", 32000, 2), ("STREAMING_S", 8000, 4),
                                      ("20ms", 1280 * 9, 2)])
def test_add_wav_info_encoder_parity(torch_cuda, base, L, B):
    """Encoder with the waveform branch against the oracle, and the branch alone as the difference of two encoders
    without conformer blocks (dmodel 144 -> stage channels 64/96/128; 256 -> the same, final conv 7*128 -> 256)."""
    from tensorflowasr_amd.models import ConformerEncoder
    cfg = small_cfg(1, {"M": co.CONFORMER_M, "STREAMING_S": co.STREAMING_S, "20ms": dict(co.CONFORMER_S, stride_ms=20)}.get(base))
    w = co.encoder_weights(cfg, seed=21)
    hop = cfg["stride_ms"] * 16 * cfg.get("reduction_factor", 4)        # 640 -> strides [8,5,4,4]; 1280 -> [16,5,4,4]
    w.update(co.wave_pick_weights(cfg["dmodel"], hop, seed=22))
    x = waves(B, L, 90)
    e = ConformerEncoder(**dict(encoder_kwargs(cfg), mel_layer_type="Melspectrogram", add_wav_info=True))
    e.load_weights(w, by_name=False)
    ref = co.conformer_encoder(x.astype(np.float64), w, dict(cfg, add_wav_info=True))
    got = e(x).cpu().numpy()
    assert got.shape == ref.shape
    assert maxdiff(got, ref) < TOL
    # the branch itself: no blocks, with minus without
    cfg0 = dict(cfg, num_blocks=0)
    w0 = {k: v for k, v in w.items() if not k.startswith("conformer_block_")}
    ea = ConformerEncoder(**dict(encoder_kwargs(cfg0), mel_layer_type="Melspectrogram", add_wav_info=True))
    ea.load_weights(w0, by_name=False)
    eb = ConformerEncoder(**dict(encoder_kwargs(cfg0), mel_layer_type="Melspectrogram"))
    eb.load_weights({k: v for k, v in w0.items() if not k.startswith("wav_layer/")}, by_name=False)
    branch = ea(x).cpu().numpy().astype(np.float64) - eb(x).cpu().numpy()
    ref_b = co.wave_pick_model(x.astype(np.float64), w, cfg["dmodel"], hop)
    assert np.abs(ref_b).max() > 0.05                    # the branch is not negligible in the sum
    assert maxdiff(branch, ref_b) < TOL


def test_add_wav_info_with_leaf_ctc_and_errors(torch_cuda):
    """The reference's newest configuration: leaf frontend + waveform branch, through ConformerCTC."""
    from tensorflowasr_amd.models import ConformerCTC
    cfg = small_cfg(1)
    V = 40
    w = _leaf_weights(cfg, 31)
    w.update(co.wave_pick_weights(cfg["dmodel"], 640, seed=32))
    w.update(co.ctc_decoder_weights(cfg, V, seed=33))
    kw = dict(encoder_kwargs(cfg))
    kw.pop("mel_layer_type")
    m = ConformerCTC(V, ctcdecoder_num_blocks=cfg["ctcdecoder_num_blocks"], mel_layer_type="leaf", add_wav_info=True, **kw)
    m.load_weights(w, by_name=False)
    x = waves(2, 16000, 95)
    enc_ref = co.conformer_encoder(x.astype(np.float64), w, dict(cfg, mel_layer_type="leaf", add_wav_info=True))
    enc = m.encode(x)
    assert maxdiff(enc.cpu().numpy(), enc_ref) < TOL
    logits_ref = co.ctc_decoder(enc_ref, w, cfg)
    assert maxdiff(m.ctc_logits(enc).cpu().numpy(), logits_ref) < 2e-3
    ids, lens = m.recognize(x)
    T = logits_ref.shape[1]
    lg, am = m.ctc_logits(enc, return_argmax=True)
    assert_frames_and_ids(lg.cpu().numpy(), am.cpu().numpy(), ids.cpu().numpy(), lens.cpu().numpy(), logits_ref,
                          np.full(2, T), V - 1, max_undecided=0.05, tol=2e-3)
    # a length that is not a multiple of hop_size: both branches yield ceil(L / hop_size) frames (nested SAME strides)
    xr = waves(1, 16000 + 160, 96)
    ref_r = co.conformer_encoder(xr.astype(np.float64), w, dict(cfg, mel_layer_type="leaf", add_wav_info=True))
    assert ref_r.shape[1] == 26
    assert maxdiff(m.encode(xr).cpu().numpy(), ref_r) < TOL


def test_opt_in_kernel_variants_in_a_subprocess(torch_cuda):
    """Every kernel-choice switch that survives round 4's pruning (tests/test_host.py::test_environment_switches_...
    lists them; the library reads each once per process): the fp32-MFMA fallbacks, the three-term bf16 versions of the kernels
    that default to two fp16 terms, and the structural fallbacks (separate launches instead of folded prologues), with the
    fused block path forced for a small batch (MI355ASR_SMALL_M=0).  Each must agree with the oracle like the defaults."""
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, "tests")
from helpers import co, encoder_kwargs, maxdiff, small_cfg, waves
from tensorflowasr_amd.models import ConformerEncoder
cfg = small_cfg(2)
w = co.encoder_weights(cfg, seed=3)
e = ConformerEncoder(**encoder_kwargs(cfg)); e.load_weights(w, by_name=False)
rng = np.random.default_rng(5)
x = rng.standard_normal((3, 250, 144)).astype(np.float32)
blk = maxdiff(e.conformer_block(1, x).cpu().numpy(), co.conformer_block(x.astype(np.float64), w, "conformer_block_1", 36))
wav = waves(2, 16000, 11)
enc = maxdiff(e(wav).cpu().numpy(), co.conformer_encoder(wav.astype(np.float64), w, cfg))
print("RESULT %.3e %.3e" % (blk, enc))
'''
    for extra in ({"MI355ASR_OUTGLU_SPLIT": "0"}, {"MI355ASR_SUBCONV_F32": "1"},
                  {"MI355ASR_SUBLINEAR_SPLIT": "2"}, {"MI355ASR_SUBLINEAR_SPLIT": "0"},
                  {"MI355ASR_SUBLINEAR_SPLIT": "2", "MI355ASR_PP_SUBLINEAR": "0"}, {"MI355ASR_MEL_BAND": "0"},
                  {"MI355ASR_FF1QKV_RING": "0"}, {"MI355ASR_HEAD_RING": "0"}, {"MI355ASR_FUSED": "0"},
                  {"MI355ASR_TAILFF2_RING": "0"}, {"MI355ASR_PP": "0"}, {"MI355ASR_PP": "0", "MI355ASR_TAIL_FF1": "0"},
                  {"MI355ASR_TAIL_FF1": "0"}, {"MI355ASR_PP_DW": "0"}, {"MI355ASR_PP_OGF": "0"}, {"MI355ASR_PP_HEAD": "0"},
                  {"MI355ASR_SUBLINEAR_SPLIT": "2", "MI355ASR_PP_PRE": "0"},
                  {"MI355ASR_ATTN_SPLIT": "0"}, {"MI355ASR_ATTN_LDS": "0"}, {"MI355ASR_FFT_SPLIT": "0"}, {"MI355ASR_FFT": "0"},
                  # the three-term bf16 versions of the kernels that default to the two-term fp16 scheme
                  {"MI355ASR_SUBCONV_TERMS": "3"}, {"MI355ASR_ATTN_TERMS": "3"}, {"MI355ASR_PP_OUTGLU": "0"}, {"MI355ASR_FFT_TERMS": "3"},
                  {"MI355ASR_SUBCONV_TERMS": "3", "MI355ASR_ATTN_TERMS": "3", "MI355ASR_PP": "0"}):
        env = dict(os.environ, MI355ASR_SMALL_M="0", **extra)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600,
                             cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT")]
        assert line, (extra, out.stderr[-2000:])
        blk, enc = (float(v) for v in line[0].split()[1:])
        assert blk < TOL and enc < TOL, (extra, blk, enc)


def test_fused_block_path_at_short_utterances_in_a_subprocess(torch_cuda):
    """The fused dmodel-144 block kernels (taken above 800 rows by default) forced for every row count
    (MI355ASR_SMALL_M=0): many short utterances -- the shapes a large batch of short clips or of streaming chunks brings -- and
    utterance lengths around the 64-frame tiling of the depthwise-conv fold (63 / 64 / 65, below it the separate kernel),
    around the 16-key limit of the split attention kernel, and single utterances.  Each against the fp64 oracle."""
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, "tests")
from helpers import co, encoder_kwargs, maxdiff, small_cfg
from tensorflowasr_amd.models import ConformerEncoder
cfg = small_cfg(2)
w = co.encoder_weights(cfg, seed=3)
e = ConformerEncoder(**encoder_kwargs(cfg)); e.load_weights(w, by_name=False)
res = []
for B, T in ((64, 13), (40, 30), (12, 75), (3, 100), (2, 130), (5, 63), (5, 64), (5, 65), (3, 16), (3, 17), (70, 7), (1, 250)):
    rng = np.random.default_rng(B * 1000 + T)
    x = rng.standard_normal((B, T, 144)).astype(np.float32)
    ref = co.conformer_block(x.astype(np.float64), w, "conformer_block_1", 36)
    res.append(maxdiff(e.conformer_block(1, x).cpu().numpy(), ref))
print("RESULT " + " ".join("%.3e" % v for v in res))
'''
    env = dict(os.environ, MI355ASR_SMALL_M="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT")]
    assert line, out.stderr[-2000:]
    errs = [float(v) for v in line[0].split()[1:]]
    assert len(errs) == 12 and max(errs) < 1e-4, errs


def test_cpp_session_example_runs(torch_cuda):
    """examples/asr_session.cpp -- the reference's C++ Session on the C ABI, no Python in the process: enumerates the
    tensors with mi355asr_weight_shape, fills them, recognises a synthetic utterance twice with identical ids."""
    import subprocess
    from tensorflowasr_amd import build as b
    exe = b.build_example(verbose=False)                 # no-op when examples/asr_session is up to date
    assert exe and os.path.exists(exe)
    out = subprocess.run([exe, "3"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-1000:]
    assert "repeatable: yes" in out.stdout and "75 encoder frames" in out.stdout
