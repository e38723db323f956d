"""Golden vectors from the REFERENCE ITSELF for the rows of SURVEY section 8 that nothing in this container can pin
(no TensorFlow / librosa here): mel frontend (a2-a4), ConvSubsampling (a5), the encoders (a10, a11), CTCDecoder + greedy
ctc_decode (a12, a13), Translator (8f-1), LEAF and WavePickModel (8f-4), ChunkConformer.predict (a15).

Run on any machine that has the reference checkout, TensorFlow 2.8+ and librosa:

    REFERENCE_ROOT=/path/to/TensorflowASR python tests/golden/make_tf_goldens.py          # writes tests/golden/tf_*.npz

What it does, per component: builds the reference's OWN Keras object, overwrites every variable with this repository's
seeded weights (oracle.conformer_oracle.*_weights -- the same tensors the parity tests load into libmi355asr.so),
mapping Keras variable names with tensorflowasr_amd.checkpoint.keras_names_to_abi (ChunkConformer: object-graph
attribute paths with chunk_checkpoint_keys_to_abi) and FAILING if any variable of the model stays unassigned, runs it on
this repository's seeded inputs with training=False, and stores inputs' seeds and the outputs.  The fixed frontend
matrices the reference generates with librosa (DFT kernels, `freq2mel`) are stored as they come out of the reference:
they are weight inputs on our side (librosa's `norm=1` changed meaning across versions).

tests/test_tf_goldens.py picks the fixtures up when they exist: the NumPy oracle and (on the GPU box) libmi355asr.so are
both compared with them -- that is what turns those rows from "parity unpinned" into pinned.  Nothing here is imported
by the product; the fixtures are test data.
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.environ.get("MI355ASR_TF_GOLDEN_OUT", HERE)          # tests/test_tf_recipe_runs.py regenerates into a scratch directory
REF = os.environ.get("REFERENCE_ROOT", "/root/reference")
for p in (ROOT, os.path.join(ROOT, "tests"), REF):
    if p not in sys.path:
        sys.path.insert(0, p)

from helpers import chunk_config_dict, co, pick_bias_for_ragged_counts, waves  # noqa: E402
from tensorflowasr_amd import checkpoint  # noqa: E402


def _abi_of_keras_names(model, extra=None):
    m = checkpoint.keras_names_to_abi([v.name for v in model.weights])     # includes the mel layer's constants (melspectrogram/...)
    if extra:
        m.update(extra([v.name for v in model.weights]))
    return m


def assign_by_name(model, weights, keep=("mel_layer/real_kernels", "mel_layer/imag_kernels", "mel_layer/freq2mel")):
    """every variable of `model` <- weights[abi name]; the frontend constants in `keep` stay as the reference built them and
    are returned so that the fixture can carry them"""
    m = _abi_of_keras_names(model)
    kept, missing = {}, []
    for v in model.weights:
        abi = m.get(v.name)
        if abi in keep:
            kept[abi] = v.numpy()
            continue
        if abi is None or abi not in weights:
            missing.append((v.name, abi))
            continue
        v.assign(np.asarray(weights[abi], np.float32).reshape(v.shape))
    if missing:
        raise RuntimeError("variables without a seeded tensor (name mapping incomplete): %s" % missing[:8])
    return kept


def assign_by_object_path(model, weights, tf):
    """ChunkConformer: checkpoint keys are attribute paths from the model root; walk them to reach each variable"""
    with tempfile.TemporaryDirectory() as d:
        prefix = os.path.join(d, "w")
        model.save_weights(prefix)
        keys = [k for k, _ in tf.train.list_variables(prefix)]
    m = checkpoint.chunk_checkpoint_keys_to_abi(keys)
    kept, done = {}, 0
    for key, abi in m.items():
        obj = model
        for part in key.split("/.ATTRIBUTES/")[0].split("/"):
            obj = obj[int(part)] if part.isdigit() else getattr(obj, part)
        if abi.startswith("front/mel_layer/"):
            kept[abi] = obj.numpy()
            continue
        obj.assign(np.asarray(weights[abi], np.float32).reshape(obj.shape))
        done += 1
    expected = {k for k in weights if not k.startswith("front/mel_layer/")}
    if done != len(expected):
        raise RuntimeError("assigned %d of %d ChunkConformer tensors: unmapped %s"
                           % (done, len(expected), sorted(expected - set(m.values()))[:8]))
    return kept


FIXTURES = {}
SUFFIX = [""]            # "" = the float32 pass; "_f64" = the stand-in's wide pass (float outputs only, added to the same file)


GENERATOR = ["unknown"]   # "numpy stand-in (oracle/_tfshim) <version>" or "tensorflow <version>": stored in every fixture


def save(name, **arrays):
    fx = FIXTURES.setdefault(name, {})
    if not SUFFIX[0]:
        fx["meta_generator"] = np.asarray(GENERATOR[0])
    for k, v in arrays.items():
        v = np.asarray(v)
        if not SUFFIX[0]:
            fx[k] = v.astype(np.float32) if v.dtype == np.float64 else v
        elif v.dtype.kind == "f" and v.ndim > 0 and not k.startswith(("freq2mel", "real_kernels", "imag_kernels")):
            fx[k + SUFFIX[0]] = v.astype(np.float64)


def write_all():
    for name, fx in FIXTURES.items():
        path = os.path.join(OUT, name)
        np.savez_compressed(path, **fx)
        print("wrote %s (%d KB): %s" % (path, os.path.getsize(path) // 1024, sorted(fx)))


def import_tensorflow():
    """the real TensorFlow when there is one; otherwise the NumPy stand-in under oracle/_tfshim (round 5: the reference's own
    Python executed on stand-in primitives, see oracle/_tfshim/README.md)"""
    try:
        import tensorflow as tf
        if "standin" not in getattr(tf, "__version__", ""):
            return tf, False
        return tf, True
    except ImportError:
        sys.path.insert(0, os.path.join(ROOT, "oracle", "_tfshim"))
        import tensorflow as tf
        return tf, True


def main():
    tf, standin = import_tensorflow()
    from asr.models import conformer_blocks as cb
    from asr.models.layers.time_frequency import Melspectrogram
    meta = dict(tf_version=tf.__version__, backend="numpy stand-in (oracle/_tfshim)" if standin else "tensorflow")
    GENERATOR[0] = "%s %s" % (meta["backend"], tf.__version__)
    try:
        import librosa
        meta["librosa_version"] = librosa.__version__
    except Exception:
        pass
    done = []

    # ---- a2-a4: Melspectrogram layer (time_frequency.py:100-189) -------------------------------------------------
    for L in (32000, 67263):
        mel = Melspectrogram(sr=16000, n_mels=80, n_hop=160, n_dft=1024, trainable_fb=False)
        x = waves(2, L, 5)
        y = mel(tf.constant(x[..., None])).numpy()                      # [B, F, 80, 1]
        consts = {v.name: v.numpy() for v in mel.weights}
        fm = [v for k, v in consts.items() if k.split(":")[0].rsplit("/", 1)[-1].startswith("Variable")][0]
        rk = [v for k, v in consts.items() if "real_kernels" in k][0]
        ik = [v for k, v in consts.items() if "imag_kernels" in k][0]
        save("tf_mel_L%d.npz" % L, wave_seed=5, L=L, mel=y[..., 0], freq2mel=fm,
             real_kernels_bins=rk.reshape(1024, -1)[:, [0, 1, 37, 256, 512]], imag_kernels_bins=ik.reshape(1024, -1)[:, [0, 1, 37, 256, 512]],
             **{"meta_" + k: v for k, v in meta.items()})
    done.append("mel")

    # ---- a5: ConvSubsampling (conformer_blocks.py:67-96) -----------------------------------------------------------
    cfg = dict(co.CONFORMER_S, num_blocks=2)
    w = co.encoder_weights(cfg, seed=0)
    sub = cb.ConvSubsampling(odim=144, reduction_factor=4, dropout=0.0)
    rng = np.random.default_rng(200)
    melin = (-80.0 * rng.random((3, 200, 80, 1))).astype(np.float32)
    sub(tf.constant(melin), training=False)
    full = lambda n: n if "conv_subsampling" in n else "conv_subsampling/" + n      # noqa: E731 (scope of a stand-alone layer)
    names = {full(v.name): v for v in sub.weights}
    m = checkpoint.keras_names_to_abi(list(names))
    if len(m) != len(names):
        raise RuntimeError("ConvSubsampling variables not mapped: %s" % [n for n in names if n not in m])
    for n, v in names.items():
        v.assign(w[m[n]].reshape(v.shape))
    save("tf_conv_subsampling.npz", mel_seed=200, out=sub(tf.constant(melin), training=False).numpy(), weights_seed=0)
    done.append("conv_subsampling")

    # ---- a10, a12, a13: ConformerEncoder (2 blocks) + CTCDecoder + ctc_decode --------------------------------------
    enc = cb.ConformerEncoder(dmodel=144, reduction_factor=4, num_blocks=2, head_size=36, num_heads=4, kernel_size=32,
                              fc_factor=0.5, dropout=0.0, add_wav_info=False, sample_rate=16000, n_mels=80,
                              mel_layer_type="Melspectrogram", mel_layer_trainable=False, stride_ms=10)
    enc._build()
    kept = assign_by_name(enc, w)
    V = 50
    wc = co.ctc_decoder_weights(cfg, V, seed=1)
    ctc = cb.CTCDecoder(num_classes=V, dmodel=144, num_blocks=1, head_size=36, num_heads=4, kernel_size=32,
                        dropout=0.0, fc_factor=0.5)
    ctc._build()
    assign_by_name(ctc, wc)
    x = waves(2, 32000)
    e = enc(tf.constant(x[..., None]), training=False)
    lg = ctc(e, training=False)
    dec = tf.keras.backend.ctc_decode(tf.nn.softmax(lg, -1), np.array([e.shape[1]] * 2, "int32"))[0][0].numpy()
    save("tf_encoder_ctc.npz", wave_seed=0, L=32000, enc=e.numpy(), logits=lg.numpy(), ctc_decode=dec, num_classes=V,
         enc_weights_seed=0, ctc_weights_seed=1, freq2mel=kept.get("mel_layer/freq2mel"))
    done.append("encoder_ctc")

    # ---- a11: StreamingConformerEncoder (Streaming_ConformerS.yml dims, 2 blocks) ---------------------------------
    scfg = dict(co.STREAMING_S, num_blocks=2)
    ws = co.encoder_weights(scfg, seed=2)
    senc = cb.StreamingConformerEncoder(dmodel=256, reduction_factor=4, num_blocks=2, head_size=64, num_heads=4,
                                        kernel_size=5, fc_factor=0.5, dropout=0.0, add_wav_info=False, sample_rate=16000,
                                        n_mels=80, mel_layer_type="Melspectrogram", mel_layer_trainable=False, stride_ms=10)
    senc.add_chunk_size(8000, 80, 640)
    senc._build()
    kept_s = assign_by_name(senc, ws)
    xs = waves(2, 24000, 9)
    save("tf_streaming_encoder.npz", wave_seed=9, L=24000, enc=senc(tf.constant(xs[..., None]), training=False).numpy(),
         weights_seed=2, freq2mel=kept_s.get("mel_layer/freq2mel"))
    done.append("streaming_encoder")

    # ---- 8f-1: Translator ------------------------------------------------------------------------------------------
    tcfg = dict(co.CONFORMER_S, translator_num_blocks=2, translator_kernel_size=32, translator_fc_factor=0.5)
    wt = co.translator_weights(tcfg, 60, 80, seed=13)
    tr = cb.Translator(inp_classes=60, tar_classes=80, dmodel=144, num_blocks=2, head_size=36, num_heads=4,
                       fc_factor=0.5, dropout=0.0, kernel_size=32)
    tr._build()
    assign_by_name(tr, wt)
    rng = np.random.default_rng(3040)
    ids = rng.integers(0, 60, (3, 40)).astype(np.int32)
    encin = rng.standard_normal((3, 250, 144)).astype(np.float32)
    save("tf_translator.npz", seed=3040, logits=tr([tf.constant(ids), tf.constant(encin)], training=False).numpy(),
         weights_seed=13)
    done.append("translator")

    # ---- a1-a4, a10, a12, a13 on the reference's OWN speech recordings (round 6) ----------------------------------------------
    # asr/BAC009S0764W0121.wav is the file test_asr.py:272-275 transcribes, Inference/CppInference/onnx/test.wav the C++ demo's.
    # Driven as test_asr.py:186-200 drives them: SpeechFeaturizer.load_wav (utils/speech_featurizers.py:10-22,68-70) ->
    # reshape [1, -1, 1] (no peak normalisation: the recordings peak at 0.05 / 0.3) -> encoder -> CTCDecoder with the reference's
    # trained weights (the exported ctc_model.onnx) -> softmax -> ctc_decode.  Real speech has what the synthetic waves lack:
    # silences at the -80 dB floor, int16 quantisation, a 30 dB dynamic range inside one utterance.  The fixture carries the PCM
    # samples (data the reference's demo holds) so that the GPU box, which has no reference checkout, runs the same input.
    from utils.speech_featurizers import SpeechFeaturizer
    import wave as _wave
    sf_ = SpeechFeaturizer({"sample_rate": 16000})
    wtr = dict(np.load(os.path.join(HERE, "ctc_decoder_weights.npz")))
    ctc_t = cb.CTCDecoder(num_classes=1332, dmodel=144, num_blocks=1, head_size=36, num_heads=4, kernel_size=32, dropout=0.0, fc_factor=0.5)
    ctc_t._build()
    assign_by_name(ctc_t, wtr)
    mel_l = Melspectrogram(sr=16000, n_mels=80, n_hop=160, n_dft=1024, trainable_fb=False)
    sp = {}
    for tag, rel in (("bac", "asr/BAC009S0764W0121.wav"), ("cpp", "Inference/CppInference/onnx/test.wav")):
        path = os.path.join(REF, rel)
        data = sf_.load_wav(path)
        with _wave.open(path, "rb") as f_:
            pcm = np.frombuffer(f_.readframes(f_.getnframes()), "<i2").copy()
        assert data.dtype == np.float32 and np.array_equal(data, pcm.astype(np.float32) / np.float32(32768.0))
        xin = tf.constant(data.reshape([1, -1, 1]))
        e1 = enc(xin, training=False)
        lg1 = ctc_t(e1, training=False).numpy()
        # the trained head answers `blank` to a random-weight encoder (its decode is empty): the seeded 50-class CTCDecoder of the
        # section above is run as well, so that the greedy decode of real speech has tokens to collapse
        lg2 = ctc(e1, training=False)
        n1 = np.array([e1.shape[1]], "int32")
        dec1 = tf.keras.backend.ctc_decode(tf.nn.softmax(tf.constant(lg1), -1), n1)[0][0].numpy()
        dec2 = tf.keras.backend.ctc_decode(tf.nn.softmax(lg2, -1), n1)[0][0].numpy()
        order = np.argsort(-lg1, axis=-1, kind="stable")[..., :4]
        sp.update({tag + "_pcm": pcm, tag + "_mel": mel_l(xin).numpy()[..., 0], tag + "_enc": e1.numpy(),
                   tag + "_trained_logits_every4": lg1[:, ::4], tag + "_trained_top4_idx": order.astype(np.int16),
                   tag + "_trained_top4_val": np.take_along_axis(lg1, order, -1), tag + "_trained_ctc_decode": dec1,
                   tag + "_logits50": lg2.numpy(), tag + "_ctc_decode50": dec2, tag + "_path": np.asarray(rel)})
    save("tf_speech.npz", enc_weights_seed=0, ctc50_weights_seed=1, freq2mel=kept.get("mel_layer/freq2mel"), **sp)
    done.append("speech")

    # ---- 8f-4: LEAF frontend and the WavePickModel branch ----------------------------------------------------------
    try:
        from leaf_audio import frontend
        leaf = frontend.Leaf(n_filters=80, sample_rate=16000, window_stride=10,
                             complex_conv_init=frontend.initializers.GaborInit(sample_rate=16000, min_freq=60, max_freq=7800))
        xl = waves(2, 16000, 60)
        leaf(tf.constant(xl[..., None]), training=False)
        for v in leaf.weights:             # the fixture stores float32 values: run on exactly those (matters for the stand-in's wide pass)
            v.assign(np.asarray(v.numpy(), np.float32))
        yl = leaf(tf.constant(xl[..., None]), training=False).numpy()
        save("tf_leaf.npz", wave_seed=60, L=16000, out=yl, **{v.name.replace("/", "|"): v.numpy() for v in leaf.weights})
        done.append("leaf")
    except Exception as e:      # tensorflow_addons / leaf dependencies may be missing
        print("LEAF skipped:", repr(e))
    from asr.models.wav_model import WavePickModel
    wp = WavePickModel(144, 640)
    xw = waves(2, 16000, 90)
    wp(tf.constant(xw[..., None]), training=False)
    ww = co.wave_pick_weights(144, 640, seed=22)
    names = {(v.name if "wave_pick_model" in v.name else "wave_pick_model/" + v.name): v for v in wp.weights}
    m = checkpoint.keras_names_to_abi(list(names))
    if len(m) != len(names):
        raise RuntimeError("WavePickModel variables not mapped: %s" % [n for n in names if n not in m][:8])
    for n, v in names.items():
        v.assign(ww[m[n]].reshape(v.shape))
    save("tf_wave_pick.npz", wave_seed=90, L=16000, out=wp(tf.constant(xw[..., None]), training=False).numpy(), weights_seed=22)
    done.append("wave_pick")

    # ---- a15: ChunkConformer.predict (chunk_conformerS.yml dims, 2 encoder blocks, small vocabularies) -------------
    # 6 s utterances: T = 150 frames, so that the band mask (win_front 36) cuts; the picker's blank bias is set so that about
    # half of the frames are kept -- feature_pick's tf.while_loop compacts ragged counts and zero-pads to the batch maximum
    from asr.models.chunk_conformer_blocks import ChunkConformer
    ccfg = dict(co.CHUNK_S, enc_num_blocks=2, picker_num_classes=30, decoder_num_classes=40)
    wch = co.chunk_weights(ccfg, seed=3)
    xc = waves(2, 96000, 40)
    wch["picker/fully_connected/bias"][-1] = np.float32(pick_bias_for_ragged_counts(ccfg, wch, xc))
    model = ChunkConformer(chunk_config_dict(ccfg), 30, 40)
    kept = assign_by_object_path(model, wch, tf)
    front = model.front(tf.constant(xc[..., None]), training=False)
    encc = model.encoder(front, training=False)
    phone, hidden = model.phone_picker(encc, training=False)
    picked_f, picked_c = model.feature_pick(hidden, phone)
    save("tf_chunk_predict.npz", wave_seed=40, L=96000, front=front.numpy(), enc=encc.numpy(), picker_logits=phone.numpy(),
         picker_hidden=hidden.numpy(), picked=picked_f.numpy(), text_logits=model.predict(tf.constant(xc[..., None])).numpy(), weights_seed=3,
         picker_blank_bias=wch["picker/fully_connected/bias"][-1], freq2mel=kept.get("front/mel_layer/freq2mel"))
    done.append("chunk_predict")

    # ---- a15 streaming: picker_stream_predict / feature_pick / decoder_stream_predict fed 2560 samples at a time with explicit
    # caches, driven as test_chunk_asr.py:60-83 drives them (chunk_conformer_blocks.py:72-91, 209-229, 297-316, 449-456, 522-560,
    # 646-673, 824-852): the valid outputs of every step, which steps produced text, the look-ahead rows, the final caches
    xs1 = xc[:1, :2560 * 30]
    pc, dc = model.init_picker_caches(1), model.init_decoder_caches(1)
    ph, hid, txt, unv, steps = [], [], [], None, []
    for i in range(30):
        vp, _, vh, pc = model.picker_stream_predict(tf.constant(xs1[:, i * 2560:(i + 1) * 2560, None]), pc)
        if vp.shape[1] == 0:
            continue
        ph.append(vp.numpy()), hid.append(vh.numpy())
        f, _ = model.feature_pick(vh, vp)
        if f.shape[1] != 0:
            vt, unv, dc = model.decoder_stream_predict(f, dc)
            txt.append(vt.numpy())
            steps.append((i, int(vt.shape[1])))
    save("tf_chunk_stream.npz", wave_seed=40, samples=2560, nchunks=30, weights_seed=3, picker_blank_bias=wch["picker/fully_connected/bias"][-1],
         picker_logits=np.concatenate(ph, 1), picker_hidden=np.concatenate(hid, 1), text_logits=np.concatenate(txt, 1),
         unvalid_text_logits=unv.numpy(), steps=np.asarray(steps, np.int32),
         cache_front_wav=pc[0].numpy(), cache_front_sub=pc[1].numpy(), cache_enc_mha=pc[2].numpy(), cache_enc_cnn=pc[3].numpy(),
         cache_picker_mha=pc[4].numpy(), cache_picker_cnn=pc[5].numpy(), cache_picker_dec_inp=pc[6].numpy(),
         cache_helper_mha=dc[0].numpy(), cache_decoder_mha=dc[2].numpy(), cache_decoder_cnn=dc[3].numpy(), cache_decoder_dec_inp=dc[4].numpy(),
         freq2mel=kept.get("front/mel_layer/freq2mel"))
    done.append("chunk_stream")
    print("done:", done)


if __name__ == "__main__":
    import pickle
    import subprocess
    if "--wide-pass" in sys.argv:
        # the same reference code carried in float64 (float32-rounded constants).  Its own process: leaf_audio/frontend.py:101
        # builds the PCEN layer as a default ARGUMENT, i.e. once per process, and a second Leaf would inherit the first one's
        # float32 variables.
        tf_, standin_ = import_tensorflow()
        assert standin_
        tf_.set_wide(True)
        SUFFIX[0] = "_f64"
        main()
        with open(sys.argv[sys.argv.index("--wide-pass") + 1], "wb") as f:
            pickle.dump(FIXTURES, f)
        sys.exit(0)
    main()
    if import_tensorflow()[1]:
        with tempfile.TemporaryDirectory() as td:
            tmp = os.path.join(td, "wide.pkl")
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--wide-pass", tmp])
            with open(tmp, "rb") as f:
                for name, fx in pickle.load(f).items():
                    FIXTURES[name].update(fx)
    write_all()
