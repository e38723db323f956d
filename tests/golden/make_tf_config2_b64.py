"""The benched batch (BASELINE config 2: ConformerCTC(S), 64 x 10 s) through the REFERENCE'S OWN model code: the reference's
`ConformerEncoder` (13 blocks) + `CTCDecoder` + `tf.keras.backend.ctc_decode`, imported unmodified from /root/reference and
executed on the NumPy stand-in for TensorFlow (oracle/_tfshim; a real TensorFlow is used when there is one), on bench.py's
inputs `synth_batch(0, 64, 160000)` with the two weight sets of tests/golden/make_config2_b64.py:

  head "trained": bench.py's model -- Keras-default encoder (seed 0) + the reference's exported CTCDecoder weights
  head "tokens":  oracle encoder weights seed 0 + CTCDecoder seed 1 with the class bias stored in config2_oracle_b64.npz
                  (15 411 tokens over the batch)

    python tests/golden/make_tf_config2_b64.py        (about 5 minutes on 8 cores; writes tests/golden/tf_config2_b64.npz)

Same keys as config2_oracle_b64.npz (`<head>_enc_every10`, `_top4_idx`, `_top4_val`, `_logits_every50`, `_ids`, `_lens`) from
the float32 run, plus `<head>_top4_val_f64` / `<head>_top4_idx_f64` from the run that carries float32 tensors in float64.
tests/test_tf_goldens.py checks the oracle fixture against it (CPU: ids of all 64 utterances, the top-4 logits of every frame)
and re-runs two utterances of this recipe; tests/test_gpu_baseline_shapes.py holds libmi355asr.so to it at the benched shape.
"""
import os
import pickle
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests"), HERE, os.environ.get("REFERENCE_ROOT", "/root/reference")):
    if p not in sys.path:
        sys.path.insert(0, p)
OUT = os.path.join(os.environ.get("MI355ASR_TF_GOLDEN_OUT", HERE), "tf_config2_b64.npz")
B, L, V = 64, 160000, 1332


def weights_of(head):
    import make_config2_b64 as c2
    if head == "trained":
        return c2.bench_weights()
    from helpers import co
    cfg = dict(co.CONFORMER_S)
    w = co.encoder_weights(cfg, seed=0)
    w.update(co.ctc_decoder_weights(cfg, V, seed=1))
    w["fully_connected/bias"] = np.load(os.path.join(HERE, "config2_oracle_b64.npz"))["tokens_fc_bias"]
    return w


def run(heads=("trained", "tokens"), utterances=None, chunk=8):
    """-> {key: array} of one precision (the stand-in's current one) for the given utterances (default: all 64)"""
    import make_tf_goldens as g
    tf, _ = g.import_tensorflow()
    from asr.models import conformer_blocks as cb
    from tensorflowasr_amd.synthetic import synth_batch
    utt = list(range(B)) if utterances is None else list(utterances)
    x = synth_batch(0, B, L)[utt]
    out = {}
    for head in heads:
        w = weights_of(head)
        # test_asr.py:26-75: encoder, then the CTC decoder
        enc = cb.ConformerEncoder(dmodel=144, reduction_factor=4, num_blocks=13, head_size=36, num_heads=4, kernel_size=32, fc_factor=0.5,
                                  dropout=0.0, add_wav_info=False, sample_rate=16000, n_mels=80, mel_layer_type="Melspectrogram",
                                  mel_layer_trainable=False, stride_ms=10)
        enc._build()
        ctc = cb.CTCDecoder(num_classes=V, dmodel=144, num_blocks=1, head_size=36, num_heads=4, kernel_size=32, dropout=0.0, fc_factor=0.5)
        ctc._build()
        g.assign_by_name(enc, w, keep=())                 # the DFT kernels / mel matrix too: exactly the benched model's tensors
        g.assign_by_name(ctc, w)
        parts = {k: [] for k in ("enc_every10", "top4_idx", "top4_val", "logits_every50", "ids", "lens")}
        for i in range(0, len(utt), chunk):
            xb = x[i:i + chunk]
            e = enc(tf.constant(xb[..., None]), training=False)
            lg = ctc(e, training=False)
            dec = tf.keras.backend.ctc_decode(tf.nn.softmax(lg, -1), np.array([e.shape[1]] * len(xb), "int32"))[0][0].numpy()
            e, lg = e.numpy(), lg.numpy()
            order = np.argsort(-lg, axis=-1, kind="stable")[..., :4]
            parts["enc_every10"].append(e[:, ::10])
            parts["top4_idx"].append(order.astype(np.int16))
            parts["top4_val"].append(np.take_along_axis(lg, order, -1))
            parts["logits_every50"].append(lg[:, ::50])
            ids = -np.ones((len(xb), 250), np.int32)
            ids[:, :dec.shape[1]] = dec
            parts["ids"].append(ids)
            parts["lens"].append((dec >= 0).sum(1).astype(np.int32))
            print(head, i + len(xb), "/", len(utt), flush=True)
        for k, v in parts.items():
            out[head + "_" + k] = np.concatenate(v)
    return out


if __name__ == "__main__":
    import make_tf_goldens as g
    if "--wide-pass" in sys.argv:
        tf_, standin_ = g.import_tensorflow()
        assert standin_
        tf_.set_wide(True)
        res = run()
        with open(sys.argv[sys.argv.index("--wide-pass") + 1], "wb") as f:
            pickle.dump({k: v for k, v in res.items() if k.endswith(("top4_val", "top4_idx"))}, f)
        sys.exit(0)
    res = {k: (v.astype(np.float32) if v.dtype == np.float64 else v) for k, v in run().items()}
    if g.import_tensorflow()[1]:
        with tempfile.TemporaryDirectory() as td:
            tmp = os.path.join(td, "wide.pkl")
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--wide-pass", tmp])
            with open(tmp, "rb") as f:
                for k, v in pickle.load(f).items():
                    res[k + "_f64"] = v
    tf_mod, standin = g.import_tensorflow()
    res["meta_generator"] = np.asarray("%s %s" % ("numpy stand-in (oracle/_tfshim)" if standin else "tensorflow", tf_mod.__version__))
    np.savez_compressed(OUT, **res)
    print("wrote %s (%d KB): %s" % (OUT, os.path.getsize(OUT) // 1024, sorted(res)))
