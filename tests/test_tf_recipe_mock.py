"""tests/golden/make_tf_goldens.py cannot run in this image (no TensorFlow / librosa), so nothing would notice if the weight
names it maps drifted -- the f1 / f4 names moved twice before (round-3 verdict, item 10).  This test runs the WHOLE recipe
against stand-ins: a `tensorflow` module that is NumPy underneath and "reference" model classes whose `.weights` carry the
variable names Keras would print for the reference's layers (auto-numbered per kind, in construction order) with the shapes of
this repository's CURRENT handles, and whose calls return arrays of the right rank.  Every `assign` checks name -> tensor ->
shape; the recipe's own coverage checks ("variables without a seeded tensor", "assigned n of m ChunkConformer tensors") must
pass for every component, and every fixture it would write must carry the keys tests/test_tf_goldens.py reads.  The day
someone has a TensorFlow box, the script does not fail on name drift."""
import importlib.util
import os
import re
import sys
import types

import numpy as np
import pytest

from helpers import ROOT, co
from test_host import _keras_style_names


class _T:
    """a tensor that is an array"""
    def __init__(self, a):
        self.a = np.asarray(a)
        self.shape = self.a.shape

    def numpy(self):
        return self.a


class _Var:
    def __init__(self, name, shape):
        self.name, self.shape, self.value, self.assigned = name, tuple(shape), np.zeros(shape, np.float32), False

    def assign(self, v):
        v = np.asarray(v)
        assert v.shape == self.shape, (self.name, v.shape, self.shape)
        self.value, self.assigned = v, True

    def numpy(self):
        return self.value


class _Model:
    """weights named the Keras way for the ABI names of `handle_model`; call -> zeros of `out_shape(inputs)`"""
    def __init__(self, scope, named_shapes, out_shape, start=7, stand_alone=False):
        k2a = _keras_style_names(scope, [n for n, _ in named_shapes], start=start)
        shapes = dict(named_shapes)
        self.weights = []
        for k, abi in k2a.items():
            name = k.split("/", 1)[1] if stand_alone else k          # a stand-alone layer prints no model scope
            if abi.startswith("mel_layer/"):
                leaf = abi.split("/")[1]
                name = scope + "/mel_layer/" + ("Variable:0" if leaf == "freq2mel" else leaf + ":0")
            self.weights.append(_Var(name, shapes[abi]))
        self._out = out_shape

    def _build(self):
        return self

    def add_chunk_size(self, *a):
        return None

    def __call__(self, x, training=False):
        return _T(np.zeros(self._out(x), np.float32))

    def all_assigned(self, skip=("real_kernels", "imag_kernels", "Variable")):
        return [v.name for v in self.weights if not v.assigned and not any(s in v.name for s in skip)]


def _fake_tf():
    tf = types.ModuleType("tensorflow")
    tf.__version__ = "mock"
    tf.constant = lambda x: np.asarray(x)
    tf.nn = types.SimpleNamespace(softmax=lambda x, axis=-1: _T(np.zeros(np.asarray(getattr(x, "a", x)).shape, np.float32)))
    tf.keras = types.SimpleNamespace(backend=types.SimpleNamespace(
        ctc_decode=lambda p, lens: [[_T(np.zeros((len(lens), 3), np.int64))]]))
    tf.train = types.SimpleNamespace(list_variables=lambda prefix: tf._chunk_keys)
    return tf


def _shape_of(x):
    return np.asarray(getattr(x, "a", x)).shape


def test_tf_golden_recipe_resolves_every_variable_against_the_current_weight_names(monkeypatch, tmp_path):
    from tensorflowasr_amd.models import ChunkConformer, ConformerEncoder, CTCDecoder, StreamingConformerEncoder, Translator
    tf = _fake_tf()
    made = {}

    def names_of(model, drop_mel=False):
        return [(n, s) for n, s in model._names_and_shapes() if not (drop_mel and n.startswith("mel_layer/"))]

    # ---- the reference's classes, as far as the recipe touches them --------------------------------------------------
    def ConvSubsampling(odim, reduction_factor, dropout):
        e = ConformerEncoder(dmodel=odim, num_blocks=1, mel_layer_type="Melspectrogram")
        m = _Model("conformer_encoder", [(n, s) for n, s in names_of(e) if n.startswith("conv_subsampling/")],
                   lambda x: (_shape_of(x)[0], -(-_shape_of(x)[1] // 4), odim), stand_alone=True)
        made["sub"] = m
        return m

    def Encoder(cls, key, scope):
        def make(**kw):
            kw = {k: v for k, v in kw.items() if k not in ("dropout", "mel_layer_trainable")}
            e = cls(**kw)
            m = _Model(scope, names_of(e), lambda x: (_shape_of(x)[0], -(-_shape_of(x)[1] // 640), kw["dmodel"]))
            made[key] = m
            return m
        return make

    def CTC(num_classes, **kw):
        kw.pop("dropout", None)
        d = CTCDecoder(num_classes=num_classes, **kw)
        m = _Model("ctc_decoder", names_of(d), lambda x: _shape_of(x)[:2] + (num_classes,), start=53)
        made["ctc"] = m
        return m

    def Trans(inp_classes, tar_classes, **kw):
        kw.pop("dropout", None)
        t = Translator(inp_classes=inp_classes, tar_classes=tar_classes, **kw)
        m = _Model("translator", [(n, s) for n, s in zip(t._h.weight_names(), [dict(t._names_and_shapes())[n] for n in t._h.weight_names()])],
                   lambda x: _shape_of(x[0]) + (tar_classes,), start=2)
        made["translator"] = m
        return m

    def Mel(**kw):
        e = ConformerEncoder(dmodel=144, num_blocks=1, mel_layer_type="Melspectrogram")
        m = _Model("melspectrogram", [(n, s) for n, s in names_of(e) if n.startswith("mel_layer/")],
                   lambda x: (_shape_of(x)[0], -(-_shape_of(x)[1] // 160), 80, 1))
        for v in m.weights:          # stand-alone layer: "melspectrogram/real_kernels:0", the filterbank an unnamed Variable
            v.value = np.zeros(v.shape, np.float32)
        return m

    def WavePick(d, hop):
        e = ConformerEncoder(dmodel=d, num_blocks=1, add_wav_info=True, mel_layer_type="Melspectrogram")
        m = _Model("conformer_encoder", [(n, s) for n, s in names_of(e) if n.startswith("wav_layer/")],
                   lambda x: (_shape_of(x)[0], _shape_of(x)[1] // hop, d), stand_alone=True)
        made["wave_pick"] = m
        return m

    class Chunk:
        """object graph addressed by attribute paths (chunk_conformer_blocks.py): built from the checkpoint keys"""
        def __init__(self, cfg, phone, txt):
            mine = ChunkConformer(cfg, phone, txt)
            shapes = dict(mine._names_and_shapes())
            from test_host import test_chunk_checkpoint_keys_map_onto_every_chunk_tensor as _t  # noqa: F401 (same key construction below)
            inv_root = {"front": "front", "encoder": "encoder", "picker": "phone_picker", "decoder": "decoder", "helper": "helper"}
            inv_blk = {"ff_module_1": "ffm1", "ff_module_2": "ffm2", "mhsa_module": "mhsam", "conv_module": "convm"}
            inv_mha = {"query": "_query_dense", "key": "_key_dense", "value": "_value_dense", "attention_output": "_output_dense"}
            inv_mel = {"real_kernels": "dft_real_kernels", "imag_kernels": "dft_imag_kernels", "freq2mel": "freq2mel"}
            self.vars, keys = {}, []
            for n in sorted(shapes):
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
                keys.append(k + "/.ATTRIBUTES/VARIABLE_VALUE")
                self._plant(k.split("/"), _Var(k, shapes[n]))
            tf._chunk_keys = [(k, None) for k in keys] + [("_CHECKPOINTABLE_OBJECT_GRAPH", None), ("optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE", None)]
            self._d, self._phone, self._txt = cfg["model_config"]["ChunkConformerEncoder"]["dmodel"], phone, txt
            made["chunk"] = self

        def _plant(self, parts, var):
            node = self.__dict__.setdefault("_tree", {})
            for q in parts[:-1]:
                node = node.setdefault(q, {})
            node[parts[-1]] = var
            self.vars["/".join(parts)] = var

        def __getattr__(self, name):
            tree = self.__dict__.get("_tree", {})
            if name in tree:
                return _Node(tree[name])
            raise AttributeError(name)

        def save_weights(self, prefix):
            return None

        def predict(self, x):
            return _T(np.zeros((_shape_of(x)[0], 5, self._txt), np.float32))

    class _Node:
        def __init__(self, tree):
            self._t = tree

        def __getattr__(self, name):
            v = self._t[name]
            return v if isinstance(v, _Var) else _Node(v)

        def __getitem__(self, i):
            v = self._t[str(i)]
            return v if isinstance(v, _Var) else _Node(v)

        def __call__(self, x, training=False):                 # front / encoder / phone_picker stages
            B = _shape_of(x)[0]
            if "fc" in self._t:                                # the phone picker returns (logits, hidden)
                return _T(np.zeros((B, 9, 30), np.float32)), _T(np.zeros((B, 9, 144), np.float32))
            return _T(np.zeros((B, 9, 144), np.float32))

    cbm = types.ModuleType("asr.models.conformer_blocks")
    cbm.ConvSubsampling = ConvSubsampling
    cbm.ConformerEncoder = Encoder(ConformerEncoder, "encoder", "conformer_encoder")
    cbm.StreamingConformerEncoder = Encoder(StreamingConformerEncoder, "streaming", "stream_conformer_encoder")
    cbm.CTCDecoder = CTC
    cbm.Translator = Trans
    tfm = types.ModuleType("asr.models.layers.time_frequency")
    tfm.Melspectrogram = Mel
    wvm = types.ModuleType("asr.models.wav_model")
    wvm.WavePickModel = WavePick
    chm = types.ModuleType("asr.models.chunk_conformer_blocks")
    chm.ChunkConformer = Chunk
    pk = lambda n: types.ModuleType(n)          # noqa: E731
    mods = {"tensorflow": tf, "asr": pk("asr"), "asr.models": pk("asr.models"), "asr.models.layers": pk("asr.models.layers"),
            "asr.models.conformer_blocks": cbm, "asr.models.layers.time_frequency": tfm, "asr.models.wav_model": wvm,
            "asr.models.chunk_conformer_blocks": chm}
    mods["asr.models"].conformer_blocks = cbm
    for k, v in mods.items():
        monkeypatch.setitem(sys.modules, k, v)
    monkeypatch.setitem(sys.modules, "leaf_audio", None)        # the recipe treats LEAF as optional (tensorflow_addons)
    monkeypatch.setitem(sys.modules, "librosa", None)

    spec = importlib.util.spec_from_file_location("make_tf_goldens_mocked", os.path.join(ROOT, "tests", "golden", "make_tf_goldens.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    written = {}
    monkeypatch.setattr(g, "save", lambda name, **arrays: written.__setitem__(name, set(arrays)))
    g.main()                                                    # raises on any unmapped / unassigned variable

    # every component ran, every variable of every stand-in model received a tensor of its shape
    assert set(written) >= {"tf_mel_L32000.npz", "tf_mel_L67263.npz", "tf_conv_subsampling.npz", "tf_encoder_ctc.npz",
                            "tf_streaming_encoder.npz", "tf_translator.npz", "tf_wave_pick.npz", "tf_chunk_predict.npz"}
    for key in ("sub", "encoder", "streaming", "ctc", "translator", "wave_pick"):
        assert made[key].all_assigned() == [], (key, made[key].all_assigned()[:5])
    ch = made["chunk"]
    assert [k for k, v in ch.vars.items() if not v.assigned and "mel_layer" not in k] == []
    # ... and the fixtures would hold what tests/test_tf_goldens.py reads
    assert {"mel", "freq2mel", "real_kernels_bins", "imag_kernels_bins", "wave_seed"} <= written["tf_mel_L32000.npz"]
    assert {"enc", "logits", "ctc_decode", "num_classes", "enc_weights_seed", "ctc_weights_seed", "wave_seed", "L"} <= written["tf_encoder_ctc.npz"]
    assert {"front", "enc", "picker_logits", "picker_hidden", "text_logits", "weights_seed"} <= written["tf_chunk_predict.npz"]
