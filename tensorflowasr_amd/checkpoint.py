"""Checkpoint readers that funnel into the C-ABI weight names.

* `.npz` of Keras-layout tensors: `Model.load_weights(path)` reads it directly (models.py).
* tf2onnx exports of the reference's models (`test_asr.py:232-242` writes them; the repository ships
  `Inference/PythonInference/asr/models/offline/ctc_model.onnx`): `ctc_decoder_weights_from_onnx` below recovers the
  CTCDecoder's Keras tensors from the graph.  tf2onnx keeps the TF scope names on most nodes but const-folds some
  weights into anonymous initializers and lowers the attention einsums to Gemm/MatMul, so the tensors are located
  *structurally* (node-name suffixes inside each block scope, and for the attention projections by following the data
  flow from the `truediv` / `Softmax` nodes), not by the export's initializer numbering.
* TensorFlow checkpoints (tensor bundle: `.index` + `.data-*`; what `save_weights(prefix)` writes and the ChunkConformer
  trainer uses): `tfbundle.py` (sorted-string-table index, CRC32C-checked tensors, object graph -> variable names);
  `tf_checkpoint_to_abi(path)`.
* Keras `.h5` weight files: read with the pure-Python HDF5 reader `h5lite.py` (h5py is not available next to the system
  interpreter of this image), variable names mapped by `keras_names_to_abi`; `keras_h5_to_abi(path)`.

No onnx / protobuf package is used: the ONNX file is a protobuf whose few fields of interest are decoded here
(onnx.proto3: ModelProto.graph = 7; GraphProto.node = 1, initializer = 5; NodeProto.input = 1, output = 2, name = 3,
op_type = 4, attribute = 5; AttributeProto.name = 1, i = 3, ints = 8; TensorProto.dims = 1, data_type = 2,
float_data = 4, int32_data = 5, int64_data = 7, name = 8, raw_data = 9)."""
import os
import re
import struct

import numpy as np


# ---------------------------------------------------------------------------------------------------------
# protobuf wire format
# ---------------------------------------------------------------------------------------------------------
def _varint(b, p):
    v = s = 0
    while True:
        c = b[p]
        p += 1
        v |= (c & 0x7F) << s
        if c < 0x80:
            return v, p
        s += 7


def _msg(b):
    """(field, wire type, value) triples of one message; length-delimited values are memoryview slices."""
    p, n = 0, len(b)
    while p < n:
        key, p = _varint(b, p)
        f, t = key >> 3, key & 7
        if t == 0:
            v, p = _varint(b, p)
        elif t == 1:
            v, p = bytes(b[p:p + 8]), p + 8
        elif t == 2:
            ln, p = _varint(b, p)
            v, p = b[p:p + ln], p + ln
        elif t == 5:
            v, p = bytes(b[p:p + 4]), p + 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % t)
        yield f, t, v


def _packed_varints(v):
    out, p = [], 0
    while p < len(v):
        x, p = _varint(v, p)
        out.append(x)
    return out


_DTYPES = {1: np.float32, 6: np.int32, 7: np.int64, 11: np.float64, 9: np.bool_, 10: np.float16}


def _tensor(b):
    dims, dtype, name, raw = [], 1, "", None
    fdata, idata = [], []
    for f, t, v in _msg(b):
        if f == 1:
            dims += _packed_varints(v) if t == 2 else [v]
        elif f == 2:
            dtype = v
        elif f == 8:
            name = bytes(v).decode()
        elif f == 9:
            raw = bytes(v)
        elif f == 4:
            fdata += list(struct.unpack("<%df" % (len(v) // 4), bytes(v))) if t == 2 else [struct.unpack("<f", v)[0]]
        elif f in (5, 7):
            idata += _packed_varints(v) if t == 2 else [v]
    if dtype not in _DTYPES:
        return name, None
    dt = _DTYPES[dtype]
    if raw is not None:
        arr = np.frombuffer(raw, dtype=dt).copy()
    elif fdata:
        arr = np.asarray(fdata, dt)
    else:
        arr = np.asarray([x - (1 << 64) if x >= (1 << 63) else x for x in idata], dt)
    return name, arr.reshape(dims)


class OnnxNode:
    __slots__ = ("op", "name", "inputs", "outputs", "ints")

    def __init__(self):
        self.op, self.name, self.inputs, self.outputs, self.ints = "", "", [], [], {}


def read_onnx(path):
    """-> (nodes in file order, {initializer name: ndarray})."""
    with open(path, "rb") as fh:
        buf = memoryview(fh.read())
    graph = None
    for f, t, v in _msg(buf):
        if f == 7:
            graph = v
    if graph is None:
        raise ValueError("%s: no GraphProto" % path)
    nodes, inits = [], {}
    for f, t, v in _msg(graph):
        if f == 1:
            n = OnnxNode()
            for g, u, w in _msg(v):
                if g == 1:
                    n.inputs.append(bytes(w).decode())
                elif g == 2:
                    n.outputs.append(bytes(w).decode())
                elif g == 3:
                    n.name = bytes(w).decode()
                elif g == 4:
                    n.op = bytes(w).decode()
                elif g == 5:
                    an, ai, ais = "", None, []
                    for h, x, y in _msg(w):
                        if h == 1:
                            an = bytes(y).decode()
                        elif h == 3:
                            ai = y
                        elif h == 8:
                            ais += _packed_varints(y) if x == 2 else [y]
                    n.ints[an] = ais if ais else ai
            nodes.append(n)
        elif f == 5:
            name, arr = _tensor(v)
            if arr is not None:
                inits[name] = arr
    return nodes, inits


# ---------------------------------------------------------------------------------------------------------
# CTCDecoder (conformer_blocks.py:385-438) from a tf2onnx graph
# ---------------------------------------------------------------------------------------------------------
_PASS = ("Reshape", "Squeeze", "Unsqueeze", "Transpose", "Identity", "Cast")


class _Graph:
    def __init__(self, nodes, inits):
        self.nodes, self.inits = nodes, inits
        self.producer = {o: n for n in nodes for o in n.outputs}
        self.consumers = {}
        for n in nodes:
            for i in n.inputs:
                self.consumers.setdefault(i, []).append(n)

    def node(self, suffix, op=None):
        hits = [n for n in self.nodes if n.name.endswith(suffix) and (op is None or n.op == op)]
        if len(hits) != 1:
            raise ValueError("expected exactly one %s node named *%s, found %d" % (op or "", suffix, len(hits)))
        return hits[0]

    def back_to(self, tensor, ops):
        """follow the first input through shape-only ops until a node of one of `ops` produced the tensor"""
        while True:
            n = self.producer.get(tensor)
            if n is None:
                raise ValueError("tensor %s has no producer of type %s" % (tensor, ops))
            if n.op in ops:
                return n
            if n.op not in _PASS:
                raise ValueError("unexpected %s node %s while tracing back to %s" % (n.op, n.name, ops))
            tensor = n.inputs[0]

    def forward_to(self, tensor, op):
        """follow the data consumers (ignoring Shape side branches) through shape-only ops to the first `op` node"""
        while True:
            cs = [c for c in self.consumers.get(tensor, []) if c.op != "Shape"]
            hit = [c for c in cs if c.op == op]
            if hit:
                return hit[0], tensor
            cs = [c for c in cs if c.op in _PASS and c.inputs[0] == tensor]
            if len(cs) != 1:
                raise ValueError("cannot follow %s forward to a %s node" % (tensor, op))
            tensor = cs[0].outputs[0]

    def const_of(self, tensor):
        """the initializer behind `tensor` (possibly through shape-only ops) and the shape it is used with"""
        chain = []
        while tensor not in self.inits:
            n = self.producer.get(tensor)
            if n is None or n.op not in _PASS:
                raise ValueError("%s is not a constant" % tensor)
            chain.append(n)
            tensor = n.inputs[0]
        return np.asarray(self.inits[tensor], np.float32)


def ctc_decoder_weights_from_onnx(path, num_heads=4):
    """Keras-layout CTCDecoder tensors (names of DESIGN.md section 6) from the reference's `ctc_model.onnx` export.
    The exported BatchNormalization arrives folded to (scale, shift); it is returned as gamma = scale, beta = shift,
    moving_mean = 0, moving_variance = 1 - eps (eps = 1e-3), which reproduces scale * x + shift exactly."""
    nodes, inits = read_onnx(path)
    g = _Graph(nodes, inits)
    f32 = lambda name: np.asarray(inits[name], np.float32)  # noqa: E731
    w = {}

    def dense(pattern):
        """(kernel, bias) initializers of Dense layers matching `pattern` with one group = the layer index, ascending"""
        idx = sorted({int(m.group(1)) for n in inits for m in [re.fullmatch(pattern + r"/Tensordot/ReadVariableOp:0", n)] if m})
        return [(f32(pattern.replace(r"(\d+)", str(i)) + "/Tensordot/ReadVariableOp:0"),
                 f32(pattern.replace(r"(\d+)", str(i)) + "/BiasAdd/ReadVariableOp:0")) for i in idx]

    (w["project/kernel"], w["project/bias"]), = dense(r"dense_(\d+)")
    w["fully_connected/kernel"] = f32("fully_connected/Tensordot/ReadVariableOp:0")
    w["fully_connected/bias"] = f32("fully_connected/BiasAdd/ReadVariableOp:0")
    d = w["project/kernel"].shape[0]
    hs = d // num_heads
    blocks = sorted({int(m.group(1)) for n in nodes for m in [re.match(r"decoder_conformer_block_(\d+)/", n.name)] if m})
    for bi in blocks:
        blk = "decoder_conformer_block_%d" % bi

        def ln(dst, scope):
            idx = {int(m.group(1)) for n in nodes
                   for m in [re.fullmatch(re.escape(blk + "/" + scope) + r"layer_normalization_(\d+)/mul_3", n.name)] if m}
            if len(idx) != 1:
                raise ValueError("LayerNormalization under %s/%s not found" % (blk, scope))
            base = "%s/%slayer_normalization_%d" % (blk, scope, idx.pop())
            w[dst + "/gamma"] = f32(g.node(base + "/mul_3", "Mul").inputs[1])
            w[dst + "/beta"] = f32(g.node(base + "/add", "Add").inputs[1])

        ln(blk + "/ff_module_1/ln", "ff_module_1/")
        ln(blk + "/mhsa_module/ln", "mhsa_module/")
        ln(blk + "/conv_module/ln", "conv_module/")
        ln(blk + "/ff_module_2/ln", "ff_module_2/")
        ln(blk + "/ln", "")
        for ff in ("ff_module_1", "ff_module_2"):
            (k1, b1), (k2, b2) = dense(re.escape("%s/%s/" % (blk, ff)) + r"dense_(\d+)")
            p = "%s/%s" % (blk, ff)
            w[p + "/ffn1/kernel"], w[p + "/ffn1/bias"], w[p + "/ffn2/kernel"], w[p + "/ffn2/bias"] = k1, b1, k2, b2
        # attention: query = the Gemm feeding `truediv`; key = the other operand of the MatMul the scaled query enters;
        # value = the other operand of the MatMul the Softmax output enters.  tf2onnx turned einsum "BNI,HIO->BNHO"
        # into Gemm(transB = 1) with W[h * hs + o, i] = kernel[h, i, o].
        m = blk + "/mhsa_module/mha"
        mha = [n for n in nodes if re.fullmatch(re.escape(blk) + r"/mhsa_module/multi_head_attention_(\d+)/truediv", n.name)]
        if len(mha) != 1:
            raise ValueError("%s: attention scaling node not found" % blk)
        scope = mha[0].name[:-len("/truediv")]
        q_gemm = g.back_to(mha[0].inputs[0], ("Gemm",))
        qk, q_in = g.forward_to(mha[0].outputs[0], "MatMul")
        k_gemm = g.back_to([i for i in qk.inputs if i != q_in][0], ("Gemm",))
        sm = g.node(scope + "/Softmax", "Softmax")
        pv, p_in = g.forward_to(sm.outputs[0], "MatMul")
        v_gemm = g.back_to([i for i in pv.inputs if i != p_in][0], ("Gemm",))
        for nm, gm in (("query_kernel", q_gemm), ("key_kernel", k_gemm), ("value_kernel", v_gemm)):
            if gm.ints.get("transB") != 1:
                raise ValueError("%s: expected Gemm(transB=1) for %s" % (blk, nm))
            w[m + "/" + nm] = np.ascontiguousarray(g.const_of(gm.inputs[1]).reshape(num_heads, hs, d).transpose(0, 2, 1))
        badd = g.node(scope + "/add", "Add")
        o_gemm = g.back_to(badd.inputs[0], ("Gemm",))
        proj = g.const_of(o_gemm.inputs[1])                       # [.., O, H, I] -> projection_kernel[h, i, o]
        w[m + "/projection_kernel"] = np.ascontiguousarray(proj.reshape(d, num_heads, hs).transpose(1, 2, 0))
        w[m + "/projection_bias"] = f32(badd.inputs[1])
        c = blk + "/conv_module"
        conv = lambda suffix: g.node(c + suffix, "Conv")  # noqa: E731
        pw1 = conv("/pw_conv_1/conv1d")
        w[c + "/pw_conv_1/kernel"] = np.ascontiguousarray(g.const_of(pw1.inputs[1])[:, :, 0, 0].T[None])
        w[c + "/pw_conv_1/bias"] = f32(g.node(c + "/pw_conv_1/BiasAdd", "Add").inputs[1]).reshape(-1)
        dw = conv("/dw_conv/separable_conv2d/depthwise")
        w[c + "/dw_conv/depthwise_kernel"] = np.ascontiguousarray(g.const_of(dw.inputs[1])[:, 0, 0, :].T[:, :, None])
        pw = conv("/dw_conv/BiasAdd")
        w[c + "/dw_conv/pointwise_kernel"] = np.ascontiguousarray(g.const_of(pw.inputs[1])[:, :, 0, 0].T[None])
        w[c + "/dw_conv/bias"] = f32(pw.inputs[2]).reshape(-1)
        bn = [n for n in nodes if re.fullmatch(re.escape(c) + r"/batch_normalization_(\d+)/batchnorm/mul_1", n.name)]
        if len(bn) != 1:
            raise ValueError("%s: folded BatchNormalization not found" % blk)
        scale = f32(bn[0].inputs[1]).reshape(-1)
        shift = f32(g.node(bn[0].name[:-len("mul_1")] + "add_1", "Add").inputs[1]).reshape(-1)
        w[c + "/bn/gamma"], w[c + "/bn/beta"] = scale, shift
        w[c + "/bn/moving_mean"] = np.zeros_like(scale)
        w[c + "/bn/moving_variance"] = np.full_like(scale, 1.0 - 1e-3)
        pw2 = conv("/pw_conv_2/conv1d")
        w[c + "/pw_conv_2/kernel"] = np.ascontiguousarray(g.const_of(pw2.inputs[1])[:, :, 0, 0].T[None])
        w[c + "/pw_conv_2/bias"] = f32(g.node(c + "/pw_conv_2/BiasAdd", "Add").inputs[1]).reshape(-1)
    return w


# ---------------------------------------------------------------------------------------------------------
# Keras variable names -> C-ABI names
# ---------------------------------------------------------------------------------------------------------
_MEL_SCOPE = re.compile(r"^(mel_layer|melspectrogram(_\d+)?|spectrogram(_\d+)?)$")
_AUTO = re.compile(r"^(layer_normalization|dense|conv2d|multi_head_attention|batch_normalization|conv1d|separable_conv1d|"
                   r"tf_residual_stack|embedding)(?:_(\d+))?$")


def keras_names_to_abi(names):
    """Map the Keras variable names of one of the reference's sub-models (ConformerEncoder, CTCDecoder; e.g.
    `conformer_encoder/conformer_block_3/ff_module_1/dense_12/kernel:0`, as `model.weights` / the `.h5` `weight_names`
    attributes list them) to the C-ABI names of DESIGN.md section 6.  Keras numbers auto-named layers (`dense_12`,
    `layer_normalization_7`, ...) globally in construction order, so only the ORDER of the numbers inside one scope
    is meaningful: the first Dense of an FFModule is ffn1, the second ffn2; the first Conv2D of conv_subsampling is
    conv1, the second conv2 (conformer_blocks.py:76-86, 116-123).  Returns {keras name: abi name}; names that are
    not part of the inference path (e.g. optimizer slots) are left out.

    Typical use on a machine that has TensorFlow:
        m = keras_names_to_abi([v.name for v in model.weights])
        np.savez("encoder.npz", **{m[v.name]: v.numpy() for v in model.weights if v.name in m})"""
    anchors = ("mel_layer", "melspectrogram", "spectrogram", "conv_subsampling", "wav_layer", "wave_pick_model", "conformer_block_",
               "decoder_conformer_block_", "fully_connected", "dense", "embedding", "inp_embedding")
    parsed = []
    mel = {}
    names = list(names)
    # a LEAF frontend is present when any variable lives under a `leaf` scope (frontend.py:75, `name='leaf'`)
    has_leaf = any("leaf" in n.split(":")[0].split("/")[:-1] for n in names)
    for full in names:
        parts = full.split(":")[0].split("/")
        if parts == ["kernel"]:
            # leaf_audio/convolution.py:155-187: GaborConv1D creates its [n_filters, 2] kernel in __init__, before any layer
            # scope exists, so Keras prints it as a bare `kernel:0`.  Observed on the stand-in's emulation of Keras name scoping
            # (oracle/_tfshim, round 5), not on a TensorFlow build.  Only mapped next to `leaf/...` variables: an unscoped
            # `kernel:0` in the name list of a Melspectrogram model (an auxiliary or optimizer variable) is not a Gabor kernel.
            if has_leaf:
                mel[full] = "mel_layer/tfbanks_complex_conv/kernel"
            continue
        if "leaf" in parts[:-1]:
            # leaf_audio/frontend.py:75-194 (`name='leaf'`): leaf/{tfbanks_preemp, learnable_pooling, PCEN[/EMA], tfbanks_instancenorm}/...
            mel[full] = "/".join(["mel_layer"] + parts[parts.index("leaf") + 1:])
            continue
        start = next((i for i, p in enumerate(parts) if p.startswith(anchors)), None)
        if start is None:
            continue
        parts = parts[start:]
        if _MEL_SCOPE.match(parts[0]):
            # the Melspectrogram / Spectrogram layer (time_frequency.py:51-53, 160): the attribute is `mel_layer` but the layer
            # is auto-named after its class, so Keras prints conformer_encoder/melspectrogram[_n]/{real_kernels, imag_kernels,
            # Variable}:0 -- the filterbank is an unnamed K.variable.  (Executing the reference's classes showed this in round 5;
            # `mel_layer/...` is still accepted for files written by this repository's own tools.)
            leaf = parts[-1]
            if len(parts) == 2 and leaf in ("real_kernels", "imag_kernels"):
                mel[full] = "mel_layer/" + leaf
            elif len(parts) == 2 and (leaf.startswith("Variable") or leaf == "freq2mel"):
                mel[full] = "mel_layer/freq2mel"
            elif len(parts) > 2 and parts[0] == "mel_layer":
                # files written by this repository's own tools name the LEAF sub-layers by the attribute:
                # mel_layer/{tfbanks_complex_conv, learnable_pooling, PCEN[/EMA], tfbanks_instancenorm, tfbanks_preemp}/...
                mel[full] = "/".join(parts)
            continue
        if parts[0].startswith(("wave_pick_model", "wav_layer")):
            # WavePickModel (wav_model.py:108-131) is a Layer named wave_pick_model holding one Sequential: the scope is
            # wave_pick_model/sequential[_n]/<layer>/...; the C-ABI prefix is the attribute name, wav_layer
            parts = ["wav_layer"] + [q for q in parts[1:] if not re.match(r"^sequential(_\d+)?$", q)]
        parsed.append((full, parts))

    def auto_index(p):
        m = _AUTO.match(p)
        return (m.group(1), int(m.group(2) or 0)) if m else None

    # rank of every auto-numbered layer among its siblings of the same kind inside the same scope
    groups = {}
    for _, parts in parsed:
        for depth, p in enumerate(parts[:-1]):
            ai = auto_index(p)
            if ai:
                groups.setdefault((tuple(parts[:depth]), ai[0]), set()).add(ai[1])
    rank = {k: {n: i for i, n in enumerate(sorted(v))} for k, v in groups.items()}

    out = {}
    for full, parts in parsed:
        var = parts[-1]
        scope = []
        ok = True
        for depth, p in enumerate(parts[:-1]):
            ai = auto_index(p)
            if not ai:
                scope.append(p)
                continue
            kind, r = ai[0], rank[(tuple(parts[:depth]), ai[0])][ai[1]]
            parent = parts[depth - 1] if depth else ""
            if kind == "layer_normalization":
                scope.append("ln")
            elif kind == "batch_normalization":
                scope.append("bn")
            elif kind == "multi_head_attention":
                scope.append("mha")
            elif kind == "conv2d" and parent == "conv_subsampling":
                scope.append(("conv1", "conv2")[r] if r < 2 else None)
            elif kind == "dense" and parent == "conv_subsampling":
                scope.append("linear")
            elif kind == "dense" and parent.startswith("ff_module"):
                scope.append(("ffn1", "ffn2")[r] if r < 2 else None)
            elif kind == "dense" and depth == 0:
                scope.append("project")                 # CTCDecoder.project (conformer_blocks.py:400)
            elif kind == "embedding" and depth == 0:
                scope.append("inp_embedding")           # Translator.inp_embedding (conformer_blocks.py:536)
            elif kind == "separable_conv1d" and parent == "wav_layer":
                scope.append("sep_conv")
            elif kind == "tf_residual_stack" and parent == "wav_layer":
                scope.append("res_%d" % (r + 1))        # construction order (wav_model.py:118-124)
            elif kind == "conv1d" and parent == "wav_layer":
                # the strided Conv1D of each stage, then the final Conv1D(dout, 7): the last one in construction order
                n_here = len(rank[(tuple(parts[:depth]), "conv1d")])
                scope.append("final" if r == n_here - 1 else "conv_%d" % (r + 1))
            elif kind == "conv1d" and auto_index(parent) and auto_index(parent)[0] == "tf_residual_stack":
                scope.append(("conv5", "conv1")[r] if r < 2 else None)      # TFResidualStack.blocks (wav_model.py:78-92)
            else:
                ok = False
            if scope and scope[-1] is None:
                ok = False
        if ok:
            out[full] = "/".join(scope + [var])
    out.update(mel)
    return out


def keras_h5_to_abi(path):
    """{C-ABI name: array} from a Keras `.h5` weight file of one of the reference's sub-models (`model.save_weights`,
    `ctc_runners.py:272-325`): variables read with h5lite.keras_weights, names mapped with keras_names_to_abi.  The mel
    layer's DFT kernels / filterbank (non-trainable variables `real_kernels`, `imag_kernels` and an unnamed `Variable`)
    are mapped when present; everything that is not part of the inference path is dropped."""
    from . import h5lite
    raw = h5lite.keras_weights(path)
    m = keras_names_to_abi(list(raw))
    return {m[k]: np.asarray(v, np.float32) for k, v in raw.items() if k in m}


def tf_checkpoint_to_abi(path):
    """{C-ABI name: array} from a TensorFlow checkpoint in the tensor-bundle format (`model.save_weights(prefix)`,
    `tf.train.Checkpoint`, a SavedModel's `variables/`): tensors and their Keras variable names are read by
    tfbundle.Bundle (object graph -> full_name), then mapped with keras_names_to_abi.  For the ChunkConformer
    sub-models, whose Keras scopes differ from the C-ABI prefixes (front / encoder / picker / helper / decoder), use
    `tfbundle.Bundle(prefix).variables_by_name()` and a rename of your own."""
    from . import tfbundle
    raw = tfbundle.Bundle(tfbundle.checkpoint_prefix(path)).variables_by_name()
    m = keras_names_to_abi(list(raw))
    return {m[k]: np.asarray(v, np.float32) for k, v in raw.items() if k in m}


# ---------------------------------------------------------------------------------------------------------
# ChunkConformer: object-graph paths of a TensorFlow checkpoint -> C-ABI names
# ---------------------------------------------------------------------------------------------------------
_CHUNK_ROOT = {"front": "front", "encoder": "encoder", "phone_picker": "picker", "decoder": "decoder", "helper": "helper"}
_CHUNK_BLOCK = {"ffm1": "ff_module_1", "ffm2": "ff_module_2", "mhsam": "mhsa_module", "convm": "conv_module", "ln": "ln"}
_CHUNK_MHA = {"_query_dense": "query", "_key_dense": "key", "_value_dense": "value", "_output_dense": "attention_output"}
_CHUNK_MEL = {"dft_real_kernels": "real_kernels", "dft_imag_kernels": "imag_kernels", "freq2mel": "freq2mel"}


def chunk_checkpoint_keys_to_abi(keys):
    """Map the keys of a ChunkConformer checkpoint written by `model.save_weights(prefix)` / `tf.train.Checkpoint` to
    the C-ABI names of the chunk handle.  Checkpoint keys are *attribute paths* in the Python object graph
    (`encoder/conformer_blocks/3/ffm1/ffn1/kernel/.ATTRIBUTES/VARIABLE_VALUE`), which the reference's source fixes
    (chunk_conformer_blocks.py: ChunkConformer.{front, encoder, phone_picker, decoder, helper} :782-786; ChunkConformerBlock.
    {ffm1, mhsam, convm, ffm2, ln} :343-365; FFModule.{ln, ffn1, ffn2} :108-114; ChunkMHSAModule.{ln, mha} :146-147 with the
    Keras MultiHeadAttention sub-layers `_query_dense`, ...; ChunkConvModule.{ln, pw_conv_1, dw_conv, bn, pw_conv_2}
    :244-265; ChunkConformerEncoder.conformer_blocks :488; ChunkCTCDecoder.{project, decode_layers, fc} :588-609;
    ChunkConformerFront.{conv_subsampling, mel_layer} :414-421) -- unlike Keras variable *names*, they do not depend on
    layer auto-numbering.  Returns {checkpoint key: abi name}; optimizer slots and unknown paths are left out.
    Not verified against a real ChunkConformer checkpoint (none is available); the inverse construction from the C-ABI
    names is tested."""
    out = {}
    for key in keys:
        path = key.split("/.ATTRIBUTES/")[0].split("/")
        if len(path) < 3 or path[0] not in _CHUNK_ROOT or ".OPTIMIZER_SLOT" in key:
            continue
        root = _CHUNK_ROOT[path[0]]
        rest = path[1:]
        abi = None
        if root == "front":
            if rest[0] == "conv_subsampling" and len(rest) == 3 and rest[1] in ("conv1", "conv2", "linear"):
                abi = "front/conv_subsampling/%s/%s" % (rest[1], rest[2])
            elif rest[0] == "mel_layer" and len(rest) == 2 and rest[1] in _CHUNK_MEL:
                abi = "front/mel_layer/" + _CHUNK_MEL[rest[1]]
        else:
            lst = "conformer_blocks" if root == "encoder" else "decode_layers"
            if rest[0] == lst and len(rest) >= 4 and rest[1].isdigit() and rest[2] in _CHUNK_BLOCK:
                blk = ("encoder/chunk_conformer_block_%s" if root == "encoder" else root + "/block_%s") % rest[1]
                mod, tail = _CHUNK_BLOCK[rest[2]], rest[3:]
                if mod == "ln" and len(tail) == 1:
                    abi = "%s/ln/%s" % (blk, tail[0])
                elif mod == "mhsa_module" and tail[0] == "mha" and len(tail) == 3 and tail[1] in _CHUNK_MHA:
                    abi = "%s/mhsa_module/mha/%s/%s" % (blk, _CHUNK_MHA[tail[1]], tail[2])
                elif mod != "ln" and len(tail) == 2:
                    abi = "%s/%s/%s/%s" % (blk, mod, tail[0], tail[1])
            elif root != "encoder" and rest[0] == "project" and len(rest) == 2:
                abi = "%s/project/%s" % (root, rest[1])
            elif root != "encoder" and rest[0] == "fc" and len(rest) == 2:
                abi = "%s/fully_connected/%s" % (root, rest[1])
        if abi:
            out[key] = abi
    return out


def chunk_checkpoint_to_abi(path):
    """{C-ABI name: array} for a ChunkConformer handle from a TensorFlow tensor-bundle checkpoint of the reference's
    `ChunkConformer` model (see chunk_checkpoint_keys_to_abi); the DFT kernels are reshaped to the handle's 2-D layout."""
    from . import tfbundle
    b = tfbundle.Bundle(tfbundle.checkpoint_prefix(path))
    m = chunk_checkpoint_keys_to_abi(b.keys())
    out = {}
    for key, abi in m.items():
        v = np.asarray(b.tensor(key), np.float32)
        if abi.endswith(("real_kernels", "imag_kernels")) and v.ndim == 4:
            v = v.reshape(v.shape[0], v.shape[-1])
        out[abi] = v
    return out


# ---------------------------------------------------------------------------------------------------------
# checkpoint directories (test_asr.py:95-114, ctc_runners.py:272-325)
# ---------------------------------------------------------------------------------------------------------
_STEP = re.compile(r"^model_(\d+)(\.npz|\.h5|\.hdf5|\.index|\.ckpt\.index|\.onnx)?$")
_PREFER = {".npz": 0, ".h5": 1, ".hdf5": 1, ".index": 2, ".ckpt.index": 2, ".onnx": 3}


def latest_checkpoint(checkpoint_dir):
    """The newest `model_<step>` of a `<outdir>/<sub-model>-ckpt/` directory, as the reference picks it
    (`files.sort(key=lambda x: int(x.split('_')[-1].replace('.h5', '')))`, test_asr.py:96-100): Keras `.h5` weight
    files as the trainers write them, TensorFlow checkpoints (`model_<step>.index` + data shards), or the `.npz` /
    `.onnx` forms this package also reads.  Files that do not look like `model_<step>.<ext>` are ignored; among
    several formats of the same step `.npz` wins, then `.h5`, then the TensorFlow bundle.  Returns the path to hand
    to `Model.load_weights` (for a TensorFlow bundle: the prefix without `.index`)."""
    best = None
    for f in os.listdir(checkpoint_dir):
        m = _STEP.match(f)
        if not m or not m.group(2):
            continue
        key = (int(m.group(1)), -_PREFER[m.group(2)])
        if best is None or key > best[0]:
            best = (key, f, m.group(2))
    if best is None:
        raise FileNotFoundError("no model_<step>.{h5,npz,index} in %s" % checkpoint_dir)
    path = os.path.join(checkpoint_dir, best[1])
    return path[:-len(".index")] if best[2].endswith(".index") else path


def latest_tf_checkpoint(checkpoint_dir):
    """`tf.train.latest_checkpoint(dir)` (test_chunk_asr.py:41, chunk_tester.py:74): the prefix named by the
    `model_checkpoint_path` line of the directory's `checkpoint` state file; without that file, the `*.index` with the
    highest trailing number."""
    state = os.path.join(checkpoint_dir, "checkpoint")
    if os.path.exists(state):
        for line in open(state, encoding="utf-8"):
            m = re.match(r'\s*model_checkpoint_path:\s*"(.*)"\s*$', line)
            if m:
                p = m.group(1)
                p = p if os.path.isabs(p) else os.path.join(checkpoint_dir, p)
                if os.path.exists(p + ".index"):
                    return p
    best = None
    for f in os.listdir(checkpoint_dir):
        m = re.match(r"^(.*?)(\d+)\.index$", f)
        if m and (best is None or int(m.group(2)) > best[0]):
            best = (int(m.group(2)), f[:-len(".index")])
    if best is None:
        raise FileNotFoundError("no TensorFlow checkpoint (checkpoint state file or *.index) in %s" % checkpoint_dir)
    return os.path.join(checkpoint_dir, best[1])
