"""TEST INFRASTRUCTURE ONLY -- not part of the product path.

Minimal ONNX reader + NumPy interpreter, just large enough to execute the
reference's own exported CTCDecoder graph
(`Inference/PythonInference/asr/models/offline/ctc_model.onnx`, tf2onnx 1.9.3,
opset 13) without onnx / onnxruntime (neither is installed).  It is used once,
by `tests/golden/make_golden.py`, to (a) pull the trained CTCDecoder weights out
of the graph and (b) produce golden input/logit pairs that pin the NumPy
restatement in `oracle/conformer_oracle.py`.

Wire format facts follow the public ONNX protobuf schema (onnx.proto3):
ModelProto.graph=7; GraphProto.node=1/initializer=5/input=11/output=12;
NodeProto.input=1/output=2/name=3/op_type=4/attribute=5;
AttributeProto.name=1/f=2/i=3/s=4/t=5/floats=7/ints=8;
TensorProto.dims=1/data_type=2/float_data=4/int32_data=5/int64_data=7/name=8/raw_data=9.
"""
import struct
import numpy as np


# ----------------------------------------------------------------------------
# protobuf wire-format decoding
# ----------------------------------------------------------------------------
def _varint(buf, pos):
    result = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _fields(buf):
    """Yield (field_number, wire_type, value) for one message."""
    pos = 0
    n = len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            val = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError("unsupported wire type %d" % wt)
        yield fno, wt, val


def _signed(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _packed_varints(buf):
    out = []
    pos = 0
    while pos < len(buf):
        v, pos = _varint(buf, pos)
        out.append(_signed(v))
    return out


_DTYPES = {1: np.float32, 6: np.int32, 7: np.int64, 9: np.bool_, 11: np.float64}


def _tensor(buf):
    dims, dtype, name, raw = [], 1, "", None
    fdata, i32, i64 = [], [], []
    for fno, wt, val in _fields(buf):
        if fno == 1:
            dims += _packed_varints(val) if wt == 2 else [_signed(val)]
        elif fno == 2:
            dtype = val
        elif fno == 4:
            fdata += list(struct.unpack("<%df" % (len(val) // 4), val)) if wt == 2 \
                else [struct.unpack("<f", val)[0]]
        elif fno == 5:
            i32 += _packed_varints(val) if wt == 2 else [_signed(val)]
        elif fno == 7:
            i64 += _packed_varints(val) if wt == 2 else [_signed(val)]
        elif fno == 8:
            name = bytes(val).decode()
        elif fno == 9:
            raw = bytes(val)
    npdt = _DTYPES[dtype]
    if raw is not None:
        arr = np.frombuffer(raw, dtype=npdt).copy()
    elif fdata:
        arr = np.array(fdata, dtype=npdt)
    elif i64:
        arr = np.array(i64, dtype=npdt)
    elif i32:
        arr = np.array(i32, dtype=npdt)
    else:
        arr = np.zeros(0, dtype=npdt)
    return name, arr.reshape(dims) if dims else (arr.reshape(()) if arr.size == 1 else arr)


def _attribute(buf):
    name, val = "", None
    ints, floats = [], []
    for fno, wt, v in _fields(buf):
        if fno == 1:
            name = bytes(v).decode()
        elif fno == 2:
            val = struct.unpack("<f", v)[0]
        elif fno == 3:
            val = _signed(v)
        elif fno == 4:
            val = bytes(v)
        elif fno == 5:
            val = _tensor(v)[1]
        elif fno == 7:
            floats += list(struct.unpack("<%df" % (len(v) // 4), v)) if wt == 2 \
                else [struct.unpack("<f", v)[0]]
        elif fno == 8:
            ints += _packed_varints(v) if wt == 2 else [_signed(v)]
    if ints:
        val = ints
    elif floats:
        val = floats
    return name, val


class Node:
    __slots__ = ("op", "name", "inputs", "outputs", "attrs")

    def __init__(self):
        self.op, self.name, self.inputs, self.outputs, self.attrs = "", "", [], [], {}


def load(path):
    """-> (nodes, initializers{name: ndarray}, graph_inputs, graph_outputs)."""
    with open(path, "rb") as f:
        model = memoryview(f.read())
    graph = None
    for fno, _, val in _fields(model):
        if fno == 7:
            graph = val
    nodes, inits, gin, gout = [], {}, [], []
    for fno, _, val in _fields(graph):
        if fno == 1:
            nd = Node()
            for f2, _, v2 in _fields(val):
                if f2 == 1:
                    nd.inputs.append(bytes(v2).decode())
                elif f2 == 2:
                    nd.outputs.append(bytes(v2).decode())
                elif f2 == 3:
                    nd.name = bytes(v2).decode()
                elif f2 == 4:
                    nd.op = bytes(v2).decode()
                elif f2 == 5:
                    k, a = _attribute(v2)
                    nd.attrs[k] = a
            nodes.append(nd)
        elif fno == 5:
            name, arr = _tensor(val)
            inits[name] = arr
        elif fno in (11, 12):
            for f2, _, v2 in _fields(val):
                if f2 == 1:
                    (gin if fno == 11 else gout).append(bytes(v2).decode())
    gin = [n for n in gin if n not in inits]
    return nodes, inits, gin, gout


# ----------------------------------------------------------------------------
# interpreter (only the 25 op types the graph uses)
# ----------------------------------------------------------------------------
def _conv(x, w, b, attrs):
    """NCHW / OIHW convolution with auto_pad SAME_UPPER | explicit pads, groups."""
    n, c, h, wd = x.shape
    o, cg, kh, kw = w.shape
    group = attrs.get("group", 1)
    strides = attrs.get("strides", [1, 1])
    dil = attrs.get("dilations", [1, 1])
    assert dil == [1, 1]
    auto = attrs.get("auto_pad", b"NOTSET")
    if auto in (b"SAME_UPPER", b"SAME_LOWER"):
        pads = []
        for size, k, s in ((h, kh, strides[0]), (wd, kw, strides[1])):
            out = -(-size // s)
            tot = max((out - 1) * s + k - size, 0)
            lo = tot // 2 if auto == b"SAME_UPPER" else tot - tot // 2
            pads.append((lo, tot - lo))
    else:
        p = attrs.get("pads", [0, 0, 0, 0])
        pads = [(p[0], p[2]), (p[1], p[3])]
    xp = np.pad(x, ((0, 0), (0, 0), pads[0], pads[1]))
    oh = (xp.shape[2] - kh) // strides[0] + 1
    ow = (xp.shape[3] - kw) // strides[1] + 1
    y = np.zeros((n, o, oh, ow), dtype=np.float64)
    og = o // group
    for g in range(group):
        xs = xp[:, g * cg:(g + 1) * cg].astype(np.float64)
        ws = w[g * og:(g + 1) * og].astype(np.float64)
        for i in range(kh):
            for j in range(kw):
                patch = xs[:, :, i:i + oh * strides[0]:strides[0], j:j + ow * strides[1]:strides[1]]
                y[:, g * og:(g + 1) * og] += np.einsum("nchw,oc->nohw", patch, ws[:, :, i, j])
    if b is not None:
        y += b.reshape(1, -1, 1, 1)
    return y.astype(x.dtype)


def run(nodes, inits, feeds, outputs):
    env = dict(inits)
    env.update(feeds)
    for nd in nodes:
        i = [env[n] if n else None for n in nd.inputs]
        a = nd.attrs
        op = nd.op
        if op == "Reshape":
            shp = [int(i[0].shape[k]) if s == 0 else int(s) for k, s in enumerate(i[1].tolist())]
            r = i[0].reshape(shp)
        elif op == "Cast":
            r = i[0].astype(_DTYPES[a["to"]])
        elif op == "Gather":
            r = np.take(i[0], i[1], axis=a.get("axis", 0))
        elif op == "Shape":
            r = np.array(i[0].shape, dtype=np.int64)
        elif op == "Squeeze":
            r = np.squeeze(i[0], axis=tuple(int(v) for v in i[1].tolist())) if len(i) > 1 else np.squeeze(i[0])
        elif op == "Unsqueeze":
            r = i[0]
            for ax in sorted(int(v) for v in i[1].tolist()):
                r = np.expand_dims(r, ax)
        elif op == "Add":
            r = i[0] + i[1]
        elif op == "Sub":
            r = i[0] - i[1]
        elif op == "Mul":
            r = i[0] * i[1]
        elif op == "Div":
            r = i[0] / i[1] if i[0].dtype.kind == "f" else i[0] // i[1]
        elif op == "Max":
            r = np.maximum(i[0], i[1])
        elif op == "Concat":
            r = np.concatenate(i, axis=a["axis"])
        elif op == "ReduceProd":
            r = np.prod(i[0], axis=tuple(a["axes"]) if "axes" in a else None, keepdims=bool(a.get("keepdims", 1)))
        elif op == "ReduceMean":
            r = np.mean(i[0], axis=tuple(a["axes"]), keepdims=bool(a.get("keepdims", 1)))
        elif op == "ReduceSumSquare":
            r = np.sum(i[0] * i[0], axis=tuple(a["axes"]), keepdims=bool(a.get("keepdims", 1)))
        elif op == "Slice":
            starts, ends = i[1].tolist(), i[2].tolist()
            axes = i[3].tolist() if len(i) > 3 and i[3] is not None else list(range(len(starts)))
            steps = i[4].tolist() if len(i) > 4 and i[4] is not None else [1] * len(starts)
            sl = [slice(None)] * i[0].ndim
            for s, e, ax, st in zip(starts, ends, axes, steps):
                sl[ax] = slice(int(s), int(e), int(st))
            r = i[0][tuple(sl)]
        elif op == "Expand":
            r = i[0] * np.ones([int(v) for v in i[1].tolist()], dtype=i[0].dtype)
        elif op == "Transpose":
            r = np.transpose(i[0], a["perm"])
        elif op == "MatMul":
            r = np.matmul(i[0], i[1])
        elif op == "Gemm":
            A = i[0].T if a.get("transA", 0) else i[0]
            Bm = i[1].T if a.get("transB", 0) else i[1]
            r = a.get("alpha", 1.0) * (A @ Bm)
            if len(i) > 2 and i[2] is not None:
                r = r + a.get("beta", 1.0) * i[2]
            r = r.astype(i[0].dtype)
        elif op == "BatchNormalization":
            x, sc, bi, mean, var = i
            shp = [1, -1] + [1] * (x.ndim - 2)
            r = (x - mean.reshape(shp)) / np.sqrt(var.reshape(shp) + a.get("epsilon", 1e-5)) \
                * sc.reshape(shp) + bi.reshape(shp)
            r = r.astype(x.dtype)
        elif op == "Sigmoid":
            r = (1.0 / (1.0 + np.exp(-i[0].astype(np.float64)))).astype(i[0].dtype)
        elif op == "Conv":
            r = _conv(i[0], i[1], i[2] if len(i) > 2 else None, a)
        elif op == "Softmax":
            ax = a.get("axis", -1)
            z = i[0].astype(np.float64)
            z = z - z.max(axis=ax, keepdims=True)
            e = np.exp(z)
            r = (e / e.sum(axis=ax, keepdims=True)).astype(i[0].dtype)
        elif op == "Split":
            ax = a.get("axis", 0)
            if len(i) > 1 and i[1] is not None:
                idx = np.cumsum(i[1].tolist())[:-1]
                parts = np.split(i[0], idx, axis=ax)
            else:
                parts = np.split(i[0], len(nd.outputs), axis=ax)
            for nm, p in zip(nd.outputs, parts):
                env[nm] = p
            continue
        elif op == "Identity":
            r = i[0]
        else:
            raise NotImplementedError(op)
        env[nd.outputs[0]] = r
    return [env[o] for o in outputs]
