"""Reader for TensorFlow checkpoints in the tensor-bundle format (`<prefix>.index` + `<prefix>.data-0000i-of-0000n`):
what `model.save_weights(prefix)` / `tf.train.Checkpoint.save` write -- the reference's ChunkConformer trainer among
them -- and what a SavedModel keeps under `variables/`.  Pure Python, no TensorFlow.

  * `.index` is an LevelDB-style sorted string table (tensorflow/core/lib/io/table*: data blocks with prefix-compressed
    keys and restart arrays, an index block, a 48-byte footer ending in the magic 0xdb4775248b80fb57; block trailer =
    1 compression byte + masked CRC32C).  Key "" holds the BundleHeaderProto, every other key a BundleEntryProto
    (dtype = 1, shape = 2, shard_id = 3, offset = 4, size = 5, crc32c = 6) -- tensorflow/core/protobuf/tensor_bundle.proto.
  * tensor bytes are read from the shard files and checked against the entry's masked CRC32C.
  * key `_CHECKPOINTABLE_OBJECT_GRAPH` holds the serialized TrackableObjectGraph (trackable_object_graph.proto): per
    node its children (local_name -> node) and attributes (checkpoint_key, full_name = the Keras variable name).
    `variables_by_name()` uses it to return {Keras variable name: array}, which checkpoint.keras_names_to_abi maps on.

Verified against the bundle the reference ships (`vad/online_vad_model/variables`): every tensor's CRC32C matches."""
import os
import struct

import numpy as np

_MAGIC = 0xDB4775248B80FB57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
           17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
_DT_STRING = 7


class BundleError(ValueError):
    pass


def _crc32c_table():
    poly = 0x82F63B78
    t = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ poly if c & 1 else c >> 1
        t.append(c)
    return np.asarray(t, np.uint32)


_T = _crc32c_table()


def crc32c(data):
    """CRC-32C (Castagnoli), byte-wise table; vectorised in 8 interleaved lanes is not needed at checkpoint sizes."""
    c = 0xFFFFFFFF
    t = _T.tolist()
    for b in bytes(data):
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(b, p):
    v = s = 0
    while True:
        c = b[p]
        p += 1
        v |= (c & 0x7F) << s
        if c < 0x80:
            return v, p
        s += 7


def _fields(b):
    p, n = 0, len(b)
    while p < n:
        key, p = _varint(b, p)
        f, t = key >> 3, key & 7
        if t == 0:
            v, p = _varint(b, p)
        elif t == 1:
            v, p = b[p:p + 8], p + 8
        elif t == 2:
            ln, p = _varint(b, p)
            v, p = b[p:p + ln], p + ln
        elif t == 5:
            v, p = b[p:p + 4], p + 4
        else:
            raise BundleError("unsupported protobuf wire type %d" % t)
        yield f, t, v


def _block(buf, off, size, verify=True):
    data = buf[off:off + size]
    ctype = buf[off + size]
    if verify:
        want = struct.unpack("<I", buf[off + size + 1:off + size + 5])[0]
        if masked_crc32c(buf[off:off + size + 1]) != want:
            raise BundleError("index block checksum mismatch at %d" % off)
    if ctype == 1:
        raise BundleError("snappy-compressed index blocks are not supported")
    if ctype != 0:
        raise BundleError("unknown block compression %d" % ctype)
    return data


def _block_entries(data):
    nrestart = struct.unpack("<I", data[-4:])[0]
    end = len(data) - 4 - 4 * nrestart
    p, key = 0, b""
    while p < end:
        shared, p = _varint(data, p)
        non_shared, p = _varint(data, p)
        vlen, p = _varint(data, p)
        key = key[:shared] + bytes(data[p:p + non_shared])
        p += non_shared
        yield key, data[p:p + vlen]
        p += vlen


class Bundle:
    def __init__(self, prefix, verify="auto"):
        """prefix: path without `.index` (e.g. `.../variables/variables`, `.../ckpt-12`); verify: True / "auto" / False"""
        self.prefix = prefix
        self.verify = verify
        with open(prefix + ".index", "rb") as fh:
            buf = fh.read()
        if len(buf) < 48 or struct.unpack("<Q", buf[-8:])[0] != _MAGIC:
            raise BundleError("%s.index: not a tensor-bundle index (bad table magic)" % prefix)
        foot = buf[-48:]
        _, p = _varint(foot, 0)                      # metaindex handle (offset, size)
        _, p = _varint(foot, p)
        ioff, p = _varint(foot, p)
        isize, p = _varint(foot, p)
        self.entries = {}
        self.header = {}
        for _, handle in _block_entries(_block(buf, ioff, isize, bool(verify))):
            boff, q = _varint(handle, 0)
            bsize, q = _varint(handle, q)
            for key, val in _block_entries(_block(buf, boff, bsize, bool(verify))):
                if key == b"":
                    self.header = {f: v for f, t, v in _fields(val) if t == 0}
                else:
                    self.entries[key.decode("utf8")] = self._entry(val)
        self.num_shards = self.header.get(1, 1)
        if self.header.get(2, 0) == 1:
            raise BundleError("big-endian bundles are not supported")
        self._shards = {}

    @staticmethod
    def _entry(val):
        e = {"dtype": 0, "shape": (), "shard": 0, "offset": 0, "size": 0, "crc": None, "sliced": False}
        for f, t, v in _fields(val):
            if f == 1:
                e["dtype"] = v
            elif f == 2:
                dims = []
                for g, u, w in _fields(v):
                    if g == 2:
                        d = 0
                        for h, x, y in _fields(w):
                            if h == 1:
                                d = y - (1 << 64) if y >= (1 << 63) else y
                        dims.append(d)
                e["shape"] = tuple(dims)
            elif f == 3:
                e["shard"] = v
            elif f == 4:
                e["offset"] = v
            elif f == 5:
                e["size"] = v
            elif f == 6:
                e["crc"] = struct.unpack("<I", v)[0]
            elif f == 7:
                e["sliced"] = True
        return e

    def keys(self):
        return sorted(self.entries)

    def _shard(self, i):
        if i not in self._shards:
            with open("%s.data-%05d-of-%05d" % (self.prefix, i, self.num_shards), "rb") as fh:
                self._shards[i] = fh.read()
        return self._shards[i]

    def raw(self, key):
        e = self.entries[key]
        if e["sliced"]:
            raise BundleError("%s: partitioned (sliced) variables are not supported" % key)
        data = self._shard(e["shard"])[e["offset"]:e["offset"] + e["size"]]
        if len(data) != e["size"]:
            raise BundleError("%s: data shard is truncated" % key)
        # verify=True: every tensor; verify="auto" (default): tensors up to 8 MB (the CRC is a pure-Python byte loop)
        check = self.verify is True or (self.verify == "auto" and e["size"] <= (8 << 20))
        if check and e["crc"] is not None and e["dtype"] != _DT_STRING and masked_crc32c(data) != e["crc"]:
            raise BundleError("%s: tensor checksum mismatch" % key)
        return e, data

    def tensor(self, key):
        e, data = self.raw(key)
        if e["dtype"] == _DT_STRING:
            n = int(np.prod(e["shape"])) if e["shape"] else 1
            p, lens = 0, []
            for _ in range(n):
                ln, p = _varint(data, p)
                lens.append(ln)
            p += 4                                           # masked crc32c of the lengths
            out = []
            for ln in lens:
                out.append(bytes(data[p:p + ln]))
                p += ln
            return out[0] if not e["shape"] else np.asarray(out, dtype=object).reshape(e["shape"])
        if e["dtype"] not in _DTYPES:
            raise BundleError("%s: unsupported dtype enum %d" % (key, e["dtype"]))
        return np.frombuffer(data, dtype=_DTYPES[e["dtype"]]).reshape(e["shape"]).copy()

    def object_graph(self):
        """[(node children {local_name: node_id}, attributes [(name, full_name, checkpoint_key)])] per node"""
        if "_CHECKPOINTABLE_OBJECT_GRAPH" not in self.entries:
            return []
        blob = self.tensor("_CHECKPOINTABLE_OBJECT_GRAPH")
        nodes = []
        for f, t, v in _fields(blob):
            if f != 1:
                continue
            children, attrs = {}, []
            for g, u, w in _fields(v):
                if g == 1:
                    nid, name = 0, ""
                    for h, x, y in _fields(w):
                        if h == 1:
                            nid = y
                        elif h == 2:
                            name = bytes(y).decode("utf8")
                    children[name] = nid
                elif g == 2:
                    a = {1: "", 2: "", 3: ""}
                    for h, x, y in _fields(w):
                        if h in a and x == 2:
                            a[h] = bytes(y).decode("utf8")
                    attrs.append((a[1], a[2], a[3]))
            nodes.append((children, attrs))
        return nodes

    def variables_by_name(self):
        """{Keras variable name (full_name, ':0' appended): array} for every VARIABLE_VALUE in the object graph; keys
        without an object-graph entry (name-based checkpoints) are returned under their checkpoint key."""
        out, seen = {}, set()
        for _, attrs in self.object_graph():
            for name, full, ckey in attrs:
                if name == "VARIABLE_VALUE" and ckey in self.entries and self.entries[ckey]["dtype"] != _DT_STRING:
                    out[(full or ckey) + (":0" if full and ":" not in full else "")] = self.tensor(ckey)
                    seen.add(ckey)
        for k, e in self.entries.items():
            if k not in seen and k != "_CHECKPOINTABLE_OBJECT_GRAPH" and e["dtype"] != _DT_STRING and not e["sliced"]:
                out.setdefault(k, self.tensor(k))
        return out


def checkpoint_prefix(path):
    """accepts `<prefix>`, `<prefix>.index`, a SavedModel directory or a directory with a `checkpoint` state file"""
    if path.endswith(".index"):
        return path[:-6]
    if os.path.isdir(path):
        for cand in (os.path.join(path, "variables", "variables"), os.path.join(path, "variables")):
            if os.path.exists(cand + ".index"):
                return cand
        state = os.path.join(path, "checkpoint")
        if os.path.exists(state):
            for line in open(state):
                if line.startswith("model_checkpoint_path:"):
                    return os.path.join(path, line.split(":", 1)[1].strip().strip('"'))
    return path
