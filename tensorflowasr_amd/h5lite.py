"""A small pure-Python HDF5 reader -- just what Keras `.h5` weight files need (h5py is not
available next to the system interpreter of this image).

Supported (HDF5 File Format Specification 3.0, the subset the HDF5 library writes with its default "earliest" format
bounds, which is what h5py / Keras produce): superblock versions 0-3; version-1 object headers with continuation
blocks (and version-2 "OHDR" headers without creation indices); old-style groups (symbol table = v1 B-tree + local
heap + SNOD nodes) and compact link messages; datasets with compact, contiguous or chunked (v1 B-tree) layout, the
deflate and shuffle filters; little/big-endian integer and IEEE float types; fixed-length and variable-length (global
heap) strings; attributes (message versions 1-3) of those types.  Not supported: dense link / attribute storage
(fractal heaps), version-4 chunk indices, virtual / external storage, compound types -- a clear error is raised.

    f = H5File(path); f.attrs["layer_names"]; f["layer/sub/kernel:0"].read() -> numpy array; f.visit() -> dataset paths

Verified against files written by a real HDF5 library (tests/golden/make_h5_fixtures.py, h5py 3.3 / HDF5 1.10.6)."""
import struct
import zlib

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(ValueError):
    pass


class _Type:
    def __init__(self, kind, size, dtype=None, vlen_string=False):
        self.kind, self.size, self.dtype, self.vlen_string = kind, size, dtype, vlen_string


class _Obj:
    """parsed object header: messages as (type, bytes)"""

    def __init__(self, f, addr):
        self.f, self.addr = f, addr
        self.msgs = f._read_header(addr)

    def first(self, mtype):
        for t, d in self.msgs:
            if t == mtype:
                return d
        return None


class H5File:
    def __init__(self, path):
        with open(path, "rb") as fh:
            self.buf = fh.read()
        b = self.buf
        base = 0
        while b[base:base + 8] != _SIG:                 # the superblock may sit at 0, 512, 1024, ...
            base = 512 if base == 0 else base * 2
            if base >= len(b):
                raise H5Error("%s: not an HDF5 file" % path)
        ver = b[base + 8]
        if ver in (0, 1):
            self.O, self.L = b[base + 13], b[base + 14]
            p = base + 24 + (4 if ver == 1 else 0)
            self.base = self._addr(p)
            p += 4 * self.O                                  # base, free-space, eof, driver info
            p += self.O                                      # root entry: link name offset
            root = self._addr(p)
        elif ver in (2, 3):
            self.O, self.L = b[base + 9], b[base + 10]
            p = base + 12
            self.base = self._addr(p)
            root = self._addr(p + 3 * self.O)
        else:
            raise H5Error("unsupported superblock version %d" % ver)
        if self.base == _UNDEF:
            self.base = 0
        self.root = Group(self, root, "/")
        self.attrs = self.root.attrs

    # ---- primitives
    def _int(self, p, n):
        return int.from_bytes(self.buf[p:p + n], "little")

    def _addr(self, p):
        return self._int(p, self.O)

    def _len(self, p):
        return self._int(p, self.L)

    def __getitem__(self, name):
        return self.root[name]

    def keys(self):
        return self.root.keys()

    def visit(self):
        """paths of all datasets below the root"""
        out = []

        def walk(g, prefix):
            for k in g.keys():
                o = g[k]
                if isinstance(o, Group):
                    walk(o, prefix + k + "/")
                else:
                    out.append(prefix + k)
        walk(self.root, "")
        return out

    # ---- object headers
    def _read_header(self, addr):
        b = self.buf
        a = self.base + addr
        msgs = []
        if b[a:a + 4] == b"OHDR":
            flags = b[a + 5]
            p = a + 6
            if flags & 0x20:
                p += 16
            if flags & 0x10:
                p += 4
            n = 1 << (flags & 3)
            size = self._int(p, n)
            p += n
            blocks = [(p, size)]
            while blocks:
                p, size = blocks.pop(0)
                end = p + size
                while p + 4 <= end:
                    t = b[p]
                    sz = self._int(p + 1, 2)
                    p += 4 + (2 if flags & 0x04 else 0)
                    d = b[p:p + sz]
                    p += sz
                    if t == 0x10:
                        ca, cl = self._addr_of(d, 0), int.from_bytes(d[self.O:self.O + self.L], "little")
                        blocks.append((self.base + ca + 4, cl - 8))           # skip "OCHK", drop the checksum
                    elif t != 0:
                        msgs.append((t, d))
            return msgs
        if b[a] != 1:
            raise H5Error("unsupported object header version %d at %d" % (b[a], addr))
        nmsg = self._int(a + 2, 2)
        size = self._int(a + 8, 4)
        blocks = [(a + 16, size)]
        while blocks and len(msgs) < nmsg + 64:
            p, size = blocks.pop(0)
            end = p + size
            while p + 8 <= end:
                t = self._int(p, 2)
                sz = self._int(p + 2, 2)
                d = b[p + 8:p + 8 + sz]
                p += 8 + sz
                if t == 0x10:
                    blocks.append((self.base + self._addr_of(d, 0), int.from_bytes(d[self.O:self.O + self.L], "little")))
                elif t != 0:
                    msgs.append((t, d))
        return msgs

    def _addr_of(self, d, off):
        return int.from_bytes(d[off:off + self.O], "little")

    # ---- message decoders
    def _dataspace(self, d):
        ver, rank, flags = d[0], d[1], d[2]
        if ver == 1:
            p = 8
        elif ver == 2:
            if d[3] == 2:                                    # null dataspace
                return None
            p = 4
        else:
            raise H5Error("dataspace message version %d" % ver)
        return tuple(int.from_bytes(d[p + i * self.L:p + (i + 1) * self.L], "little") for i in range(rank))

    def _datatype(self, d):
        cls, ver = d[0] & 0x0F, d[0] >> 4
        bits0 = d[1]
        size = int.from_bytes(d[4:8], "little")
        order = ">" if bits0 & 1 else "<"
        if cls == 0:
            signed = bool(bits0 & 0x08)
            return _Type("int", size, np.dtype("%s%s%d" % (order, "i" if signed else "u", size)))
        if cls == 1:
            return _Type("float", size, np.dtype("%sf%d" % (order, size)))
        if cls == 3:
            return _Type("string", size, np.dtype("S%d" % size))
        if cls == 9:
            is_str = (bits0 & 0x0F) == 1
            return _Type("vlen", size, vlen_string=is_str)
        raise H5Error("unsupported datatype class %d (version %d)" % (cls, ver))

    def _vlen_string(self, raw, i):
        e = 4 + self.O + 4
        ln = int.from_bytes(raw[i * e:i * e + 4], "little")
        col = int.from_bytes(raw[i * e + 4:i * e + 4 + self.O], "little")
        idx = int.from_bytes(raw[i * e + 4 + self.O:i * e + e], "little")
        if ln == 0 or col in (0, _UNDEF):
            return ""
        b = self.buf
        a = self.base + col
        if b[a:a + 4] != b"GCOL":
            raise H5Error("global heap collection expected at %d" % col)
        csize = self._len(a + 8)
        p, end = a + 8 + self.L, a + csize
        while p + 8 + self.L <= end:
            oi = self._int(p, 2)
            osz = self._len(p + 8)
            if oi == 0:
                break
            if oi == idx:
                return bytes(b[p + 8 + self.L:p + 8 + self.L + ln]).decode("utf8")
            p += 8 + self.L + ((osz + 7) & ~7)
        raise H5Error("global heap object %d not found" % idx)

    def _decode(self, typ, shape, raw):
        n = int(np.prod(shape)) if shape else 1
        if typ.kind == "vlen":
            if not typ.vlen_string:
                raise H5Error("variable-length sequences are not supported")
            vals = [self._vlen_string(raw, i) for i in range(n)]
            return vals[0] if not shape else np.asarray(vals, dtype=object).reshape(shape)
        arr = np.frombuffer(raw[:n * typ.size], dtype=typ.dtype, count=n)
        if typ.kind in ("int", "float"):
            arr = arr.astype(typ.dtype.newbyteorder("="))
        return arr.reshape(shape) if shape else arr.reshape(())[()]

    def _attributes(self, obj):
        out = {}
        for t, d in obj.msgs:
            if t == 0x15:                                       # attribute info: dense storage if it names a fractal heap
                q = 2 + (2 if d[1] & 1 else 0)
                if self._addr_of(d, q) != _UNDEF:
                    raise H5Error("dense attribute storage (fractal heap) is not supported")
                continue
            if t != 0x0C:
                continue
            ver = d[0]
            nsz, tsz, ssz = (int.from_bytes(d[2 + 2 * i:4 + 2 * i], "little") for i in range(3))
            if ver == 1:
                pad = lambda x: (x + 7) & ~7  # noqa: E731
                p = 8
            elif ver in (2, 3):
                pad = lambda x: x  # noqa: E731
                p = 8 if ver == 2 else 9
            else:
                raise H5Error("attribute message version %d" % ver)
            name = bytes(d[p:p + nsz]).split(b"\0")[0].decode("utf8")
            p += pad(nsz)
            typ = self._datatype(d[p:p + tsz])
            p += pad(tsz)
            shape = self._dataspace(d[p:p + ssz])
            p += pad(ssz)
            out[name] = None if shape is None else self._decode(typ, shape, d[p:])
        return out


class Group:
    def __init__(self, f, addr, name):
        self.f, self.name = f, name
        self.obj = _Obj(f, addr)
        self._links = None

    @property
    def attrs(self):
        return self.f._attributes(self.obj)

    def _load(self):
        if self._links is not None:
            return
        f = self.f
        links = {}
        st = self.obj.first(0x11)
        if st is not None:
            btree, heap = f._addr_of(st, 0), f._addr_of(st, f.O)
            h = f.base + heap
            if f.buf[h:h + 4] != b"HEAP":
                raise H5Error("local heap expected at %d" % heap)
            data = f.base + f._addr(h + 8 + 2 * f.L)

            def name_at(off):
                e = f.buf.index(b"\0", data + off)
                return bytes(f.buf[data + off:e]).decode("utf8")

            def walk(node):
                a = f.base + node
                if f.buf[a:a + 4] == b"SNOD":
                    n = f._int(a + 6, 2)
                    p = a + 8
                    for _ in range(n):
                        links[name_at(f._addr(p))] = f._addr(p + f.O)
                        p += 2 * f.O + 24
                    return
                if f.buf[a:a + 4] != b"TREE":
                    raise H5Error("B-tree node expected at %d" % node)
                used = f._int(a + 6, 2)
                p = a + 8 + 2 * f.O + f.L                   # skip the first key
                for _ in range(used):
                    walk(f._addr(p))
                    p += f.O + f.L
            if btree != _UNDEF:
                walk(btree)
        for t, d in self.obj.msgs:
            if t == 0x02 and len(d) >= 2:
                flags = d[1]
                p = 2 + (8 if flags & 1 else 0)
                if f._addr_of(d, p) != _UNDEF:
                    raise H5Error("dense link storage (fractal heap) is not supported")
            if t == 0x06:                                       # link message (compact new-style group)
                flags = d[1]
                p = 2
                ltype = 0
                if flags & 0x08:
                    ltype = d[p]
                    p += 1
                if flags & 0x04:
                    p += 8
                if flags & 0x10:
                    p += 1
                n = 1 << (flags & 3)
                ln = int.from_bytes(d[p:p + n], "little")
                p += n
                nm = bytes(d[p:p + ln]).decode("utf8")
                p += ln
                if ltype == 0:
                    links[nm] = f._addr_of(d, p)
        self._links = links

    def keys(self):
        self._load()
        return sorted(self._links)

    def __contains__(self, name):
        try:
            self[name]
            return True
        except KeyError:
            return False

    def __getitem__(self, name):
        node = self
        parts = [p for p in name.split("/") if p]
        for i, part in enumerate(parts):
            if not isinstance(node, Group):
                raise KeyError(name)
            node._load()
            if part not in node._links:
                raise KeyError(name)
            addr = node._links[part]
            obj = _Obj(self.f, addr)
            if obj.first(0x08) is not None:                      # has a data layout: dataset
                node = Dataset(self.f, obj, part)
            else:
                g = Group.__new__(Group)
                g.f, g.name, g.obj, g._links = self.f, part, obj, None
                node = g
        return node


class Dataset:
    def __init__(self, f, obj, name):
        self.f, self.obj, self.name = f, obj, name
        self.shape = f._dataspace(obj.first(0x01))
        self.type = f._datatype(obj.first(0x03))

    @property
    def attrs(self):
        return self.f._attributes(self.obj)

    def _filters(self):
        d = self.obj.first(0x0B)
        if d is None:
            return []
        ver, n = d[0], d[1]
        p = 8 if ver == 1 else 2
        out = []
        for _ in range(n):
            fid = int.from_bytes(d[p:p + 2], "little")
            p += 2
            nlen = 0
            if ver == 1 or fid >= 256:
                nlen = int.from_bytes(d[p:p + 2], "little")
                p += 2
            p += 2                                               # flags
            ncd = int.from_bytes(d[p:p + 2], "little")
            p += 2
            p += ((nlen + 7) & ~7) if ver == 1 else nlen
            cd = [int.from_bytes(d[p + 4 * i:p + 4 * i + 4], "little") for i in range(ncd)]
            p += 4 * ncd
            if ver == 1 and ncd % 2:
                p += 4
            out.append((fid, cd))
        return out

    def read(self):
        f = self.f
        lay = self.obj.first(0x08)
        ver, cls = lay[0], lay[1]
        if ver not in (3, 4) or (ver == 4 and cls == 2):
            raise H5Error("data layout message version %d (class %d) is not supported" % (ver, cls))
        shape = self.shape
        if shape is None:
            return None
        n = int(np.prod(shape)) if shape else 1
        esz = self.type.size if self.type.kind != "vlen" else 4 + f.O + 4
        if cls == 0:
            size = int.from_bytes(lay[2:4], "little")
            return f._decode(self.type, shape, lay[4:4 + size])
        if cls == 1:
            addr = f._addr_of(lay, 2)
            if addr == _UNDEF:
                return np.zeros(shape, self.type.dtype)
            return f._decode(self.type, shape, f.buf[f.base + addr:f.base + addr + n * esz])
        if cls != 2:
            raise H5Error("data layout class %d" % cls)
        nd = lay[2]                                              # rank + 1
        btree = f._addr_of(lay, 3)
        p = 3 + f.O
        cdims = [int.from_bytes(lay[p + 4 * i:p + 4 * i + 4], "little") for i in range(nd)]
        chunk_shape = tuple(cdims[:-1])
        if self.type.kind == "vlen":
            raise H5Error("chunked variable-length datasets are not supported")
        out = np.zeros(shape, self.type.dtype.newbyteorder("="))
        filters = self._filters()
        if btree == _UNDEF:
            return out

        def walk(node):
            a = f.base + node
            if f.buf[a:a + 4] != b"TREE" or f.buf[a + 4] != 1:
                raise H5Error("chunk B-tree node expected at %d" % node)
            level, used = f.buf[a + 5], f._int(a + 6, 2)
            ksz = 8 + 8 * nd
            p = a + 8 + 2 * f.O
            for _ in range(used):
                csize = f._int(p, 4)
                mask = f._int(p + 4, 4)
                offs = [f._int(p + 8 + 8 * i, 8) for i in range(nd - 1)]
                child = f._addr(p + ksz)
                if level > 0:
                    walk(child)
                else:
                    raw = bytes(f.buf[f.base + child:f.base + child + csize])
                    for i, (fid, cd) in reversed(list(enumerate(filters))):
                        if mask & (1 << i):
                            continue
                        if fid == 1:
                            raw = zlib.decompress(raw)
                        elif fid == 2:
                            es = cd[0] if cd else self.type.size
                            m = len(raw) // es
                            raw = np.frombuffer(raw[:m * es], np.uint8).reshape(es, m).T.tobytes() + raw[m * es:]
                        elif fid == 3:
                            raw = raw[:-4]                        # fletcher32 checksum (not verified)
                        else:
                            raise H5Error("unsupported filter %d" % fid)
                    chunk = np.frombuffer(raw, self.type.dtype, count=int(np.prod(chunk_shape))).reshape(chunk_shape)
                    sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, chunk_shape, shape))
                    out[sl] = chunk[tuple(slice(0, s.stop - s.start) for s in sl)]
                p += ksz + f.O
        walk(btree)
        return out


def keras_weights(path):
    """{variable name: array} of a Keras `save_weights(path.h5)` / `save(path.h5)` file (keras/saving/hdf5_format.py:
    root -- or the `model_weights` group -- has attr `layer_names`; each layer group has attr `weight_names` and the
    datasets under those names; attributes larger than 64 KB are split into `<name>0`, `<name>1`, ...)."""
    f = H5File(path)
    root = f.root
    if "layer_names" not in root.attrs and "model_weights" in root.keys():
        root = root["model_weights"]

    def attr_list(g, name):
        at = g.attrs
        if name in at:
            vals = list(np.atleast_1d(at[name]))
        else:
            vals, i = [], 0
            while "%s%d" % (name, i) in at:
                vals += list(np.atleast_1d(at["%s%d" % (name, i)]))
                i += 1
        return [v.decode("utf8") if isinstance(v, (bytes, np.bytes_)) else str(v) for v in vals]

    out = {}
    for layer in attr_list(root, "layer_names"):
        g = root[layer]
        for wn in attr_list(g, "weight_names"):
            out[wn] = np.asarray(g[wn].read())
    return out
