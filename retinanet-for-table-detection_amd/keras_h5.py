"""Keras HDF5 checkpoints by layer name, without h5py.

RetinaNet.py saves and reloads its models as Keras .h5 files (RetinaNet.py:70-79,153-163,320-340; model/Parameters.py:17).
h5py / libhdf5 are not available to this package, so the subset of the HDF5 file format those files use is read (and written)
here directly, following the HDF5 File Format Specification (version 0/1 superblock, version 1 object headers, symbol-table
groups = v1 B-tree + local heap + SNOD nodes, compact "Link" messages, contiguous / compact / chunked (optionally gzip +
shuffle) little-endian float and integer datasets, version 1-3 attributes with fixed-length strings).

  read_datasets(path)        {'/group/.../name': ndarray} for every dataset in the file
  read_attributes(path)      {'/group': {attr: value}} for the string / numeric attributes Keras writes
  load_keras_state(path)     {'<layer>/kernel': HWIO array, '<layer>/bias', '<bn>/gamma|beta|moving_mean|moving_variance'} from
                             either a save_weights() file ('/<layer>/<layer>/kernel:0') or a model.save() file ('/model_weights/...')
  save_keras_weights(path, state)   a save_weights()-style file: one group per layer with 'weight_names', root 'layer_names',
                             'backend', 'keras_version'

PARITY UNPINNED: the reference ships no .h5 file and libhdf5 is absent here, so reader and writer are validated against each
other and against hand-assembled structures only (tests/test_keras_h5.py); files using structures outside the subset above
(version 2 object headers, dense link storage, variable-length data) raise H5FormatError naming the structure.
"""
import struct
import zlib

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5FormatError(ValueError):
    pass


# ================================================================ reader ======================================================
class _File:
    def __init__(self, data):
        self.d = data
        base = 0
        while True:
            if self.d[base:base + 8] == SIGNATURE:
                break
            base = 512 if base == 0 else base * 2
            if base + 8 > len(self.d):
                raise H5FormatError("not an HDF5 file (no superblock signature)")
        ver = self.d[base + 8]
        if ver in (0, 1):
            self.O, self.L = self.d[base + 13], self.d[base + 14]
            p = base + 24 + (4 if ver == 1 else 0)
            self.base = self._addr(p)
            p += 4 * self.O
            # root symbol table entry: link name offset, object header address, cache type, reserved, scratch
            self.root = self._addr(p + self.O)
        elif ver in (2, 3):
            self.O, self.L = self.d[base + 9], self.d[base + 10]
            p = base + 12
            self.base = self._addr(p)
            self.root = self._addr(p + 3 * self.O)
        else:
            raise H5FormatError("superblock version %d is not supported" % ver)
        if self.O != 8 or self.L != 8:
            raise H5FormatError("only 8-byte offsets and lengths are supported (file has %d/%d)" % (self.O, self.L))

    def _addr(self, p):
        return struct.unpack_from("<Q", self.d, p)[0]

    # ---- object headers -------------------------------------------------------------------------------------------------------
    def messages(self, addr):
        """[(type, flags, bytes)] of the version 1 object header at `addr` (continuation blocks followed)."""
        a = addr + self.base
        if self.d[a:a + 4] == b"OHDR":
            raise H5FormatError("version 2 object headers (libver='latest' files) are not supported")
        if self.d[a] != 1:
            raise H5FormatError("object header version %d at %d is not supported" % (self.d[a], addr))
        nmsgs, = struct.unpack_from("<H", self.d, a + 2)
        size, = struct.unpack_from("<I", self.d, a + 8)
        blocks = [(a + 16, size)]
        out = []
        while blocks and len(out) < nmsgs:
            p, left = blocks.pop(0)
            end = p + left
            while p + 8 <= end and len(out) < nmsgs:
                mtype, msize, mflags = struct.unpack_from("<HHB", self.d, p)
                body = self.d[p + 8:p + 8 + msize]
                p += 8 + msize
                if mtype == 0x0010:
                    off, ln = struct.unpack_from("<QQ", body, 0)
                    blocks.append((off + self.base, ln))
                out.append((mtype, mflags, body))
        return out

    # ---- groups ---------------------------------------------------------------------------------------------------------------
    def _heap_name(self, heap_addr, offset):
        a = heap_addr + self.base
        if self.d[a:a + 4] != b"HEAP":
            raise H5FormatError("bad local heap signature at %d" % heap_addr)
        seg = self._addr(a + 8 + 2 * self.L) + self.base
        end = self.d.index(b"\0", seg + offset)
        return self.d[seg + offset:end].decode("utf-8")

    def _btree_group(self, node_addr, heap_addr, out):
        a = node_addr + self.base
        if self.d[a:a + 4] == b"SNOD":
            n, = struct.unpack_from("<H", self.d, a + 6)
            p = a + 8
            for _ in range(n):
                name_off, obj = struct.unpack_from("<QQ", self.d, p)
                out.append((self._heap_name(heap_addr, name_off), obj))
                p += 2 * self.O + 24
            return
        if self.d[a:a + 4] != b"TREE":
            raise H5FormatError("bad group B-tree node signature at %d" % node_addr)
        if self.d[a + 4] != 0:
            raise H5FormatError("group B-tree node of type %d" % self.d[a + 4])
        n, = struct.unpack_from("<H", self.d, a + 6)
        p = a + 8 + 2 * self.O
        for _ in range(n):
            p += self.L                                     # key i
            self._btree_group(self._addr(p), heap_addr, out)
            p += self.O

    def links(self, addr):
        """[(name, object header address)] of the group at `addr`, or None when the object is not a group."""
        out, is_group = [], False
        for mtype, _, body in self.messages(addr):
            if mtype == 0x0011:                             # symbol table: B-tree + local heap
                bt, heap = struct.unpack_from("<QQ", body, 0)
                self._btree_group(bt, heap, out)
                is_group = True
            elif mtype == 0x0002:                           # link info (new-style group)
                is_group = True
                flags = body[1]
                p = 2 + (8 if flags & 1 else 0)
                fheap = struct.unpack_from("<Q", body, p)[0]
                if fheap != UNDEF:
                    raise H5FormatError("dense link storage (fractal heap) is not supported")
            elif mtype == 0x0006:                           # link
                is_group = True
                flags = body[1]
                p = 2
                ltype = 0
                if flags & 8:
                    ltype = body[p]
                    p += 1
                if flags & 4:
                    p += 8
                if flags & 16:
                    p += 1
                w = 1 << (flags & 3)
                ln = int.from_bytes(body[p:p + w], "little")
                p += w
                name = body[p:p + ln].decode("utf-8")
                p += ln
                if ltype == 0:
                    out.append((name, struct.unpack_from("<Q", body, p)[0]))
        return out if is_group else None

    # ---- datasets -------------------------------------------------------------------------------------------------------------
    @staticmethod
    def _dtype(body):
        cls, ver = body[0] & 15, body[0] >> 4
        bits0 = body[1]
        size, = struct.unpack_from("<I", body, 4)
        order = ">" if bits0 & 1 else "<"
        if cls == 0:
            return np.dtype("%s%s%d" % (order, "i" if bits0 & 8 else "u", size))
        if cls == 1:
            return np.dtype("%sf%d" % (order, size))
        if cls == 3:
            return np.dtype("S%d" % size)
        if cls == 9:
            raise H5FormatError("variable-length data is not supported")
        raise H5FormatError("datatype class %d (version %d) is not supported" % (cls, ver))

    @staticmethod
    def _shape(body):
        ver, rank, flags = body[0], body[1], body[2]
        p = 8 if ver == 1 else 4
        if ver not in (1, 2):
            raise H5FormatError("dataspace message version %d" % ver)
        return tuple(struct.unpack_from("<%dQ" % rank, body, p)) if rank else ()

    def _chunks(self, node_addr, ndim, out):
        a = node_addr + self.base
        if self.d[a:a + 4] != b"TREE" or self.d[a + 4] != 1:
            raise H5FormatError("bad chunk B-tree node at %d" % node_addr)
        level = self.d[a + 5]
        n, = struct.unpack_from("<H", self.d, a + 6)
        p = a + 8 + 2 * self.O
        ksize = 8 + 8 * ndim
        for _ in range(n):
            csize, fmask = struct.unpack_from("<II", self.d, p)
            offs = struct.unpack_from("<%dQ" % ndim, self.d, p + 8)
            child = self._addr(p + ksize)
            if level == 0:
                out.append((offs[:-1], csize, fmask, child))
            else:
                self._chunks(child, ndim, out)
            p += ksize + self.O

    def dataset(self, addr):
        """ndarray of the dataset object at `addr`, or None when the object has no data layout."""
        shape = dtype = layout = None
        filters = []
        for mtype, _, body in self.messages(addr):
            if mtype == 0x0001:
                shape = self._shape(body)
            elif mtype == 0x0003:
                dtype = self._dtype(body)
            elif mtype == 0x0008:
                layout = body
            elif mtype == 0x000B:
                filters = self._filters(body)
        if layout is None or shape is None or dtype is None:
            return None
        n = int(np.prod(shape, dtype=np.int64)) if shape else 1
        if layout[0] != 3:
            raise H5FormatError("data layout message version %d is not supported" % layout[0])
        cls = layout[1]
        if cls == 0:
            sz, = struct.unpack_from("<H", layout, 2)
            raw = layout[4:4 + sz]
        elif cls == 1:
            a, sz = struct.unpack_from("<QQ", layout, 2)
            raw = b"\0" * (n * dtype.itemsize) if a == UNDEF else self.d[a + self.base:a + self.base + sz]
        elif cls == 2:
            ndim = layout[2]
            bt = self._addr_from(layout, 3)
            cdims = struct.unpack_from("<%dI" % ndim, layout, 3 + self.O)[:-1]
            arr = np.zeros(shape, dtype)
            if bt != UNDEF:
                chunks = []
                self._chunks(bt, ndim, chunks)
                for offs, csize, fmask, caddr in chunks:
                    buf = self.d[caddr + self.base:caddr + self.base + csize]
                    for k in range(len(filters) - 1, -1, -1):
                        if fmask & (1 << k):
                            continue
                        fid = filters[k][0]
                        if fid == 1:
                            buf = zlib.decompress(buf)
                        elif fid == 2:
                            es = dtype.itemsize
                            buf = np.frombuffer(buf, np.uint8).reshape(es, -1).T.tobytes()
                        elif fid == 3:
                            buf = buf[:-4]                  # fletcher32 checksum appended
                        else:
                            raise H5FormatError("filter %d is not supported" % fid)
                    chunk = np.frombuffer(buf, dtype, count=int(np.prod(cdims))).reshape(cdims)
                    sel = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, shape))
                    arr[sel] = chunk[tuple(slice(0, s.stop - s.start) for s in sel)]
            return arr
        else:
            raise H5FormatError("data layout class %d is not supported" % cls)
        return np.frombuffer(raw, dtype, count=n).reshape(shape).copy()

    @staticmethod
    def _filters(body):
        """[(filter id, client values)] of a filter pipeline message (version 1 or 2)."""
        ver, nf = body[0], body[1]
        p = 8 if ver == 1 else 2
        out = []
        for _ in range(nf):
            fid, = struct.unpack_from("<H", body, p)
            p += 2
            nlen = 0
            if ver == 1 or fid >= 256:
                nlen, = struct.unpack_from("<H", body, p)
                p += 2
            _flags, ncv = struct.unpack_from("<HH", body, p)
            p += 4
            p += (nlen + 7) // 8 * 8 if ver == 1 else nlen
            out.append((fid, struct.unpack_from("<%dI" % ncv, body, p)))
            p += 4 * ncv
            if ver == 1 and ncv % 2:
                p += 4
        return out

    def _addr_from(self, buf, p):
        return struct.unpack_from("<Q", buf, p)[0]

    def attributes(self, addr):
        out = {}
        for mtype, _, body in self.messages(addr):
            if mtype != 0x000C:
                continue
            ver = body[0]
            nsz, tsz, ssz = struct.unpack_from("<HHH", body, 2)
            p = 8 + (1 if ver == 3 else 0)
            pad = (lambda v: (v + 7) // 8 * 8) if ver == 1 else (lambda v: v)
            name = body[p:p + nsz].split(b"\0")[0].decode("utf-8")
            p += pad(nsz)
            tbody = body[p:p + tsz]
            p += pad(tsz)
            sbody = body[p:p + ssz]
            p += pad(ssz)
            try:
                dt, shape = self._dtype(tbody), self._shape(sbody)
            except H5FormatError:
                continue                                    # e.g. variable-length strings: skipped, not needed for loading by name
            n = int(np.prod(shape, dtype=np.int64)) if shape else 1
            val = np.frombuffer(body[p:p + n * dt.itemsize], dt, count=n).reshape(shape)
            if dt.kind == "S":
                val = np.array([v.split(b"\0")[0].decode("utf-8") for v in val.reshape(-1)], dtype=object).reshape(shape)
            out[name] = val if shape else val.reshape(-1)[0]
        return out

    def walk(self):
        """Yields (path, address, links-or-None) depth first from the root group."""
        seen = set()
        stack = [("", self.root)]
        while stack:
            path, addr = stack.pop()
            if addr in seen:
                continue
            seen.add(addr)
            ln = self.links(addr)
            yield path or "/", addr, ln
            if ln:
                for name, child in sorted(ln, reverse=True):
                    stack.append((path + "/" + name, child))


def _open(path):
    with open(path, "rb") as f:
        return _File(f.read())


def read_datasets(path):
    f = _open(path)
    out = {}
    for p, addr, ln in f.walk():
        if ln is None:
            arr = f.dataset(addr)
            if arr is not None:
                out[p] = arr
    return out


def read_attributes(path):
    f = _open(path)
    return {p: f.attributes(addr) for p, addr, ln in f.walk() if ln is not None}


def load_keras_state(path):
    """A Keras weight file -> {'<layer>/<param>'} with Keras' own layouts (Conv2D kernels HWIO).  Keys come from the last two path
    components '<layer>/<param>:0' so both save_weights() and model.save() ('/model_weights/...') files, with or without the
    repeated layer-name level, are accepted."""
    state = {}
    for p, arr in read_datasets(path).items():
        parts = [q for q in p.split("/") if q]
        if len(parts) < 2 or parts[0] == "optimizer_weights":
            continue
        param = parts[-1].split(":")[0]
        key = parts[-2] + "/" + param
        if key in state and not np.array_equal(state[key], arr):
            raise H5FormatError("two different datasets map to %s" % key)
        state[key] = arr
    if not state:
        raise H5FormatError("no layer weights found in %s" % path)
    return state


# ================================================================ writer ======================================================
GROUP_LEAF_K, GROUP_INTERNAL_K = 32, 16


def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


def _msg(mtype, body, flags=0):
    body = _pad8(body)
    return struct.pack("<HHB3x", mtype, len(body), flags) + body


def _dtype_msg(dt):
    dt = np.dtype(dt)
    if dt.kind == "f" and dt.itemsize == 4:
        return struct.pack("<BBBBI", 0x11, 0x20, 31, 0, 4) + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
    if dt.kind == "f" and dt.itemsize == 8:
        return struct.pack("<BBBBI", 0x11, 0x20, 63, 0, 8) + struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
    if dt.kind in "iu":
        return struct.pack("<BBBBI", 0x10, 8 if dt.kind == "i" else 0, 0, 0, dt.itemsize) + struct.pack("<HH", 0, 8 * dt.itemsize)
    if dt.kind == "S":
        return struct.pack("<BBBBI", 0x13, 0, 0, 0, dt.itemsize)              # null-terminated ASCII
    raise H5FormatError("cannot write dtype %s" % dt)


def _space_msg(shape):
    return struct.pack("<BBB5x", 1, len(shape), 0) + b"".join(struct.pack("<Q", int(s)) for s in shape)


def _attr_msg(name, value):
    if isinstance(value, (list, tuple)) or (isinstance(value, np.ndarray) and value.dtype.kind in "OU"):
        strs = [str(v).encode("utf-8") for v in value]
        w = max([len(s) for s in strs] + [1])
        value = np.array(strs, dtype="S%d" % w)
    elif isinstance(value, (str, bytes)):
        b = value.encode("utf-8") if isinstance(value, str) else value
        value = np.array(b, dtype="S%d" % max(len(b), 1))
    value = np.asarray(value)
    nm = name.encode("utf-8") + b"\0"
    t, s = _dtype_msg(value.dtype), _space_msg(value.shape)
    body = struct.pack("<BxHHH", 1, len(nm), len(t), len(s)) + _pad8(nm) + _pad8(t) + _pad8(s) + value.tobytes()
    return _msg(0x000C, body)


class _Writer:
    def __init__(self):
        self.buf = bytearray(96)                           # superblock written last

    def alloc(self, data):
        self.buf += b"\0" * (-len(self.buf) % 8)
        at = len(self.buf)
        self.buf += data
        return at

    def header(self, msgs):
        body = b"".join(msgs)
        return self.alloc(struct.pack("<BxHII4x", 1, len(msgs), 1, len(body)) + body)

    def dataset(self, arr):
        arr = np.ascontiguousarray(arr)
        if arr.dtype.byteorder == ">":
            arr = arr.astype(arr.dtype.newbyteorder("<"))
        at = self.alloc(arr.tobytes()) if arr.size else UNDEF
        msgs = [_msg(0x0001, _space_msg(arr.shape)), _msg(0x0003, _dtype_msg(arr.dtype), flags=1),
                _msg(0x0005, struct.pack("<BBBB", 2, 2, 0, 0)),
                _msg(0x0008, struct.pack("<BBQQ", 3, 1, at, arr.nbytes))]
        return self.header(msgs)

    def group(self, children, attrs=None):
        """children: {name: object header address} -> address of the group's object header."""
        names = sorted(children, key=lambda s: s.encode("utf-8"))
        heap = bytearray(b"\0" * 8)                         # offset 0: the empty name
        offs = {}
        for n in names:
            offs[n] = len(heap)
            heap += _pad8(n.encode("utf-8") + b"\0")
        free = len(heap)
        heap += struct.pack("<QQ", 1, 16)                   # one free block: next = 1 (none), size 16
        seg = self.alloc(bytes(heap))
        heap_at = self.alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), free, seg))
        cap = 2 * GROUP_LEAF_K
        leaves = [names[i:i + cap] for i in range(0, len(names), cap)] or [[]]
        if len(leaves) > 2 * GROUP_INTERNAL_K:
            raise H5FormatError("group with %d entries needs a deeper B-tree than this writer builds" % len(names))
        keys, kids = [0], []
        for leaf in leaves:
            ent = b""
            for n in leaf:
                ent += struct.pack("<QQII16x", offs[n], children[n], 0, 0)
            ent += b"\0" * (40 * (cap - len(leaf)))
            kids.append(self.alloc(b"SNOD" + struct.pack("<BxH", 1, len(leaf)) + ent))
            keys.append(offs[leaf[-1]] if leaf else 0)
        node = b"TREE" + struct.pack("<BBHQQ", 0, 0, len(kids), UNDEF, UNDEF)
        for i, k in enumerate(kids):
            node += struct.pack("<QQ", keys[i], k)
        node += struct.pack("<Q", keys[len(kids)])
        node += b"\0" * (24 + (2 * GROUP_INTERNAL_K + 1) * 8 + 2 * GROUP_INTERNAL_K * 8 - len(node))
        bt = self.alloc(node)
        msgs = [_msg(0x0011, struct.pack("<QQ", bt, heap_at))] + [_attr_msg(k, v) for k, v in (attrs or {}).items()]
        return self.header(msgs), bt, heap_at

    def finish(self, root):
        root_hdr, bt, heap_at = root
        sb = SIGNATURE + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, GROUP_LEAF_K, GROUP_INTERNAL_K, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, len(self.buf), UNDEF)
        sb += struct.pack("<QQII", 0, root_hdr, 1, 0) + struct.pack("<QQ", bt, heap_at)
        assert len(sb) == 96
        self.buf[:96] = sb
        return bytes(self.buf)


_PARAM_ORDER = ["kernel", "bias", "gamma", "beta", "moving_mean", "moving_variance"]


# The two head stacks are nested keras.models.Model instances (model/defineModel.py:78-167: names 'classification_submodel' /
# 'regression_submodel'): Keras stores a nested model as ONE top-level layer whose weights keep their inner layer names.
_SUBMODELS = (("pyramid_regression", "regression_submodel"), ("pyramid_classification", "classification_submodel"))


def _submodel_of(layer):
    for prefix, sub in _SUBMODELS:
        if layer == prefix or (layer.startswith(prefix + "_") and layer[len(prefix) + 1:].isdigit()):
            return sub
    return None


def save_keras_weights(path, state, keras_version="2.2.4", backend="tensorflow"):
    """Write {'<layer>/<param>': array} the way Keras' save_weights() lays the reference's model out: float32 datasets
    '/<layer>/<layer>/<param>:0' with 'weight_names' = ['<layer>/<param>:0', ...] on every layer group, EXCEPT the head convs,
    which live in the nested models of model/defineModel.py:78-167 and are therefore stored under ONE group per submodel:
    '/regression_submodel/pyramid_regression_0/kernel:0' ... with 'weight_names' = ['pyramid_regression_0/kernel:0', ...] and
    'regression_submodel' / 'classification_submodel' listed in the root's 'layer_names' (this is what
    load_weights(by_name=True) of RetinaNet.py:70-79 matches on).  Root attributes: 'layer_names', 'backend', 'keras_version'.
    Weights only: no 'model_config' (a model.save() file) is produced."""
    layers = {}
    for key, arr in state.items():
        layer, param = key.rsplit("/", 1)
        layers.setdefault(layer, {})[param] = np.asarray(arr, np.float32)
    w = _Writer()
    tops, order, nested = {}, [], {}

    def ordered(layer):
        return sorted(layers[layer], key=lambda q: _PARAM_ORDER.index(q) if q in _PARAM_ORDER else len(_PARAM_ORDER))

    for layer in layers:
        sub = _submodel_of(layer)
        if sub is None:
            params = ordered(layer)
            inner = w.group({p + ":0": w.dataset(layers[layer][p]) for p in params})[0]
            tops[layer] = w.group({layer: inner}, {"weight_names": ["%s/%s:0" % (layer, p) for p in params]})[0]
            order.append(layer)
        else:
            if sub not in nested:
                nested[sub] = []
                order.append(sub)                                   # the submodel sits where its first conv appears
            nested[sub].append(layer)
    for sub, members in nested.items():
        kids, names = {}, []
        for layer in members:
            params = ordered(layer)
            kids[layer] = w.group({p + ":0": w.dataset(layers[layer][p]) for p in params})[0]
            names += ["%s/%s:0" % (layer, p) for p in params]
        tops[sub] = w.group(kids, {"weight_names": names})[0]
    root = w.group(tops, {"layer_names": order, "backend": backend, "keras_version": keras_version})
    with open(path, "wb") as f:
        f.write(w.finish(root))
