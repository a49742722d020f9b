"""Keras HDF5 checkpoints without h5py.  PARITY UNPINNED: no .h5 file ships with the reference and
libhdf5 is absent, so the reader is checked against the package's writer and against structures assembled by hand here from the
HDF5 File Format Specification (a version-1 superblock, a chunked gzip+shuffle dataset, a compact dataset, a group stored as Link
messages, an object-header continuation block)."""
import importlib
import struct
import zlib

import numpy as np
import pytest

K = importlib.import_module("retinanet-for-table-detection_amd.keras_h5")
Wt = importlib.import_module("retinanet-for-table-detection_amd.weights")


def small_state(backbone, seed=0):
    """Every layer of the backbone + FPN + heads under the checkpoint's names, with tiny arrays of distinct values."""
    rng = np.random.RandomState(seed)
    st = {}
    for name, kh, kw, cin, cout, has_bias, bn in Wt.conv_layers(backbone, 1, 9):
        st[name + "/kernel"] = rng.standard_normal((kh, kw, 2, 3)).astype(np.float32)
        if has_bias:
            st[name + "/bias"] = rng.standard_normal(3).astype(np.float32)
        if bn:
            for p in ("gamma", "beta", "moving_mean", "moving_variance"):
                st[bn + "/" + p] = rng.standard_normal(3).astype(np.float32)
    return st


@pytest.mark.parametrize("backbone", ["resnet50", "resnet152"])
def test_writer_reader_round_trip(tmp_path, backbone):
    st = small_state(backbone)
    path = str(tmp_path / "weights.h5")
    K.save_keras_weights(path, st)
    back = K.load_keras_state(path)
    assert set(back) == set(st)
    for k in st:
        assert back[k].dtype == np.float32 and back[k].shape == st[k].shape and np.array_equal(back[k], st[k]), k
    raw = K.read_datasets(path)
    assert "/conv1/conv1/kernel:0" in raw and "/bn_conv1/bn_conv1/moving_variance:0" in raw and "/P3/P3/bias:0" in raw
    attrs = K.read_attributes(path)
    layers = list(attrs["/"]["layer_names"])
    # the ten head convs sit in the two nested submodels (model/defineModel.py:78-167), every other layer is a top-level group
    n_head = len(set(k.split("/")[0] for k in st if k.startswith(("pyramid_regression", "pyramid_classification"))))
    assert layers[0] == "conv1" and len(layers) == len(set(k.split("/")[0] for k in st)) - n_head + 2    # > 64: several leaves
    assert "regression_submodel" in layers and "classification_submodel" in layers and "pyramid_regression_0" not in layers
    assert "/regression_submodel/pyramid_regression_0/kernel:0" in raw and "/classification_submodel/pyramid_classification/bias:0" in raw
    assert not any(p.startswith("/pyramid_") for p in raw)
    wn = list(attrs["/regression_submodel"]["weight_names"])
    assert wn[:2] == ["pyramid_regression_0/kernel:0", "pyramid_regression_0/bias:0"] and len(wn) == 10
    assert wn[-2:] == ["pyramid_regression/kernel:0", "pyramid_regression/bias:0"]
    assert attrs["/"]["backend"] == "tensorflow"
    assert list(attrs["/bn_conv1"]["weight_names"]) == ["bn_conv1/gamma:0", "bn_conv1/beta:0", "bn_conv1/moving_mean:0", "bn_conv1/moving_variance:0"]
    assert list(attrs["/P5"]["weight_names"]) == ["P5/kernel:0", "P5/bias:0"]


def test_full_size_resnet50_state_round_trip(tmp_path):
    st = Wt.init_state("resnet50", 1, 9, seed=3)
    path = str(tmp_path / "training-x_resnet50_48.h5")
    K.save_keras_weights(path, st)
    back = K.load_keras_state(path)
    assert set(back) == set(st) and all(np.array_equal(back[k], np.asarray(st[k], np.float32)) for k in st)
    assert back["conv1/kernel"].shape == (7, 7, 3, 64) and back["pyramid_classification/bias"].shape == (9,)


# ---- a file assembled by hand, independent of the writer -------------------------------------------------------------------------
def _msg(t, body, flags=0):
    body += b"\0" * (-len(body) % 8)
    return struct.pack("<HHB3x", t, len(body), flags) + body


def _f32_type():
    return struct.pack("<BBBBI", 0x11, 0x20, 31, 0, 4) + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)


def _space(shape):
    return struct.pack("<BBB5x", 1, len(shape), 0) + b"".join(struct.pack("<Q", s) for s in shape)


def _link(name, addr):
    n = name.encode()
    return struct.pack("<BBB", 1, 0, len(n)) + n + struct.pack("<Q", addr)


def hand_made_file(nested=False):
    buf = bytearray(100)                                       # version-1 superblock: 100 bytes with the root entry

    def alloc(b):
        buf.extend(b"\0" * (-len(buf) % 8))
        at = len(buf)
        buf.extend(b)
        return at

    def header(msgs, split=False):
        if not split:
            body = b"".join(msgs)
            return alloc(struct.pack("<BxHII4x", 1, len(msgs), 1, len(body)) + body)
        # last message moved into a continuation block
        tail = alloc(msgs[-1])
        first = b"".join(msgs[:-1]) + _msg(0x0010, struct.pack("<QQ", tail, len(msgs[-1])))
        return alloc(struct.pack("<BxHII4x", 1, len(msgs) + 1, 1, len(first)) + first)

    # (a) chunked 5x6 float32, chunks 4x4, shuffle + gzip
    a = np.arange(30, dtype=np.float32).reshape(5, 6) * 0.5 - 3
    keys = []
    for oy in (0, 4):
        for ox in (0, 4):
            c = np.zeros((4, 4), np.float32)
            blk = a[oy:oy + 4, ox:ox + 4]
            c[:blk.shape[0], :blk.shape[1]] = blk
            raw = np.frombuffer(c.tobytes(), np.uint8).reshape(-1, 4).T.tobytes()          # shuffle
            z = zlib.compress(raw)
            keys.append((len(z), (oy, ox, 0), alloc(z)))
    node = b"TREE" + struct.pack("<BBHQQ", 1, 0, len(keys), K.UNDEF, K.UNDEF)
    for size, offs, at in keys:
        node += struct.pack("<II3Q", size, 0, *offs) + struct.pack("<Q", at)
    node += struct.pack("<II3Q", 0, 0, 8, 8, 0)
    bt = alloc(node)
    pipeline = struct.pack("<BB6x", 1, 2) + struct.pack("<HHHH", 2, 0, 0, 1) + struct.pack("<II", 4, 0) \
        + struct.pack("<HHHH", 1, 0, 0, 1) + struct.pack("<II", 6, 0)
    layout = struct.pack("<BBB", 3, 2, 3) + struct.pack("<Q", bt) + struct.pack("<III", 4, 4, 4)
    chunked = header([_msg(1, _space((5, 6))), _msg(3, _f32_type()), _msg(0x0B, pipeline), _msg(8, layout)], split=True)
    # (b) compact int16 vector
    v = np.array([3, -4, 5], "<i2")
    i16 = struct.pack("<BBBBI", 0x10, 8, 0, 0, 2) + struct.pack("<HH", 0, 16)
    compact = header([_msg(1, _space((3,))), _msg(3, i16), _msg(8, struct.pack("<BBH", 3, 0, v.nbytes) + v.tobytes())])
    # (c) contiguous float32 scalar-shaped (2,) array
    w = np.array([1.5, -2.25], np.float32)
    wat = alloc(w.tobytes())
    contig = header([_msg(1, _space((2,))), _msg(3, _f32_type()), _msg(8, struct.pack("<BBQQ", 3, 1, wat, w.nbytes))])
    # groups stored as Link messages (compact new-style groups): /layer/{kernel:0,bias:0}, root {layer, extra}
    linfo = struct.pack("<BB", 0, 0) + struct.pack("<QQ", K.UNDEF, K.UNDEF)
    layer = header([_msg(2, linfo), _msg(6, _link("kernel:0", chunked)), _msg(6, _link("bias:0", contig))])
    if nested:     # Keras' layout of a nested model: /regression_submodel/pyramid_regression_0/{kernel:0,bias:0}
        sub = header([_msg(2, linfo), _msg(6, _link("pyramid_regression_0", layer))])
        root = header([_msg(2, linfo), _msg(6, _link("regression_submodel", sub)), _msg(6, _link("extra", compact))])
    else:
        root = header([_msg(2, linfo), _msg(6, _link("layer", layer)), _msg(6, _link("extra", compact))])
    sb = K.SIGNATURE + struct.pack("<BBBBBBBBHHI", 1, 0, 0, 0, 0, 8, 8, 0, 4, 16, 0) + struct.pack("<HH", 32, 0)
    sb += struct.pack("<QQQQ", 0, K.UNDEF, len(buf), K.UNDEF) + struct.pack("<QQII16x", 0, root, 0, 0)
    assert len(sb) == 100
    buf[:100] = sb
    return bytes(buf), a, v, w


def test_reader_on_hand_assembled_structures(tmp_path):
    data, a, v, w = hand_made_file()
    path = tmp_path / "hand.h5"
    path.write_bytes(data)
    ds = K.read_datasets(str(path))
    assert set(ds) == {"/layer/kernel:0", "/layer/bias:0", "/extra"}
    assert np.array_equal(ds["/layer/kernel:0"], a) and ds["/layer/kernel:0"].dtype == np.float32
    assert np.array_equal(ds["/extra"], v) and ds["/extra"].dtype == np.dtype("<i2")
    assert np.array_equal(ds["/layer/bias:0"], w)
    # a 512-byte user block in front: the superblock is found at 512 and every address counts from its base-address field
    moved = bytearray(data)
    moved[28:36] = struct.pack("<Q", 512)
    shifted = tmp_path / "shifted.h5"
    shifted.write_bytes(b"\0" * 512 + bytes(moved))
    ds2 = K.read_datasets(str(shifted))
    assert set(ds2) == set(ds) and all(np.array_equal(ds2[k], ds[k]) for k in ds)


def test_reader_on_a_hand_assembled_nested_submodel_file(tmp_path):
    """The head convs of a real Keras file sit one group deeper, inside 'regression_submodel' / 'classification_submodel'
    (model/defineModel.py:78-167): the state keys still come out as '<inner layer>/<param>'."""
    data, a, v, w = hand_made_file(nested=True)
    path = tmp_path / "nested.h5"
    path.write_bytes(data)
    ds = K.read_datasets(str(path))
    assert set(ds) == {"/regression_submodel/pyramid_regression_0/kernel:0", "/regression_submodel/pyramid_regression_0/bias:0", "/extra"}
    st = K.load_keras_state(str(path))
    assert set(st) == {"pyramid_regression_0/kernel", "pyramid_regression_0/bias"}
    assert np.array_equal(st["pyramid_regression_0/kernel"], a) and np.array_equal(st["pyramid_regression_0/bias"], w)


def test_reader_refuses_what_it_does_not_implement(tmp_path):
    p = tmp_path / "x.h5"
    p.write_bytes(b"not hdf5 at all" * 10)
    with pytest.raises(K.H5FormatError, match="not an HDF5 file"):
        K.read_datasets(str(p))
    st = {"conv1/kernel": np.ones((1, 1, 1, 1), np.float32)}
    K.save_keras_weights(str(p), st)
    raw = bytearray(p.read_bytes())
    f = K._File(bytes(raw))
    raw[f.root:f.root + 4] = b"OHDR"
    p.write_bytes(bytes(raw))
    with pytest.raises(K.H5FormatError, match="version 2 object headers"):
        K.read_datasets(str(p))
    p.write_bytes(K.SIGNATURE + b"\x07" + b"\0" * 200)
    with pytest.raises(K.H5FormatError, match="superblock version 7"):
        K.read_datasets(str(p))
    # a file with datasets but nothing shaped like '<layer>/<param>'
    data = hand_made_file()[0]
    p.write_bytes(data)
    assert set(K.load_keras_state(str(p))) == {"layer/kernel", "layer/bias"}


@pytest.mark.gpu
def test_model_saves_and_loads_h5(tmp_path):
    import os
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "retinanet-for-table-detection_amd"))
    try:
        for k in [k for k in sys.modules if k == "model" or k.startswith("model.")]:
            del sys.modules[k]
        D = importlib.import_module("model.defineModel")
        m = D.Model("resnet50", 1, 9)
        m._root()._state = Wt.init_state("resnet50", 1, 9, seed=5, randomize_bn=True, cls_bias=-1.0, tame=True)
        path = str(tmp_path / "inferModel.h5")
        m.save(path)
        m2 = D.load_model(path)
        x = (np.random.RandomState(0).rand(1, 96, 128, 3).astype(np.float32) * 2 - 1)
        r1, c1 = m.predict_on_batch(x)
        r2, c2 = m2.predict_on_batch(x)
        assert np.array_equal(r1, r2) and np.array_equal(c1, c2) and np.isfinite(r1).all()
        torch.cuda.synchronize()
    finally:
        sys.path.pop(0)
