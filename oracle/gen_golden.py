"""oracle/gen_golden.py — emit golden vectors from the REFERENCE's own NumPy code.

Runs only in the build container (it needs /root/reference, which never travels to the GPU box).
TensorFlow/Keras/OpenCV are not installed, so the reference's pure-NumPy modules are imported
behind stub modules (SURVEY.md §8c recipe): model/anchors.py (all of it) and model/utils.py's
compute_overlap / compute_resize_scale / preprocess_image.  Only OUTPUT DATA is written:
tests/golden/ref_numpy_golden.npz (+ a JSON of hashes).  Usage:  python oracle/gen_golden.py
"""
import hashlib
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _import_reference():
    sys.dont_write_bytecode = True

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Base:
        def __init__(self, *a, **k):
            pass

    kb = mod("keras.backend", floatx=lambda: "float32")
    mod("keras", backend=kb, layers=mod("keras.layers", Layer=_Base),
        initializers=mod("keras.initializers", Initializer=_Base),
        callbacks=mod("keras.callbacks", Callback=_Base), utils=mod("keras.utils", Sequence=_Base))
    mod("tensorflow", config=types.SimpleNamespace(list_physical_devices=lambda kind: []))
    mod("keras_resnet", custom_objects={})
    mod("keras_resnet.models")
    mod("cv2")
    mod("matplotlib")
    sys.path.insert(0, REF)
    from model import utils as U      # must come first (circular import in the reference)
    from model import anchors as A
    return U, A


def sha16(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def main():
    U, A = _import_reference()
    out = {}
    meta = {}

    # ---- anchors for the config canvases
    for shape in [(800, 1333, 3), (1333, 800, 3), (1028, 800, 3), (1024, 1024, 3), (64, 96, 3), (37, 53, 3)]:
        a = A.anchors_for_shape(shape)
        key = "anchors_%dx%d" % shape[:2]
        meta[key] = {"n": int(a.shape[0]), "sum": float(a.sum()), "sha16": sha16(a)}
        out[key + "_head"] = a[:18].copy()
        out[key + "_tail"] = a[-18:].copy()
        if a.shape[0] < 5000:
            out[key + "_full"] = a
    for size in A.AnchorParameters_default.sizes:
        out["base_%d" % size] = A.generate_anchors(base_size=size, ratios=A.AnchorParameters_default.ratios,
                                                   scales=A.AnchorParameters_default.scales)
    # a non power-of-two size exercises the float32 product inside generate_anchors
    out["base_48"] = A.generate_anchors(base_size=48, ratios=A.AnchorParameters_default.ratios,
                                        scales=A.AnchorParameters_default.scales)
    out["guess_shapes_800x1333"] = np.array(A.guess_shapes((800, 1333, 3), [3, 4, 5, 6, 7]))

    # ---- targets: seeded GT sets on the 800x1333 canvas + small canvases
    rng = np.random.RandomState(1234)
    cases = []
    canvas = (800, 1333, 3)
    anchors = A.anchors_for_shape(canvas)
    fixed_gt = np.array([[100, 200, 700, 600], [50, 50, 300, 120], [900, 300, 1300, 780]], dtype=np.float64)
    cases.append(("kat3", canvas, [(800, 1333)], [fixed_gt]))
    cases.append(("kat2_img800x1000", canvas, [(800, 1000)], [fixed_gt[:2]]))
    cases.append(("empty", canvas, [(800, 1333)], [np.zeros((0, 4))]))
    for ci in range(3):
        B = 2
        shapes, gts = [], []
        for _ in range(B):
            g = rng.randint(1, 7)
            w = rng.uniform(80, 900, size=g)
            h = rng.uniform(60, 600, size=g)
            x1 = rng.uniform(0, 1333 - w)
            y1 = rng.uniform(0, 800 - h)
            gts.append(np.stack([x1, y1, x1 + w, y1 + h], axis=1))
            shapes.append((int(rng.randint(600, 801)), int(rng.randint(900, 1334))))
        cases.append(("rand%d" % ci, canvas, shapes, gts))
    # small canvas with fractional boxes and an exact-tie pair of identical GT boxes
    small = (96, 160, 3)
    dup = np.array([[10.5, 8.25, 70.0, 60.0], [10.5, 8.25, 70.0, 60.0], [80, 20, 150, 90]], dtype=np.float64)
    cases.append(("small_tie", small, [(96, 160), (90, 120)], [dup, dup[2:]]))

    for name, cshape, shapes, gts in cases:
        anc = A.anchors_for_shape(cshape)
        images = [np.zeros((h, w, 3), dtype=np.uint8) for (h, w) in shapes]
        ann = [{"bboxes": g, "labels": np.zeros((g.shape[0],))} for g in gts]
        reg, lab = A.anchor_targets_bbox(anc, images, ann, num_classes=1)
        assert reg.dtype == np.float32 and lab.dtype == np.float32
        out["tgt_%s_canvas" % name] = np.array(cshape[:2])
        out["tgt_%s_shapes" % name] = np.array(shapes)
        out["tgt_%s_gtcount" % name] = np.array([g.shape[0] for g in gts])
        out["tgt_%s_gt" % name] = np.concatenate([g.reshape(-1, 4) for g in gts], axis=0) if sum(g.shape[0] for g in gts) else np.zeros((0, 4))
        state = reg[:, :, 4]
        assert np.array_equal(state, lab[:, :, 1])
        for b in range(len(shapes)):
            pos = np.nonzero(state[b] == 1)[0].astype(np.int32)
            ign = np.nonzero(state[b] == -1)[0].astype(np.int32)
            out["tgt_%s_pos%d" % (name, b)] = pos
            out["tgt_%s_ign%d" % (name, b)] = ign
            out["tgt_%s_regpos%d" % (name, b)] = reg[b, pos, :4]
            out["tgt_%s_labpos%d" % (name, b)] = lab[b, pos, 0]
        meta["tgt_%s" % name] = {"reg_sha16": sha16(reg), "lab_sha16": sha16(lab)}

    # ---- overlaps sample
    sub = anchors[::997]
    out["overlap_boxes"] = sub
    out["overlap_gt"] = fixed_gt
    out["overlap_iou"] = U.compute_overlap(sub.astype(np.float64), fixed_gt)
    ov = U.compute_overlap(anchors.astype(np.float64), fixed_gt)
    out["overlap_max_per_gt"] = ov.max(axis=0)
    out["overlap_argmax_hist"] = np.bincount(np.argmax(ov, axis=1), minlength=3)

    # ---- resize scale / preprocess
    shapes = [(2200, 1712, 3), (3300, 2552, 3), (1100, 850, 3), (600, 2000, 3), (800, 1333, 3)]
    out["resize_shapes"] = np.array(shapes)
    out["resize_scales"] = np.array([U.compute_resize_scale(s) for s in shapes], dtype=np.float64)
    px = np.arange(256, dtype=np.uint8)
    out["preprocess_in"] = px
    out["preprocess_out"] = U.preprocess_image(px, mode="custom_tf")

    os.makedirs(OUT_DIR, exist_ok=True)
    np.savez_compressed(os.path.join(OUT_DIR, "ref_numpy_golden.npz"), **out)
    with open(os.path.join(OUT_DIR, "ref_numpy_golden.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", len(out), "arrays;", json.dumps(meta["anchors_800x1333"]))


if __name__ == "__main__":
    main()
