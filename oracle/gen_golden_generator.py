"""oracle/gen_golden_generator.py — golden vectors for the data-generator row from the REFERENCE's own
NumPy code: model/transform.py's matrix half and the array-only methods of csv_generator.Generator.

Runs only in the build container (needs /root/reference; nothing here travels to the GPU box except the OUTPUT DATA,
tests/golden/ref_generator_golden.npz).  cv2 / keras / tensorflow are absent, so the modules are imported behind the empty stub
modules of gen_golden.py; only functions that never touch those stubs are called:
  transform.random_transform_generator / adjust_transform_for_image / transform_aabb       (model/transform.py:17-42,190-272,324-340)
  Generator.filter_annotations, Generator.group_images, CSVGenerator.load_annotations      (csv_generator.py:159-171,192-218,497-512)
cv2.warpAffine itself (apply_transform) and cv2.imread cannot run here: the warp is "parity unpinned" (oracle/ref_generator.py).
Usage:  python oracle/gen_golden_generator.py
"""
import os
import random
import sys
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import _import_reference, OUT_DIR, REF  # noqa: E402

TRAIN_KW = dict(min_rotation=-0.1, max_rotation=0.1, min_translation=(-0.1, -0.1), max_translation=(0.1, 0.1), min_shear=-0.1,
                max_shear=0.1, min_scaling=(0.9, 0.9), max_scaling=(1.1, 1.1), flip_x_chance=0.5, flip_y_chance=0.5)  # RetinaNet.py:213-224


def main():
    _import_reference()
    sys.path.insert(0, os.path.join(REF, "FasterRCNN"))
    from model import transform as T
    import csv_generator as G
    out = {}

    # ---- random transforms with train()'s parameters, seeded
    gen = T.random_transform_generator(prng=np.random.RandomState(7), **TRAIN_KW)
    mats = np.stack([next(gen) for _ in range(6)])
    out["tf_seed"] = np.array(7)
    out["tf_matrices"] = mats
    gen1 = T.random_transform_generator(prng=np.random.RandomState(11), flip_x_chance=0.5)      # RetinaNet.py:201
    out["tf_flip_matrices"] = np.stack([next(gen1) for _ in range(6)]).astype(np.float64)
    shapes = [(2200, 1712, 3), (300, 500, 3)]
    out["tf_shapes"] = np.array(shapes)
    adj = []
    for m in mats:
        for s in shapes:
            adj.append(T.adjust_transform_for_image(m.copy(), np.zeros(s, np.uint8), True))
    out["tf_adjusted"] = np.stack(adj)
    out["tf_adjusted_abs"] = np.stack([T.adjust_transform_for_image(m.copy(), np.zeros(shapes[0], np.uint8), False) for m in mats])
    rng = np.random.RandomState(5)
    boxes = np.stack([rng.uniform(0, 800, 12), rng.uniform(0, 1000, 12)], 1)
    boxes = np.concatenate([boxes, boxes + rng.uniform(5, 600, (12, 2))], 1)
    out["tf_boxes"] = boxes
    out["tf_boxes_out"] = np.array([[T.transform_aabb(a, b) for b in boxes] for a in out["tf_adjusted"][::2]])

    # ---- filter_annotations: boxes on/over every edge of a (120,200,3) image
    b = np.array([[10, 10, 50, 40], [50, 10, 50, 40], [10, 40, 50, 40], [60, 20, 30, 90], [-1, 5, 20, 30], [5, -0.5, 20, 30],
                  [100, 50, 200, 120], [100, 50, 200.5, 100], [100, 50, 150, 120.25], [0, 0, 200, 120], [0, 0, 1, 1]], np.float64)
    ann = [{"labels": np.arange(len(b), dtype=np.float64), "bboxes": b.copy()},
           {"labels": np.zeros((0,)), "bboxes": np.zeros((0, 4))}]
    imgs = [np.zeros((120, 200, 3), np.uint8), np.zeros((50, 60, 3), np.uint8)]
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _, fa = G.Generator.filter_annotations(None, imgs, ann, [0, 1])
    out["flt_boxes_in"] = b
    out["flt_boxes_out"] = fa[0]["bboxes"]
    out["flt_labels_out"] = fa[0]["labels"]
    out["flt_empty_out"] = fa[1]["bboxes"]

    # ---- group_images: the three methods, with wrap-around of the last group
    ratios = np.random.RandomState(2).uniform(0.5, 1.6, 11)
    for method in ("none", "random", "ratio"):
        me = types.SimpleNamespace(group_method=method, batch_size=4, size=lambda: 11,
                                   image_aspect_ratio=lambda i: float(ratios[i]))
        random.seed(3)
        G.Generator.group_images(me)
        out["grp_%s" % method] = np.array(me.groups)
    out["grp_ratios"] = ratios

    # ---- CSVGenerator.load_annotations.  It reads gtBox.objClass, which FasterRCNN/Shapes.py's GroundTruthBox does not define
    # (it has obj_cls: a latent defect of the reference), so plain records carrying that attribute stand in.
    gts = [types.SimpleNamespace(x1=12, y1=30, x2=400, y2=310, objClass="table"),
           types.SimpleNamespace(x1=7.5, y1=8, x2=90, y2=77.25, objClass="table")]
    me = types.SimpleNamespace(image_data=[types.SimpleNamespace(gt_boxes=gts)], name_to_label=lambda n: {"table": 0}[n])
    la = G.CSVGenerator.load_annotations(me, 0)
    out["ann_boxes"] = la["bboxes"]
    out["ann_labels"] = la["labels"]

    np.savez_compressed(os.path.join(OUT_DIR, "ref_generator_golden.npz"), **out)
    print("wrote", len(out), "arrays:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
