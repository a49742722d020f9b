"""GPU: the reference's Python call surface (retinanet-for-table-detection_amd/model) and the preprocessing kernels.
NumPy-half functions are bit-exact with the reference-generated golden vectors; preprocessing is bit-exact with the oracle on
the integer stages and within float32 rounding on the bicubic resize."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

from oracle import ref_numpy as R
from oracle import ref_preprocess as P
from helpers import load_case

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def M():
    """`from model import ...` exactly as RetinaNet.py does, with the package directory on sys.path."""
    sys.path.insert(0, os.path.join(ROOT, "retinanet-for-table-detection_amd"))
    for k in [k for k in sys.modules if k == "model" or k.startswith("model.")]:
        del sys.modules[k]
    mods = {n: importlib.import_module("model." + n) for n in ("anchors", "losses", "layers", "utils", "defineModel", "initializers",
                                                              "Parameters", "preprocess")}
    yield type("Mods", (), mods)
    sys.path.pop(0)


def test_anchor_functions_match_reference_golden(M, golden):
    A = M.anchors
    for shape in [(800, 1333), (37, 53)]:
        assert np.array_equal(A.anchors_for_shape(shape + (3,)), R.anchors_for_shape(shape + (3,)))
    assert np.array_equal(A.generate_anchors(48, A.AnchorParameters_default.ratios, A.AnchorParameters_default.scales), golden["base_48"])
    assert np.array_equal(np.array(A.guess_shapes((800, 1333, 3), [3, 4, 5, 6, 7])), golden["guess_shapes_800x1333"])
    assert np.array_equal(A.shift((5, 7), 16, R.base_anchors(64)), R.shifted((5, 7), 16, R.base_anchors(64)))
    assert A.AnchorParameters_default.num_anchors() == 9
    assert np.array_equal(M.utils.compute_overlap(golden["overlap_boxes"], golden["overlap_gt"]), golden["overlap_iou"])
    anchors = R.anchors_for_shape((800, 1333, 3))
    pos, ign, arg = A.compute_gt_annotations(anchors, golden["overlap_gt"])
    wp, wi, wa = R.compute_gt_annotations(anchors, golden["overlap_gt"])
    assert np.array_equal(pos, wp) and np.array_equal(ign, wi) and np.array_equal(arg, wa)
    assert pos.sum() == 82 and ign.sum() == 149                               # SURVEY §8c known answers
    t = A.bbox_transform(anchors, golden["overlap_gt"][arg])
    assert np.array_equal(t, R.bbox_transform(anchors, golden["overlap_gt"][arg]))
    np.testing.assert_allclose(t[154145], [0.06394879, -0.1448556, 3.76413169, -0.27275318], rtol=0, atol=1e-8)


@pytest.mark.parametrize("name", ["kat2_img800x1000", "empty", "rand1", "small_tie"])
def test_anchor_targets_bbox_signature_and_bits(M, golden, name):
    canvas, shapes, gts = load_case(golden, name)
    anchors = M.anchors.anchors_for_shape(canvas + (3,))
    images = [np.zeros((h, w, 3), np.uint8) for (h, w) in shapes]
    ann = [{"bboxes": g, "labels": np.zeros((g.shape[0],))} for g in gts]
    reg, lab = M.anchors.anchor_targets_bbox(anchors, images, ann, num_classes=1)
    wreg, wlab = R.anchor_targets(anchors, shapes, gts, [np.zeros(len(g)) for g in gts], 1)
    assert reg.dtype == np.float32 and np.array_equal(reg, wreg) and np.array_equal(lab, wlab)
    with pytest.raises(AssertionError):
        M.anchors.anchor_targets_bbox(anchors, images, ann[:-1] if len(ann) > 1 else [], 1)
    with pytest.raises(AssertionError):
        M.anchors.anchor_targets_bbox(anchors, images, [{"labels": np.zeros(0)} for _ in images], 1)


def test_losses_layers_utils(M):
    rng = np.random.RandomState(0)
    B, N = 2, 3000
    state = rng.choice([-1.0, 0.0, 1.0], size=(B, N), p=[0.1, 0.8, 0.1]).astype(np.float32)
    lab = np.stack([(state == 1).astype(np.float32), state], axis=-1)
    p = rng.uniform(0.01, 0.99, size=(B, N, 1)).astype(np.float32)
    regt = np.concatenate([rng.normal(size=(B, N, 4)), state[..., None]], axis=-1).astype(np.float32)
    pred = (regt[..., :4] + rng.normal(scale=0.2, size=(B, N, 4))).astype(np.float32)
    fs, npos = R.focal_loss(lab, p)
    rs, nposr = R.smooth_l1_loss(regt, pred)
    assert abs(float(M.losses.focal()(lab, p)) - fs / max(1, npos)) <= 1e-5 * fs / max(1, npos)
    assert abs(float(M.losses.smooth_l1()(regt, pred)) - rs / max(1, nposr)) <= 1e-5 * rs / max(1, nposr)
    # layers
    feat = np.zeros((2, 5, 7, 256), np.float32)
    a = M.layers.Anchors(size=64, stride=16, ratios=M.anchors.AnchorParameters_default.ratios, scales=M.anchors.AnchorParameters_default.scales)(feat)
    want = (R.shifted((5, 7), 16, R.base_anchors(64).astype(np.float32).astype(np.float64))).astype(np.float32)
    # the float32 in-graph anchors (model/layers.py:42-53 + model/utils.py:51-80): both batch rows, bit for bit
    want32 = R.anchors_f32((5 * 16, 7 * 16, 3), sizes=[64], strides=[16], levels=[4])
    assert a.shape == (2, 5 * 7 * 9, 4) and a.dtype == np.float32
    assert np.array_equal(a[0], want32) and np.array_equal(a[1], want32)
    assert np.allclose(a[1], want, atol=1e-4)
    # utils.shift is the same graph code as a free function; utils.bbox_transform_inv the arithmetic of RegressBoxes
    base32 = R.base_anchors(64).astype(np.float32)
    assert np.array_equal(M.utils.shift((5, 7), 16, base32), want32)
    reg = rng.normal(size=(2, 315, 4)).astype(np.float32)
    boxes = M.layers.RegressBoxes()([a, reg])
    assert np.array_equal(M.utils.bbox_transform_inv(a, reg), boxes)
    with pytest.raises(ValueError):
        M.utils.bbox_transform_inv(a, reg[:, :10])
    clipped = M.layers.ClipBoxes()([np.zeros((2, 40, 56, 3), np.float32), boxes])
    assert np.array_equal(clipped, R.decode_boxes_f32(a, reg, (40, 56)))
    cls = rng.uniform(0, 1, size=(2, 315, 1)).astype(np.float32)
    fb, fsc, fl = M.layers.FilterDetections()([clipped, cls])
    for b in range(2):
        wb, ws, wl = R.filter_detections(clipped[b], cls[b])
        assert np.array_equal(fb[b], wb) and np.array_equal(fsc[b], ws) and np.array_equal(fl[b], wl)
    one = M.layers.filter_detections(clipped[0], cls[0])
    assert np.array_equal(one[1], fsc[0])
    src, tgt = rng.normal(size=(1, 25, 42, 8)).astype(np.float32), np.zeros((1, 50, 84, 8), np.float32)
    up = M.layers.UpsampleLike()([src, tgt])
    ys = np.minimum(np.floor(np.arange(50, dtype=np.float32) * np.float32(0.5)).astype(int), 24)
    xs = np.minimum(np.floor(np.arange(84, dtype=np.float32) * np.float32(0.5)).astype(int), 41)
    assert np.array_equal(up, src[:, ys][:, :, xs])
    cfgd = M.layers.FilterDetections(max_detections=100).get_config()
    assert cfgd["max_detections"] == 100 and cfgd["nms_threshold"] == 0.5
    # utils
    px = np.arange(256, dtype=np.uint8)
    assert np.array_equal(M.utils.preprocess_image(px, mode="custom_tf"), R.preprocess_custom_tf(px))
    assert M.utils.compute_resize_scale((2200, 1712, 3)) == 0.4672897196261682
    assert M.initializers.PriorProbability(0.01)((3,), dtype="float32")[0] == np.float32(R.prior_bias())
    assert M.Parameters.image_min_side == 800 and M.Parameters.class_mapping == {"table": 0}


def test_backbone_model_surface(M, tmp_path):
    D = M.defineModel
    with pytest.raises(ValueError):
        D.ResNetBackbone("vgg16")
    bb = D.ResNetBackbone("resnet50")
    assert set(bb.custom_objects) >= {"UpsampleLike", "PriorProbability", "RegressBoxes", "FilterDetections", "Anchors", "ClipBoxes"}
    model = bb.retinanet(1, num_anchors=None, modifier=None)
    assert model.output_names == ["regression", "classification"]
    M.utils.check_training_model(model)
    pred_model = D.retinanet_bbox(model=model)
    with pytest.raises(AssertionError):
        M.utils.assert_training_model(pred_model)
    model.compile(loss={"regression": M.losses.smooth_l1(), "classification": M.losses.focal()}, optimizer=D.Adam(lr=1e-4, clipnorm=0.001))
    x = R.preprocess_custom_tf(np.random.RandomState(0).randint(0, 255, size=(1, 128, 192, 3)).astype(np.uint8))
    reg, cls = model.predict_on_batch(x)
    N = R.anchors_for_shape((128, 192, 3)).shape[0]
    assert reg.shape == (1, N, 4) and cls.shape == (1, N, 1)
    boxes, scores, labels = pred_model.predict_on_batch(x)
    assert boxes.shape == (1, 300, 4) and scores.shape == (1, 300) and labels.dtype == np.int32
    # one training step through the Keras-style entry, then save / load round trip
    anchors = M.anchors.anchors_for_shape((128, 192, 3))
    ann = [{"bboxes": np.array([[20.0, 30.0, 110.0, 100.0]]), "labels": np.zeros(1)}]
    rb, lb = M.anchors.anchor_targets_bbox(anchors, [np.zeros((128, 192, 3))], ann, 1)
    losses = model.train_on_batch(x, [rb, lb])
    assert len(losses) == 3 and np.isfinite(losses).all() and abs(losses[0] - losses[1] - losses[2]) < 1e-6
    path = str(tmp_path / "w.npz")
    model.save(path)
    m2 = bb.retinanet(1)
    m2.load_weights(path, by_name=True, skip_mismatch=True)
    r1, _ = model.predict_on_batch(x)
    r2, _ = m2.predict_on_batch(x)
    assert np.abs(r1 - r2).max() < 2e-2          # same weights up to the bf16 re-emission of the trained master copy


def test_preprocessing_kernels(M):
    z = np.load(os.path.join(ROOT, "tests", "golden", "sample_page_crop.npz"))
    gray = z["orig_gray"]
    out, binary = M.preprocess.preprocess_pages(gray, return_binary=True)
    want, wbin = P.preprocess_page(gray)
    # byte work: bit-exact.  Device and oracle accumulate the 11 + 11 float32 taps in the same order without contraction,
    # with taps from libm's exp summed in tap order on both sides (oracle/ref_preprocess.py:gaussian_kernel)
    flips = int((binary != wbin).sum())
    assert flips == 0, "adaptive threshold differs from the oracle on %d of %d pixels" % (flips, binary.size)
    assert np.array_equal(out, want)
    # the distance transforms alone are integer arithmetic: bit-exact on the oracle's binary image, batch of 2, BGR input path
    L = M.anchors._rt.L
    h = M.anchors._rt.handle()
    b2 = np.stack([wbin, np.roll(wbin, 37, axis=1)])
    bd = torch.as_tensor(b2).cuda()
    dst = torch.empty(2, *wbin.shape, 3, dtype=torch.uint8, device="cuda")
    ws = torch.empty(2 * wbin.size * 12, dtype=torch.uint8, device="cuda")
    h.check(L.lib.rtn_distance_transform3(h.raw, bd.data_ptr(), 2, wbin.shape[0], wbin.shape[1], dst.data_ptr(), ws.data_ptr(), ws.numel()))
    torch.cuda.synchronize()
    for i in range(2):
        w3 = np.stack([P.to_u8(P.distance_transform(b2[i], m)) for m in ("L2", "L1", "C")], axis=-1)
        assert np.array_equal(dst[i].cpu().numpy(), w3)
    bgr = np.stack([gray] * 3, axis=-1)
    assert np.array_equal(M.preprocess.preprocess_pages(bgr), out)            # BGR2GRAY is the identity on B=G=R
    # resize (model/utils.py:140-154)
    img = R.preprocess_custom_tf(want)
    got, scale = M.utils.resize_image(img, min_side=300, max_side=500)
    ref = P.resize_cubic(img, scale)
    assert got.shape == ref.shape and np.abs(got - ref).max() <= 1e-5
    canvas, scales = M.preprocess.compute_inputs_device([want, want[:400, :500]], min_side=300, max_side=500, dtype=torch.float32)
    assert canvas.shape[0] == 2 and abs(scales[0] - scale) < 1e-15
    c0 = canvas[0].cpu().numpy()
    assert np.abs(c0[:ref.shape[0], :ref.shape[1]] - ref).max() <= 1e-5
    r1 = P.resize_cubic(R.preprocess_custom_tf(want[:400, :500]), scales[1])
    c1 = canvas[1].cpu().numpy()
    assert np.abs(c1[:r1.shape[0], :r1.shape[1]] - r1).max() <= 1e-5 and np.all(c1[r1.shape[0]:] == 0) and np.all(c1[:, r1.shape[1]:] == 0)


def test_config0_test_image_flow_on_a_sample_sized_page(M, tmp_path):
    """BASELINE.json configs[0]: RetinaNet.py test() on ONE processed 3-channel page, batch 1 (RetinaNet.py:306-402), through the
    mirror surface: preprocess_image('custom_tf') -> resize_image -> predict_on_batch(image[None]) -> boxes /= scale -> draw / crop /
    write.  The reference's page is 2200x1712 (scale 0.46729 -> 1028x800, 155,331 anchors); a synthetic page of that size stands in
    (the reference's file cannot travel to the GPU box).  Two weight sets: with the prior-probability bias no
    score reaches the 0.6 drawing threshold and the "noDete" output is written; a zero bias gives detections, checked in fp32 against the float64 oracle at the
    1e-3 px bar and written out as crops."""
    from oracle.ref_net import RefNet
    D, U = M.defineModel, M.utils
    rng = np.random.RandomState(0)
    base = np.clip(rng.exponential(12.0, (2200 // 8, 1712 // 8, 3)) * 6, 0, 255)
    page = np.kron(base, np.ones((8, 8, 1))).astype(np.uint8)                    # smooth, heavy-tailed like the distance maps
    image = U.preprocess_image(page, mode="custom_tf")
    image, scale = U.resize_image(image)
    assert image.shape == (1028, 800, 3) and abs(scale - 0.4672897196261682) < 1e-15
    Wt = M.anchors._rt.weights
    for cls_bias, expect_boxes in ((None, False), (0.0, True)):
        m = D.Model("resnet50", 1, 9, dtype="f32")
        m._state = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=cls_bias, tame=True)
        pm = D.retinanet_bbox(model=m)
        boxes, scores, labels = pm.predict_on_batch(np.expand_dims(image, axis=0))
        assert boxes.shape == (1, 300, 4) and scores.shape == (1, 300) and labels.shape == (1, 300)
        reg, cls = m.predict_on_batch(np.expand_dims(image, axis=0))
        assert reg.shape == (1, 155331, 4) and cls.shape == (1, 155331, 1)
        out = tmp_path / ("bias_%s" % cls_bias)
        draw = page.copy()
        kept = U.render_detections(image, draw, boxes.copy(), scores, labels, scale, str(out), "sample_0717_023.jpg", score_threshold=0.6 if not expect_boxes else 0.5)
        names = sorted(os.listdir(out / "detections_cropped"))
        assert os.path.exists(out / "detections_inImage" / "sample_0717_023.jpg")
        if not expect_boxes:
            # nothing reaches the 0.6 the reference draws at (RetinaNet.py:373): one "noDete" file carrying the best score
            assert np.all(scores < 0.6) and kept == [] and len(names) == 1 and "noDete_minScore-_%s" % scores[0, 0] in names[0]
            continue
        oreg, ocls = RefNet(m._state, dtype=torch.float64).forward(image[None])
        a32 = R.anchors_f32((1028, 800, 3))
        dbox = np.abs(R.decode_boxes_f32(a32, reg[0], (1028, 800)).astype(np.float64) -
                      R.decode_boxes_f32(a32, oreg.numpy()[0].astype(np.float32), (1028, 800))).max()
        dcls = np.abs(cls - ocls.numpy()).max()
        print("config 0 page, fp32 path vs float64 oracle: box %.3e px, score %.3e" % (dbox, dcls))
        assert dbox <= 1e-3 and dcls <= 1e-5
        wb, ws, wl = R.filter_detections(R.decode_boxes_f32(a32, reg[0], (1028, 800)), cls[0])
        assert np.array_equal(boxes[0], wb) and np.array_equal(scores[0], ws) and np.array_equal(labels[0], wl)
        n = int((scores[0] >= 0.5).sum())
        assert n > 0 and len(kept) == n and names == sorted("sample_0717_023_%d.jpg" % k for k in range(n))
        assert all(np.array_equal(k[0], (boxes[0, i] / scale).astype(int)) for i, k in enumerate(kept))


def test_predict_generator_pipelines_batches_and_matches_predict_on_batch(M):
    """keras.Model.predict_generator on the inference model with two batches in flight (Engine.in_flight): five batches of pages, of
    two different canvases, must give exactly what predict_on_batch gives batch by batch (model/defineModel.py:296-353)."""
    D = M.defineModel
    model = D.ResNetBackbone("resnet50").retinanet(1, num_anchors=None, modifier=None)
    bbox = D.retinanet_bbox(model=model)
    rng = np.random.RandomState(3)
    batches = [rng.randint(0, 256, (2, 128 + 32 * (i % 2), 160, 3)).astype(np.uint8) for i in range(5)]
    want = [bbox.predict_on_batch(b) for b in batches]
    got = bbox.predict_generator(batches, in_flight=2)
    for k in range(3):
        assert np.array_equal(got[k], np.concatenate([w[k] for w in want], axis=0))
    one = bbox.predict_generator(batches[:2], in_flight=1)
    for k in range(3):
        assert np.array_equal(one[k], np.concatenate([w[k] for w in want[:2]], axis=0))
