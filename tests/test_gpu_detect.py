"""GPU parity: rtn_decode_filter_nms against the oracle's restatement of Anchors -> RegressBoxes -> ClipBoxes ->
filter_detections (model/layers.py:177-264).  Boxes/scores/labels must be BIT-EXACT: both sides evaluate the same
float32 expressions op by op (FP contraction is off in the kernel), and selection order is fully determined."""
import ctypes as C
import importlib

import numpy as np
import pytest
import torch

from oracle import ref_numpy as R

pytestmark = pytest.mark.gpu
DEV = "cuda"


def run_detect(pkg, handle, canvas, regression, classification, thr=0.05, iou=0.5, max_det=300):
    E = importlib.import_module(pkg.__name__ + ".engine")
    cfg, N = E.make_anchor_cfg(canvas)
    B, _, K = classification.shape
    assert regression.shape == (B, N, 4)
    reg = torch.as_tensor(regression).to(DEV)
    cls = torch.as_tensor(classification).to(DEV)
    wsb = pkg.lib.rtn_detect_workspace_bytes(B, N, K)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    boxes = torch.full((B, max_det, 4), 7.0, dtype=torch.float32, device=DEV)
    scores = torch.full((B, max_det), 7.0, dtype=torch.float32, device=DEV)
    labels = torch.full((B, max_det), 7, dtype=torch.int32, device=DEV)
    handle.check(pkg.lib.rtn_decode_filter_nms(handle.raw, C.byref(cfg), B, K, reg.data_ptr(), cls.data_ptr(), canvas[0], canvas[1],
                                               thr, iou, max_det, boxes.data_ptr(), scores.data_ptr(), labels.data_ptr(),
                                               ws.data_ptr(), wsb))
    torch.cuda.synchronize()
    return boxes.cpu().numpy(), scores.cpu().numpy(), labels.cpu().numpy()


def oracle_detect(canvas, regression, classification, thr=0.05, iou=0.5, max_det=300):
    a32 = R.anchors_f32(canvas + (3,))
    out = []
    for b in range(regression.shape[0]):
        boxes = R.decode_boxes_f32(a32, regression[b], canvas)
        out.append(R.filter_detections(boxes, classification[b], thr, max_det, iou))
    return [np.stack([o[i] for o in out]) for i in range(3)]


def compare(got, want):
    gb, gs, gl = got
    wb, ws, wl = want
    assert np.array_equal(gl, wl), "labels differ"
    assert np.array_equal(gs, ws), "scores differ"
    assert np.array_equal(gb, wb), "boxes differ"


def synth(canvas, B, K, seed, frac_above, score_hi=0.99, reg_scale=0.5, quantise=None):
    rng = np.random.RandomState(seed)
    N = R.anchors_for_shape(canvas + (3,)).shape[0]
    reg = (rng.normal(size=(B, N, 4)) * reg_scale).astype(np.float32)
    cls = rng.uniform(0.0, 0.05, size=(B, N, K)).astype(np.float32)
    hot = rng.uniform(size=(B, N, K)) < frac_above
    vals = rng.uniform(0.051, score_hi, size=(B, N, K)).astype(np.float32)
    if quantise:
        vals = (np.round(vals * quantise) / quantise).astype(np.float32)      # many exact ties
    cls[hot] = vals[hot]
    return reg, cls


def test_sparse_candidates(pkg, handle):
    canvas = (256, 384)
    reg, cls = synth(canvas, 3, 1, 0, 0.01)
    compare(run_detect(pkg, handle, canvas, reg, cls), oracle_detect(canvas, reg, cls))


def test_ties_and_more_than_300_survivors(pkg, handle):
    canvas = (512, 768)
    reg, cls = synth(canvas, 2, 1, 1, 0.08, quantise=64, reg_scale=0.2)
    got = run_detect(pkg, handle, canvas, reg, cls)
    want = oracle_detect(canvas, reg, cls)
    assert np.all(want[1][:, -1] > 0)            # the case really fills all 300 slots
    compare(got, want)


def test_no_candidates_and_single_candidate(pkg, handle):
    canvas = (128, 192)
    reg, cls = synth(canvas, 2, 1, 2, 0.0)
    cls[1, 1234, 0] = 0.7
    gb, gs, gl = run_detect(pkg, handle, canvas, reg, cls)
    assert np.all(gb[0] == -1) and np.all(gs[0] == -1) and np.all(gl[0] == -1)
    compare((gb, gs, gl), oracle_detect(canvas, reg, cls))
    assert gs[1, 0] == np.float32(0.7) and gl[1, 0] == 0 and np.all(gs[1, 1:] == -1)


def test_threshold_is_strict_and_boxes_clip(pkg, handle):
    canvas = (128, 192)
    reg, cls = synth(canvas, 1, 1, 3, 0.0)
    cls[0, 10, 0] = np.float32(0.05)             # == threshold: NOT a candidate (model/layers.py:202 uses >)
    cls[0, 11, 0] = np.nextafter(np.float32(0.05), np.float32(1))
    reg[0, 11] = [-50, -50, 50, 50]              # decodes far outside: clipped to the canvas, inclusive of W/H
    gb, gs, gl = run_detect(pkg, handle, canvas, reg, cls)
    assert gs[0, 0] == cls[0, 11, 0] and gs[0, 1] == -1
    assert gb[0, 0, 0] == 0 and gb[0, 0, 1] == 0 and gb[0, 0, 2] == 192 and gb[0, 0, 3] == 128
    compare((gb, gs, gl), oracle_detect(canvas, reg, cls))


def test_many_candidates_forces_batched_selection(pkg, handle):
    """Every anchor is a candidate (the untrained-network case with the classification bias at 0): far more than
    one 4096-candidate batch, so the radix-select + multi-batch path runs; result must still be exact."""
    canvas = (320, 480)
    rng = np.random.RandomState(4)
    N = R.anchors_for_shape(canvas + (3,)).shape[0]
    assert N > 3 * 4096
    reg = (rng.normal(size=(2, N, 4)) * 0.3).astype(np.float32)
    cls = rng.uniform(0.3, 0.7, size=(2, N, 1)).astype(np.float32)
    cls[1] = (np.round(cls[1] * 32) / 32).astype(np.float32)       # image 1: heavy ties across batch boundaries
    compare(run_detect(pkg, handle, canvas, reg, cls), oracle_detect(canvas, reg, cls))


def test_multiclass_merge_topk(pkg, handle):
    canvas = (256, 384)
    reg, cls = synth(canvas, 2, 3, 5, 0.02, quantise=128)
    got = run_detect(pkg, handle, canvas, reg, cls)
    want = oracle_detect(canvas, reg, cls)
    assert len(set(want[2][0][want[2][0] >= 0].tolist())) == 3
    compare(got, want)


def test_full_canvas_batch8_properties(pkg, handle):
    """BASELINE config size (8 x 200,700 anchors): size-independent properties instead of the O(N*300) oracle loop."""
    canvas = (800, 1333)
    reg, cls = synth(canvas, 8, 1, 6, 0.004)
    gb, gs, gl = run_detect(pkg, handle, canvas, reg, cls)
    a32 = R.anchors_f32(canvas + (3,))
    for b in range(8):
        n = int(np.sum(gs[b] >= 0))
        assert n > 0 and np.all(gs[b, :n] > 0.05) and np.all(gs[b, n:] == -1) and np.all(gl[b, n:] == -1)
        assert np.all(gs[b, :n - 1] >= gs[b, 1:n])                          # sorted
        assert np.all(gb[b, :n, 0] >= 0) and np.all(gb[b, :n, 2] <= 1333) and np.all(gb[b, :n, 3] <= 800)
        for i in range(1, n):                                               # no kept pair overlaps > 0.5
            assert not np.any(R._iou_f32(gb[b, i], gb[b, :i]) > np.float32(0.5))
        # every output row is the decode of some candidate with exactly that score
        boxes = R.decode_boxes_f32(a32, reg[b], canvas)
        for i in range(0, n, 17):
            idx = np.nonzero(cls[b, :, 0] == gs[b, i])[0]
            assert any(np.array_equal(boxes[j], gb[b, i]) for j in idx)
    # one image also against the full oracle
    want = oracle_detect(canvas, reg[:1], cls[:1])
    compare((gb[:1], gs[:1], gl[:1]), want)
