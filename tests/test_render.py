"""CPU: result rendering (model/utils.py:267-373, RetinaNet.py:366-402).  Host raster work; parity with
OpenCV's anti-aliased drawing is unpinned, so geometry, colours, crops and file names are what is checked."""
import importlib
import os

import numpy as np

U = importlib.import_module("retinanet-for-table-detection_amd.model.utils")


def test_label_color_matches_hsv_wheel():
    assert U.label_color(0) == [255, 0, 0] and len(U.label_color(79)) == 3
    try:
        import matplotlib.colors as mc
    except Exception:
        return
    want = [list((mc.hsv_to_rgb([x, 1.0, 1.0]) * 255).astype(int)) for x in np.arange(0, 1, 1.0 / 80)]
    assert all(U.label_color(i) == [int(v) for v in want[i]] for i in range(80))
    assert tuple(U.label_color(99)) == (0, 255, 0)


def test_draw_box_extract_box_and_caption():
    img = np.full((100, 120, 3), 200, np.uint8)
    U.draw_box(img, [20.7, 30.2, 90.9, 70.1], color=[255, 0, 0], thickness=5)
    assert np.all(img[30, 20:91] == 0) and np.all(img[70, 20:91] == 0) and np.all(img[30:71, 20] == 0) and np.all(img[30:71, 90] == 0)
    assert np.all(img[28:33, 18:93] == 0) and np.all(img[50, 25:88] == 200) and np.all(img[27, :] == 200) and np.all(img[:, 17] == 200)
    crop = U.extract_box(img, [20.7, 30.2, 90.9, 70.1])
    assert crop.shape == (40, 70, 3)
    U.draw_box(img, [-10, -10, 300, 300], color=None)                       # clipped, no exception
    canvas = np.full((80, 300, 3), 255, np.uint8)
    U.draw_caption(canvas, [10, 70, 100, 79], "table 0.912")
    red = (canvas == (0, 0, 255)).all(axis=2)
    assert red.sum() > 50 and red[60:].sum() == 0                            # text sits above y1 - 10
    U.draw_detections(canvas, np.array([[5, 5, 50, 50.0]]), np.array([0.9]), np.array([0]), label_to_name=lambda l: "table")
    U.draw_annotations(canvas, np.array([[5, 5, 50, 50.0, 0]]))
    U.draw_boxes(canvas, np.array([[1, 1, 10, 10]]), color=(0, 0, 0))


def test_render_detections_writes_the_reference_outputs(tmp_path):
    from PIL import Image
    page = np.full((400, 300, 3), 230, np.uint8)
    boxes = np.full((1, 300, 4), -1, np.float32)
    scores = np.full((1, 300), -1, np.float32)
    labels = np.full((1, 300), -1, np.int32)
    boxes[0, 0], scores[0, 0], labels[0, 0] = [20, 40, 120, 160], 0.93, 0
    boxes[0, 1], scores[0, 1], labels[0, 1] = [30, 100, 140, 190], 0.71, 0
    boxes[0, 2], scores[0, 2], labels[0, 2] = [5, 5, 10, 10], 0.41, 0
    kept = U.render_detections(None, page, boxes, scores, labels, 0.5, str(tmp_path), "page_7.png")
    assert [k[1:] for k in kept] == [(np.float32(0.93), 0), (np.float32(0.71), 0)] or len(kept) == 2
    assert list(kept[0][0]) == [40, 80, 240, 320]                            # boxes / image_scale
    out = sorted(os.listdir(tmp_path / "detections_cropped"))
    assert out == ["page_7_0.png", "page_7_1.png"]
    assert Image.open(tmp_path / "detections_cropped" / "page_7_0.png").size == (200, 240)
    assert Image.open(tmp_path / "detections_inImage" / "page_7.png").size == (300, 400)
    # nothing above the threshold: the page itself goes to detections_cropped under the reference's "noDete" name
    scores[0, 0] = 0.3
    d2 = tmp_path / "second"
    kept = U.render_detections(None, page.copy(), boxes, scores, labels, 1.0, str(d2), "page_8.jpg")
    names = os.listdir(d2 / "detections_cropped")
    assert kept == [] and len(names) == 1 and names[0].startswith("page_8_noDete_minScore-_0.3") and names[0].endswith(".jpg")
