"""Detection evaluator (AP at an IoU threshold) and the detection dump format — SURVEY §8(f) rank 1.

The reference has no evaluator (its training loop runs without validation: RetinaNet.py:149; test() only draws the
score-sorted boxes: RetinaNet.py:373-383), so there is nothing of the reference's to pin against: PARITY UNPINNED.  The
algorithm is the Pascal-VOC style evaluation published with keras-retinanet, the project the reference's model/ package
derives from:
  per class, over all images: detections sorted by descending score; a detection is a true positive when its best-IoU
  annotation (same image, same class) has IoU >= threshold and has not been matched before, otherwise a false positive;
  recall = cumTP / #annotations, precision = cumTP / (cumTP + cumFP); AP = area under the monotone precision envelope
  evaluated at every recall change (all-point interpolation).
IoU comes from the device (`rtn_compute_overlap`, the float32 IoU of model/utils.py:180-211): no GPU, no evaluator.
"""
import csv

import numpy as np

from .utils import compute_overlap


def compute_ap(recall, precision):
    """Area under the precision envelope; recall/precision are the cumulative curves in detection order."""
    mrec = np.concatenate(([0.], np.asarray(recall, np.float64), [1.]))
    mpre = np.concatenate(([0.], np.asarray(precision, np.float64), [0.]))
    for i in range(mpre.size - 1, 0, -1):
        mpre[i - 1] = np.maximum(mpre[i - 1], mpre[i])
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return float(np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1]))


def evaluate_detections(all_detections, all_annotations, num_classes=1, iou_threshold=0.5):
    """all_detections[i][c]: (n,5) array [x1,y1,x2,y2,score] of image i, class c; all_annotations[i][c]: (m,4) boxes.
    Returns {class: (average_precision, num_annotations)}."""
    result = {}
    for label in range(num_classes):
        scores, hits = [], []
        num_annotations = 0
        for dets, anns in zip(all_detections, all_annotations):
            d = np.asarray(dets[label], np.float64).reshape(-1, 5)
            a = np.asarray(anns[label], np.float64).reshape(-1, 4)
            num_annotations += a.shape[0]
            if d.shape[0] == 0:
                continue
            order = np.argsort(-d[:, 4], kind="stable")
            d = d[order]
            taken = np.zeros(a.shape[0], dtype=bool)
            iou = compute_overlap(d[:, :4], a) if a.shape[0] else None        # one device call per image
            for k in range(d.shape[0]):
                scores.append(d[k, 4])
                if a.shape[0] == 0:
                    hits.append(False)
                    continue
                j = int(np.argmax(iou[k]))
                if iou[k, j] >= iou_threshold and not taken[j]:
                    taken[j] = True
                    hits.append(True)
                else:
                    hits.append(False)
        if num_annotations == 0:
            result[label] = (0.0, 0)
            continue
        order = np.argsort(-np.asarray(scores, np.float64), kind="stable")
        tp = np.cumsum(np.asarray(hits, bool)[order]).astype(np.float64)
        fp = np.cumsum(~np.asarray(hits, bool)[order]).astype(np.float64)
        recall = tp / num_annotations
        precision = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
        result[label] = (compute_ap(recall, precision), num_annotations)
    return result


def mean_ap(result, weighted=False):
    """mAP over the classes that have annotations (optionally weighted by their annotation counts)."""
    present = [(ap, n) for ap, n in result.values() if n > 0]
    if not present:
        return 0.0
    if weighted:
        return float(sum(ap * n for ap, n in present) / sum(n for _, n in present))
    return float(sum(ap for ap, _ in present) / len(present))


def split_detections(boxes, scores, labels, num_classes=1, scale=1.0, score_threshold=0.05, max_detections=300):
    """One image's padded inference outputs ((300,4), (300,), (300,), -1 padded: model/defineModel.py:310-315) ->
    per-class (n,5) arrays in ORIGINAL image coordinates (boxes divided by the resize scale, RetinaNet.py:366-367)."""
    boxes, scores, labels = np.asarray(boxes, np.float64) / scale, np.asarray(scores, np.float64), np.asarray(labels)
    keep = np.where(scores > score_threshold)[0][:max_detections]
    out = []
    for c in range(num_classes):
        idx = keep[labels[keep] == c]
        out.append(np.concatenate([boxes[idx], scores[idx, None]], axis=1))
    return out


def write_detections_csv(path, image_ids, per_image_detections, class_names=None):
    """Detection dump: the reference's annotation CSV row (csv_generator.py:16-52: image_id,xmin,ymin,xmax,ymax,label)
    with the score appended."""
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        for image_id, dets in zip(image_ids, per_image_detections):
            for c, d in enumerate(dets):
                name = class_names[c] if class_names else c
                for x1, y1, x2, y2, s in np.asarray(d, np.float64).reshape(-1, 5):
                    w.writerow([image_id, "%.2f" % x1, "%.2f" % y1, "%.2f" % x2, "%.2f" % y2, name, "%.6f" % s])


def read_detections_csv(path, class_ids=None):
    """Inverse of write_detections_csv -> {image_id: {class: (n,5) array}}."""
    out = {}
    with open(path, newline="") as f:
        for image_id, x1, y1, x2, y2, label, score in csv.reader(f):
            c = class_ids[label] if class_ids else int(label)
            out.setdefault(image_id, {}).setdefault(c, []).append([float(x1), float(y1), float(x2), float(y2), float(score)])
    return {k: {c: np.asarray(v, np.float64) for c, v in d.items()} for k, d in out.items()}


def evaluate(model, images, annotations, scales=None, num_classes=1, iou_threshold=0.5, score_threshold=0.05, max_detections=300):
    """Run an inference model (model/defineModel.py:retinanet_bbox) over preprocessed images [(H,W,3) ...] and score it.
    annotations[i]: (m,5) [x1,y1,x2,y2,label] in original coordinates; scales[i]: resize scale of image i."""
    all_dets, all_anns = [], []
    for i, img in enumerate(images):
        boxes, scores, labels = model.predict_on_batch(np.expand_dims(img, 0))[:3]
        s = 1.0 if scales is None else scales[i]
        all_dets.append(split_detections(boxes[0], scores[0], labels[0], num_classes, s, score_threshold, max_detections))
        ann = np.asarray(annotations[i], np.float64).reshape(-1, 5)
        all_anns.append([ann[ann[:, 4] == c, :4] for c in range(num_classes)])
    return evaluate_detections(all_dets, all_anns, num_classes, iou_threshold)
