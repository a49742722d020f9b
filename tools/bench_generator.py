"""Measure the data-generator row: one batch of 16 sample-sized pages (2200x1712x3 uint8) through
[warp] -> normalise + bicubic resize into the padded canvas -> anchor targets, on the device, against the oracle on the host.

  python tools/bench_generator.py [--batch 16] [--iters 20] [--augment] [--cpu-pages 1]

Prints pages/s (a) with the uint8 pages already resident in HBM, (b) including the host->device copy of the pages, and the CPU
oracle's pages/s on `--cpu-pages` pages; plus the algorithmic HBM bytes per page and the GB/s they imply for (a).
"""
import argparse
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CG = importlib.import_module("retinanet-for-table-detection_amd.csv_generator")
T = importlib.import_module("retinanet-for-table-detection_amd.model.transform")

TRAIN_KW = dict(min_rotation=-0.1, max_rotation=0.1, min_translation=(-0.1, -0.1), max_translation=(0.1, 0.1), min_shear=-0.1,
                max_shear=0.1, min_scaling=(0.9, 0.9), max_scaling=(1.1, 1.1), flip_x_chance=0.5, flip_y_chance=0.5)


class MemoryGenerator(CG.Generator):
    """Pages and boxes held in memory (host arrays or device tensors): isolates the batch work from file decoding."""

    def __init__(self, pages, boxes, **kw):
        self.pages, self.boxes = pages, boxes
        super().__init__(**kw)

    def size(self):
        return len(self.pages)

    def num_classes(self):
        return 1

    def image_aspect_ratio(self, i):
        return self.pages[i].shape[1] / self.pages[i].shape[0]

    def load_image(self, i):
        return self.pages[i]

    def load_annotations(self, i):
        return {"labels": np.zeros(len(self.boxes[i])), "bboxes": self.boxes[i].copy()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--augment", action="store_true")
    ap.add_argument("--cpu-pages", type=int, default=1)
    a = ap.parse_args()
    rng = np.random.RandomState(0)
    H, W = 2200, 1712
    pages, boxes = [], []
    for i in range(a.batch):
        base = np.clip(rng.exponential(12.0, (H // 8, W // 8, 3)) * 6, 0, 255)
        pages.append(np.kron(base, np.ones((8, 8, 1))).astype(np.uint8))
        g = rng.randint(1, 7)
        w, h = rng.uniform(200, 1400, g), rng.uniform(150, 1200, g)
        x1, y1 = rng.uniform(0, W - w), rng.uniform(0, H - h)
        boxes.append(np.stack([x1, y1, x1 + w, y1 + h], 1))
    kw = dict(batch_size=a.batch, group_method="none", shuffle_groups=False, dtype=torch.bfloat16)
    if a.augment:
        kw.update(transform_generator=T.random_transform_generator(prng=np.random.RandomState(1), **TRAIN_KW),
                  transform_parameters=T.TransformParameters())
    res = {}
    for label, src in (("resident", [torch.as_tensor(p).cuda() for p in pages]), ("with_h2d", pages)):
        gen = MemoryGenerator(src, boxes, **kw)
        for _ in range(3):
            gen[0]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            x, (reg, lab) = gen[0]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.iters
        res[label] = a.batch / dt
        print("%-9s %.2f ms per batch of %d -> %.0f pages/s   canvas %s, anchors %d" % (label, dt * 1e3, a.batch, a.batch / dt, tuple(x.shape), reg.shape[1]))
        gen.close()
    Ho, Wo, N = x.shape[1], x.shape[2], reg.shape[1]
    per_page = H * W * 3 * (3 if a.augment else 1) + Ho * Wo * 3 * 2 + N * (5 + 2) * 4
    print("algorithmic HBM bytes per page: %.1f MB -> %.0f GB/s with the pages resident" % (per_page / 1e6, per_page * res["resident"] / 1e9))
    if a.cpu_pages > 0:
        from oracle import ref_generator as G
        n = a.cpu_pages
        prng = np.random.RandomState(1)
        tfs = [G.random_transform(prng, **TRAIN_KW) for _ in range(n)] if a.augment else None
        t0 = time.perf_counter()
        G.compute_input_output(pages[:n], boxes[:n], [np.zeros(len(b)) for b in boxes[:n]], 1, transforms=tfs)
        dt = time.perf_counter() - t0
        print("cpu oracle (NumPy, 1 thread): %.2f s for %d page(s) -> %.2f pages/s" % (dt, n, n / dt))


if __name__ == "__main__":
    main()
