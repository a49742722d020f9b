"""Shared helpers for the test modules (tests/ is on sys.path under pytest's default import mode)."""


def load_case(golden, name):
    """Rebuild a target-assignment case (canvas, per-image shapes, per-image GT arrays) from the golden file."""
    canvas = tuple(int(v) for v in golden["tgt_%s_canvas" % name])
    shapes = [tuple(int(v) for v in s) for s in golden["tgt_%s_shapes" % name]]
    counts = [int(c) for c in golden["tgt_%s_gtcount" % name]]
    flat = golden["tgt_%s_gt" % name]
    gts, o = [], 0
    for c in counts:
        gts.append(flat[o:o + c])
        o += c
    return canvas, shapes, gts
