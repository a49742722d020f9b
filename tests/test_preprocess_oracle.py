"""CPU: the preprocessing oracle (oracle/ref_preprocess.py, a restatement of OpenCV's algorithms) against the only artefact of
the reference that pins it — a crop of data/orig/sample_0717_023_orig.jpg and of data/processed/sample_0717_023.jpg
(tests/golden/sample_page_crop.npz; both files are lossy JPEG, so agreement is at PSNR level, not bit level)."""
import os

import numpy as np

from oracle import ref_preprocess as P

HERE = os.path.dirname(os.path.abspath(__file__))


def test_vectorised_distance_transform_equals_literal_two_pass():
    rng = np.random.RandomState(0)
    for shape, p in (((37, 53), 0.08), ((20, 70), 0.01), ((64, 17), 0.5)):
        b = (rng.uniform(size=shape) > p).astype(np.uint8) * 255
        for m in ("L2", "L1", "C"):
            assert np.array_equal(P.distance_transform(b, m), P.distance_transform_literal(b, m)), (shape, m)
    # known answers: single zero pixel -> the metric's closed form
    b = np.full((9, 9), 255, np.uint8)
    b[4, 4] = 0
    assert P.distance_transform(b, "L1")[0, 0] == 8 and P.distance_transform(b, "C")[0, 0] == 4
    d = P.distance_transform(b, "L2")
    assert d[4, 7] == 3 and abs(d[3, 2] - 2.1969) < 2e-5 and abs(d[1, 1] - 3 * 1.4) < 1e-4


def test_oracle_reproduces_reference_sample_page():
    z = np.load(os.path.join(HERE, "golden", "sample_page_crop.npz"))
    out, binary = P.preprocess_page(z["orig_gray"])
    ours_rgb = out[..., ::-1].astype(np.float64)              # stored file is RGB = (C, L1, L2)
    ref = z["processed_rgb"].astype(np.float64)
    H, W = binary.shape
    yy, xx = np.mgrid[0:H, 0:W]
    border = np.minimum(np.minimum(yy, H - 1 - yy), np.minimum(xx, W - 1 - xx))
    inside = out[..., 1].astype(np.int64) < border           # nearest ink provably inside the crop (L1 >= L2 >= C)
    assert inside.mean() > 0.7
    for ci in range(3):
        d = (ours_rgb[..., ci] - ref[..., ci])[inside]
        psnr = 10 * np.log10(255.0 ** 2 / (d ** 2).mean())
        cc = np.corrcoef(ours_rgb[..., ci][inside], ref[..., ci][inside])[0, 1]
        assert psnr > 44.0 and cc > 0.999, (ci, psnr, cc)


def test_resize_cubic_properties():
    rng = np.random.RandomState(1)
    img = rng.uniform(-1, 1, size=(40, 30, 3)).astype(np.float32)
    assert P.resize_cubic(img, 0.4672897196261682).shape == (19, 14, 3)
    const = np.full((12, 9, 3), 0.25, np.float32)
    assert np.allclose(P.resize_cubic(const, 1.7), 0.25, atol=1e-6)           # partition of unity
    # the taps are Keys' cubic convolution kernel with a = -0.75 (OpenCV's INTER_CUBIC), evaluated independently in float64
    def keys(t, a=-0.75):
        t = abs(t)
        return (a + 2) * t ** 3 - (a + 3) * t ** 2 + 1 if t <= 1 else (a * t ** 3 - 5 * a * t ** 2 + 8 * a * t - 4 * a if t < 2 else 0.0)
    sig = rng.uniform(-1, 1, size=32)
    up = P.resize_cubic(np.tile(sig.astype(np.float32)[None, :, None], (8, 1, 1)), 2.0)[3, :, 0]
    for dx in range(8, 56):
        f = (dx + 0.5) / 2.0 - 0.5
        s0 = int(np.floor(f))
        want = sum(keys(f - (s0 - 1 + k)) * np.float32(sig[s0 - 1 + k]) for k in range(4))
        assert abs(up[dx] - want) < 1e-5
