"""GPU parity of the whole RetinaNet forward (stem -> ResNet-50 -> FPN -> heads -> decode/NMS) against the
torch-CPU oracle (oracle/ref_net.py; parity unpinned: no Keras/TF here, no checkpoint in the reference).

Tolerances (stated, measured on MI355X):
  fp32 path : every decoded box within 1e-3 px and every score within 1e-5 of the float64 oracle
              (BASELINE.json north_star: "boxes within 1e-3 of the Keras reference").
  bf16 path : compared with the oracle run in bf16 emulation (bf16 weights, bf16 activations between layers,
              fp32 accumulate) — boxes within 0.5 px, scores within 5e-3; against the float64 oracle the
              measured drift is printed (bf16 has 8 significand bits; boxes are O(10^2..10^3) px).
Post-processing is checked bit-exactly by feeding the engine's own head outputs through the oracle's
filter_detections."""
import importlib

import numpy as np
import pytest
import torch

from oracle import ref_numpy as R
from oracle.ref_net import RefNet

pytestmark = pytest.mark.gpu
CANVAS = (160, 224)


def mods(pkg):
    return importlib.import_module(pkg.__name__ + ".engine"), importlib.import_module(pkg.__name__ + ".weights")


def make_images(B, seed=0):
    g = torch.Generator().manual_seed(seed)
    # heavy-tailed distance-map-like pixels in [0,255] -> custom_tf normalisation
    raw = torch.clamp(torch.empty(B, CANVAS[0], CANVAS[1], 3).exponential_(1 / 12.0, generator=g) *
                      torch.rand(B, CANVAS[0], CANVAS[1], 3, generator=g), 0, 255).round()
    return raw.to(torch.uint8)


@pytest.fixture(scope="module")
def state(pkg):
    _, Wt = mods(pkg)
    return Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=0.0)


def decoded(reg, canvas):
    a32 = R.anchors_f32(canvas + (3,))
    return np.stack([R.decode_boxes_f32(a32, reg[b], canvas) for b in range(reg.shape[0])])


def test_fp32_network_boxes_within_1e3_px(pkg, state):
    E, _ = mods(pkg)
    img_u8 = make_images(2)
    x = torch.as_tensor(R.preprocess_custom_tf(img_u8.numpy()))
    eng = E.Engine("resnet50", 1, 9, dtype="f32")
    eng.load_state(state)
    reg, cls = eng.forward(x.cuda())
    torch.cuda.synchronize()
    reg, cls = reg.cpu().numpy(), cls.cpu().numpy()
    oreg, ocls = RefNet(state, dtype=torch.float64).forward(x.numpy())
    oreg, ocls = oreg.numpy(), ocls.numpy()
    assert reg.shape == oreg.shape and cls.shape == ocls.shape
    dbox = np.abs(decoded(reg, CANVAS).astype(np.float64) - decoded(oreg.astype(np.float32), CANVAS)).max()
    dcls = np.abs(cls - ocls).max()
    print("fp32 path: max |box diff| = %.3e px, max |score diff| = %.3e, max |regression diff| = %.3e" %
          (dbox, dcls, np.abs(reg - oreg).max()))
    assert dbox <= 1e-3 and dcls <= 1e-5
    # the uint8 entry (normalisation fused into the stem packer) gives the same bits as the float entry
    reg8, cls8 = eng.forward(img_u8.cuda())
    torch.cuda.synchronize()
    assert np.array_equal(reg8.cpu().numpy(), reg) and np.array_equal(cls8.cpu().numpy(), cls)
    # post-processing: bit-exact given the same head outputs
    boxes, scores, labels = eng.detect(x.cuda())
    torch.cuda.synchronize()
    for b in range(2):
        wb, ws, wl = R.filter_detections(decoded(reg, CANVAS)[b], cls[b])
        assert np.array_equal(boxes[b].cpu().numpy(), wb) and np.array_equal(scores[b].cpu().numpy(), ws)
        assert np.array_equal(labels[b].cpu().numpy(), wl)
    assert (scores[0] > 0).sum() > 0


def test_bf16_network_vs_bf16_emulating_oracle(pkg, state):
    E, _ = mods(pkg)
    img_u8 = make_images(2, seed=1)
    x = torch.as_tensor(R.preprocess_custom_tf(img_u8.numpy()))
    eng = E.Engine("resnet50", 1, 9, dtype="bf16")
    eng.load_state(state)
    reg, cls = eng.forward(x.cuda())
    torch.cuda.synchronize()
    reg, cls = reg.cpu().numpy(), cls.cpu().numpy()
    ereg, ecls = RefNet(state, dtype=torch.float32, emulate_bf16=True).forward(x.numpy())
    oreg, ocls = RefNet(state, dtype=torch.float64).forward(x.numpy())
    ereg, ecls, oreg, ocls = ereg.numpy(), ecls.numpy(), oreg.numpy(), ocls.numpy()
    db_e = np.abs(decoded(reg, CANVAS).astype(np.float64) - decoded(ereg, CANVAS)).max()
    ds_e = np.abs(cls - ecls).max()
    db_o = np.abs(decoded(reg, CANVAS).astype(np.float64) - decoded(oreg.astype(np.float32), CANVAS)).max()
    ds_o = np.abs(cls - ocls).max()
    print("bf16 path vs bf16-emulating oracle: box %.3e px, score %.3e; vs float64 oracle: box %.3e px, score %.3e" %
          (db_e, ds_e, db_o, ds_o))
    assert db_e <= 0.5 and ds_e <= 5e-3
    assert db_o <= 4.0 and ds_o <= 3e-2
    boxes, scores, labels = eng.detect(x.cuda())
    torch.cuda.synchronize()
    for b in range(2):
        wb, ws, wl = R.filter_detections(decoded(reg, CANVAS)[b], cls[b])
        assert np.array_equal(boxes[b].cpu().numpy(), wb) and np.array_equal(scores[b].cpu().numpy(), ws)


def test_default_init_gives_empty_detections(pkg):
    """Keras-default init: classification bias -log(99) => every score ~0.01 < 0.05 => all -1."""
    E, Wt = mods(pkg)
    eng = E.Engine("resnet50", 1, 9, dtype="bf16")
    eng.load_state(Wt.init_state("resnet50", 1, 9, seed=1))
    boxes, scores, labels = eng.detect(torch.as_tensor(R.preprocess_custom_tf(make_images(1).numpy())).cuda())
    torch.cuda.synchronize()
    assert torch.all(scores == -1) and torch.all(labels == -1) and torch.all(boxes == -1)


def test_backbone_validation(pkg):
    E, _ = mods(pkg)
    with pytest.raises(ValueError):
        E.Engine("vgg16", 1, 9)
