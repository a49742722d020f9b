"""GPU parity of one whole training step (forward -> focal + smooth-L1 -> backward through heads/FPN/ResNet-50 ->
global-norm clip + Adam) against torch-CPU float64 autograd of the restated graph (oracle/ref_net.py; parity unpinned).

fp32 path : per-layer weight gradients within 2e-3 of the layer's gradient scale (max |g|), losses within 1e-5 relative,
            updated weights within 1e-6 after the Adam step.
bf16 path : gradient direction — cosine similarity with the float64 gradient >= 0.98 per layer, norms within 10 %."""
import importlib

import numpy as np
import pytest
import torch

from oracle import ref_numpy as R
from oracle.ref_net import train_step_oracle, adam_clipnorm_oracle

pytestmark = pytest.mark.gpu
CANVAS = (128, 192)
LAYERS = ["conv1", "res2a_branch2a", "res2a_branch1", "res2b_branch2b", "res2c_branch2c", "res3a_branch2a", "res3a_branch1",
          "res3d_branch2b", "res4a_branch1", "res4f_branch2c", "res5a_branch2a", "res5c_branch2b", "C5_reduced", "P5", "C4_reduced",
          "P4", "C3_reduced", "P3", "P6", "P7", "pyramid_regression_0", "pyramid_regression_3", "pyramid_regression",
          "pyramid_classification_0", "pyramid_classification_2", "pyramid_classification"]


def mods(pkg):
    return [importlib.import_module(pkg.__name__ + "." + m) for m in ("engine", "weights", "trainer")]


def make_batch(B, seed):
    g = torch.Generator().manual_seed(seed)
    raw = torch.clamp(torch.empty(B, CANVAS[0], CANVAS[1], 3).exponential_(1 / 12.0, generator=g) *
                      torch.rand(B, CANVAS[0], CANVAS[1], 3, generator=g), 0, 255).round().to(torch.uint8)
    x = R.preprocess_custom_tf(raw.numpy())
    anchors = R.anchors_for_shape(CANVAS + (3,))
    rng = np.random.RandomState(seed)
    gts, shapes = [], []
    for _ in range(B):
        n = rng.randint(1, 4)
        w, h = rng.uniform(30, 120, n), rng.uniform(25, 90, n)
        x1, y1 = rng.uniform(0, CANVAS[1] - w), rng.uniform(0, CANVAS[0] - h)
        gts.append(np.stack([x1, y1, x1 + w, y1 + h], axis=1))
        shapes.append((CANVAS[0], int(rng.randint(150, CANVAS[1] + 1))))
    reg, lab = R.anchor_targets(anchors, shapes, gts, [np.zeros(len(g_)) for g_ in gts], 1)
    assert (lab[..., 1] == 1).sum() > 10
    return x, reg, lab


def unpack_grad(tr, Wt, name):
    """flat packed gradient of the folded weights -> gradient w.r.t. the Keras HWIO kernel (x fold scale)."""
    lo = tr.eng.layout[name]
    dW, db = tr.grad_views(name)
    gs = tr.gscale[lo["woff"]:lo["woff"] + lo["rows"] * lo["K"]].view(lo["rows"], lo["K"])
    g = (dW * gs).cpu().double()
    cout = lo["cout"]
    if name == "conv1":
        k = g[:cout].reshape(cout, 8, 8, 4)[:, :7, :7, :3].permute(1, 2, 3, 0)
    else:
        k = g[:cout].reshape(cout, lo["kh"], lo["kw"], lo["cin"]).permute(1, 2, 3, 0)
    return k, db[:cout].cpu().double()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_training_step(pkg, dtype):
    E, Wt, T = mods(pkg)
    state = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=-2.0, tame=True)
    x, reg_t, lab_t = make_batch(2, seed=11)
    (l_reg, l_cls), og = train_step_oracle(state, x, reg_t, lab_t)
    eng = E.Engine("resnet50", 1, 9, dtype=dtype)
    eng.load_state(state)
    tr = T.Trainer(eng, lr=1e-4, clipnorm=0.001)
    xd = torch.as_tensor(x).cuda()
    regd, labd = torch.as_tensor(reg_t).cuda(), torch.as_tensor(lab_t).cuda()
    sums = tr.forward_backward(xd, regd, labd)
    torch.cuda.synchronize()
    s = sums.cpu().numpy()
    got_reg, got_cls = s[1] / max(1, s[3]), s[0] / max(1, s[2])
    ltol = 1e-5 if dtype == "f32" else 3e-2
    assert abs(got_reg - l_reg) <= ltol * abs(l_reg) and abs(got_cls - l_cls) <= ltol * abs(l_cls), (got_reg, l_reg, got_cls, l_cls)
    worst = (0, "")
    for name in LAYERS:
        gk, gb = unpack_grad(tr, Wt, name)
        want = og[name + "/kernel"].double()
        scale = float(want.abs().max())
        if dtype == "f32":
            err = float((gk - want).abs().max()) / scale
            worst = max(worst, (err, name))
            assert err <= 2e-3, "%s: weight-gradient error %.3e of scale %.3e" % (name, err, scale)
            if name + "/bias" in og:
                wb = og[name + "/bias"].double()
                assert float((gb - wb).abs().max()) <= 2e-3 * float(wb.abs().max()), name
        else:
            cos = float((gk * want).sum() / (gk.norm() * want.norm()))
            ratio = float(gk.norm() / want.norm())
            worst = max(worst, (1 - cos, name))
            assert cos >= 0.98 and 0.9 <= ratio <= 1.1, "%s: cos %.4f norm ratio %.3f" % (name, cos, ratio)
    print("%s path: worst layer %s (%.3e)" % (dtype, worst[1], worst[0]))
    if dtype != "f32":
        return
    # ---- optimizer: one global-norm-clipped Adam step against the float64 restatement
    params = {k: torch.as_tensor(np.asarray(v)) for k, v in state.items() if k.endswith("/kernel") or k.endswith("/bias")}
    new, norm = adam_clipnorm_oracle(params, og, {}, {}, 1)
    tr.optimizer_step()
    torch.cuda.synchronize()
    got_norm = float(torch.sqrt(tr.sumsq).item())
    assert abs(got_norm - norm) <= 2e-3 * norm, (got_norm, norm)
    for name in LAYERS:
        lo = eng.layout[name]
        wm = tr.master[lo["woff"]:lo["woff"] + lo["rows"] * lo["K"]].view(lo["rows"], lo["K"]).cpu().double()
        cout = lo["cout"]
        if name == "conv1":
            k = wm[:cout].reshape(cout, 8, 8, 4)[:, :7, :7, :3].permute(1, 2, 3, 0)
        else:
            k = wm[:cout].reshape(cout, lo["kh"], lo["kw"], lo["cin"]).permute(1, 2, 3, 0)
        want = new[name + "/kernel"]
        # Adam's first step moves every weight by ~lr: compare the MOVE, it is what the optimizer computes
        move_got, move_want = k - params[name + "/kernel"].double(), want - params[name + "/kernel"].double()
        assert float((move_got - move_want).abs().max()) <= 0.05 * 1e-4, name
    # forward weights were re-emitted: a second forward differs from the first
    reg2, _ = eng.forward(xd)
    torch.cuda.synchronize()
    assert torch.isfinite(reg2).all()


def test_loss_decreases_over_steps(pkg):
    """Ten bf16 steps on one fixed batch: the total loss goes down (sanity of the whole loop incl. weight re-emission)."""
    E, Wt, T = mods(pkg)
    state = Wt.init_state("resnet50", 1, 9, seed=1, randomize_bn=True, cls_bias=-2.0, tame=True)
    x, reg_t, lab_t = make_batch(2, seed=5)
    eng = E.Engine("resnet50", 1, 9, dtype="bf16")
    eng.load_state(state)
    tr = T.Trainer(eng, lr=1e-4, clipnorm=0.001)
    xd, regd, labd = torch.as_tensor(x).cuda(), torch.as_tensor(reg_t).cuda(), torch.as_tensor(lab_t).cuda()
    losses = [tr.train_on_batch(xd, regd, labd)[0] for _ in range(10)]
    print("losses:", ["%.4f" % v for v in losses])
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


def test_trainer_follows_engine_load_state(pkg):
    """Engine.load_state() reallocates the flat weights and drops the forward plans.  A Trainer that lived through it must not
    write its old master copy over the new weights nor differentiate through the discarded plan's activation buffers: its next
    step equals the step of a Trainer created after the load (fp32: bit-identical losses, gradients within the float-atomic
    noise of the weight gradient, 1e-5 of the gradient scale)."""
    E, Wt, T = mods(pkg)
    st_a = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=-2.0, tame=True)
    st_b = Wt.init_state("resnet50", 1, 9, seed=9, randomize_bn=True, cls_bias=-2.0, tame=True)
    x, reg_t, lab_t = make_batch(2, seed=21)
    xd, regd, labd = torch.as_tensor(x).cuda(), torch.as_tensor(reg_t).cuda(), torch.as_tensor(lab_t).cuda()
    eng = E.Engine("resnet50", 1, 9, dtype="f32")
    eng.load_state(st_a)
    tr = T.Trainer(eng, lr=1e-4, clipnorm=0.001)
    tr.train_on_batch(xd, regd, labd)
    eng.load_state(st_b)                                   # e.g. a checkpoint resume into a live model
    loss_live = tr.train_on_batch(xd, regd, labd)
    torch.cuda.synchronize()
    assert tr.step_count == 1                              # fresh optimizer state for the loaded weights
    g_live, w_live = tr.grad.clone(), tr.master.clone()
    eng2 = E.Engine("resnet50", 1, 9, dtype="f32")
    eng2.load_state(st_b)
    tr2 = T.Trainer(eng2, lr=1e-4, clipnorm=0.001)
    loss_new = tr2.train_on_batch(xd, regd, labd)
    torch.cuda.synchronize()
    assert loss_live == loss_new, (loss_live, loss_new)
    scale = float(tr2.grad.abs().max())
    assert scale > 0 and float((g_live - tr2.grad).abs().max()) <= 1e-5 * scale
    assert float((w_live - tr2.master).abs().max()) <= 1e-3 * 1e-4
    # and a load between backward and the optimizer step is refused: that gradient belongs to other weights
    tr.forward_backward(xd, regd, labd)
    eng.load_state(st_a)
    with pytest.raises(RuntimeError):
        tr.optimizer_step()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_training_steps_repeat_bit_for_bit(pkg, dtype):
    """No float atomics in the step: every weight-gradient kernel stores its pixel splits into slabs and adds them in split order
    (csrc/rtn_backward.hip, rtn_wgrad_halo.hip).  Two trainers started from the same state and fed the same two batches end with
    the same bits in the flat gradient, the Adam moments and the master weights; the wgrad lanes run on side streams meanwhile."""
    E, Wt, T = mods(pkg)
    state = Wt.init_state("resnet50", 1, 9, seed=3, randomize_bn=True, cls_bias=-2.0, tame=True)
    batches = [make_batch(2, seed=31), make_batch(2, seed=32)]
    results = []
    for run in range(2):
        eng = E.Engine("resnet50", 1, 9, dtype=dtype)
        eng.load_state(state)
        tr = T.Trainer(eng, lr=1e-4, clipnorm=0.001)
        losses = []
        for x, reg_t, lab_t in batches:
            losses.append(tr.train_on_batch(torch.as_tensor(x).cuda(), torch.as_tensor(reg_t).cuda(), torch.as_tensor(lab_t).cuda()))
        torch.cuda.synchronize()
        results.append((losses, tr.grad.clone(), tr.master.clone(), tr.m.clone(), tr.v.clone()))
    a, b = results
    assert a[0] == b[0], (a[0], b[0])
    for i, what in ((1, "gradient"), (2, "master weights"), (3, "first moment"), (4, "second moment")):
        assert torch.equal(a[i], b[i]), "%s differs between two identical runs (max %.3e)" % (what, float((a[i] - b[i]).abs().max()))


def test_per_tensor_clipnorm_switch(pkg):
    """Adam(clipnorm=0.001) clips by the global norm under standalone Keras 2.x (the default here) and per tensor under
    tf.keras / Keras >= 2.4 (RetinaNet.py:130 does not pin the version).  Trainer(global_clip=False): every kernel / bias tensor is
    scaled by clipnorm / max(its own norm, clipnorm) - checked against the float64 restatement on the device's own gradient, so that
    only the optimizer differs: first-step moves within 5 % of lr, and visibly different from the global-norm step."""
    E, Wt, T = mods(pkg)
    state = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=-2.0, tame=True)
    x, reg_t, lab_t = make_batch(2, seed=11)
    xd, regd, labd = torch.as_tensor(x).cuda(), torch.as_tensor(reg_t).cuda(), torch.as_tensor(lab_t).cuda()
    params = {k: torch.as_tensor(np.asarray(v)) for k, v in state.items() if k.endswith("/kernel") or k.endswith("/bias")}
    moves = {}
    for global_clip in (True, False):
        eng = E.Engine("resnet50", 1, 9, dtype="f32")
        eng.load_state(state)
        tr = T.Trainer(eng, lr=1e-4, clipnorm=0.001, global_clip=global_clip)
        tr.forward_backward(xd, regd, labd)
        torch.cuda.synchronize()
        grads = {}
        for name in LAYERS:
            gk, gb = unpack_grad(tr, Wt, name)
            grads[name + "/kernel"] = gk
            if eng.layout[name]["has_bias"]:
                grads[name + "/bias"] = gb
        sub = {k: params[k] for k in grads}
        want, _ = adam_clipnorm_oracle(sub, grads, {}, {}, 1, global_clip=False)      # compared below for the per-tensor run only
        tr.optimizer_step()
        torch.cuda.synchronize()
        for name in LAYERS:
            lo = eng.layout[name]
            wm = tr.master[lo["woff"]:lo["woff"] + lo["rows"] * lo["K"]].view(lo["rows"], lo["K"]).cpu().double()
            cout = lo["cout"]
            if name == "conv1":
                k = wm[:cout].reshape(cout, 8, 8, 4)[:, :7, :7, :3].permute(1, 2, 3, 0)
            else:
                k = wm[:cout].reshape(cout, lo["kh"], lo["kw"], lo["cin"]).permute(1, 2, 3, 0)
            move = k - params[name + "/kernel"].double()
            moves[(global_clip, name)] = move
            if not global_clip:                        # per tensor: the oracle on a subset of tensors is the oracle on all of them
                want_move = want[name + "/kernel"] - params[name + "/kernel"].double()
                assert float((move - want_move).abs().max()) <= 0.05 * 1e-4, name
                if lo["has_bias"]:
                    bm = tr.master[tr.NW + lo["boff"]:tr.NW + lo["boff"] + cout].cpu().double() - params[name + "/bias"].double()
                    assert float((bm - (want[name + "/bias"] - params[name + "/bias"].double())).abs().max()) <= 0.05 * 1e-4, name
    # Adam's first step has magnitude ~lr whatever the clip factor is, but elements with |g| near eps move differently: the two
    # semantics must not be the same code path
    assert any(float((moves[(True, n)] - moves[(False, n)]).abs().max()) > 0 for n in LAYERS)


def test_plan_cache_is_bounded_over_many_canvases(pkg):
    """csv_generator.compute_inputs pads every batch to its largest page, so fit_generator (RetinaNet.py:280, batch_size 1) sees a new
    canvas for almost every page shape.  A forward plan (activations, descriptors, conv workspaces) and its backward plan (gradient
    buffers, one row-info table + split slabs per weight gradient) are several GB at page size: the engine keeps the `max_plans`
    most recently used canvases and the Trainer's backward plans go with them.  Twelve canvases through a cache of three: memory
    stays flat after the third, a canvas that was evicted trains again and gives the bits it gave the first time (same weights)."""
    E, Wt, T = mods(pkg)
    L = pkg._lib
    state = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=-2.0, tame=True)
    eng = E.Engine("resnet50", 1, 9, dtype="bf16")
    eng.max_plans = 3
    eng.load_state(state)
    tr = T.Trainer(eng, lr=0.0, clipnorm=0.001)                        # lr 0: the weights stay, so a revisited canvas must repeat its bits
    canvases = [(64 + 32 * (i % 4), 96 + 32 * (i // 4)) for i in range(12)]
    g = torch.Generator().manual_seed(3)
    first, mem = {}, []
    for H, W in canvases + canvases[:2]:
        x = (torch.rand(1, H, W, 3, generator=torch.Generator().manual_seed(H * 1000 + W)) * 2 - 1).cuda()
        cfg, N = E.make_anchor_cfg((H, W))
        reg_t = torch.zeros(1, N, 5, device="cuda")
        lab_t = torch.zeros(1, N, 2, device="cuda")
        lab_t[0, ::97, 0] = 1.0                                        # some positives: labels 1, anchor state 1
        lab_t[0, ::97, 1] = 1.0
        reg_t[0, ::97, 4] = 1.0
        tr.forward_backward(x, reg_t, lab_t)
        tr.optimizer_step()
        torch.cuda.synchronize()
        gsum = tr.grad.double().abs().sum().item()
        assert np.isfinite(gsum) and gsum > 0
        if (H, W) in first:
            assert torch.equal(first[(H, W)], tr.grad.cpu()), "an evicted canvas came back with other gradients"
        else:
            first[(H, W)] = tr.grad.cpu().clone()
        assert len(eng.plans) <= 3 and len(tr.bplans) <= 3
        assert set(k[:3] for k in eng.plans) >= set(tr.bplans)         # no backward plan outlives its forward plan
        mem.append(torch.cuda.memory_allocated())
    assert max(mem[3:]) <= 1.35 * max(mem[:3]), "memory grew with the number of canvases seen: %s" % [m >> 20 for m in mem]
