"""GPU tests of the training-graph pieces: hand-written backward kernels against
the oracle (bit-exact) and one end-to-end SGD step of the trainable graph."""
import numpy as np
import pytest

import oracle
from posecnn_amd import config, synth

pytestmark = pytest.mark.gpu
F = np.float32


def T(gpu, a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(gpu)


def same(a, b, what):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape, what
    bad = a.view(np.uint32) != b.view(np.uint32)
    assert not bad.any(), "%s: %d of %d differ, first %s vs %s" % (what, bad.sum(), bad.size, a[bad][:3], b[bad][:3])


@pytest.mark.parametrize("shape,k,s", [((2, 15, 20, 64), 16, 8), ((1, 7, 9, 128), 4, 2), ((1, 5, 6, 3), 4, 2), ((1, 4, 4, 22), 4, 4)])
def test_deconv_bilinear_backward(gpu, shape, k, s):
    import torch
    from posecnn_amd import ops
    rng = np.random.default_rng(51)
    B, H, W, C = shape
    x = rng.standard_normal(shape).astype(F)
    g = rng.standard_normal((B, H * s, W * s, C)).astype(F)
    same(ops.deconv_bilinear_grad(T(gpu, g), k, s).cpu().numpy(), oracle.deconv_bilinear_bwd(g, k, s), "deconv bwd")
    xt = T(gpu, x).requires_grad_(True)
    y = ops.deconv_bilinear(xt, k, s)                       # autograd route
    same(y.detach().cpu().numpy(), oracle.deconv_bilinear(x, k, s), "deconv fwd (autograd route)")
    y.backward(T(gpu, g))
    same(xt.grad.cpu().numpy(), oracle.deconv_bilinear_bwd(g, k, s), "deconv autograd")
    # fused extras fall back to framework ops when a gradient is needed, same values to rounding
    bias = T(gpu, rng.standard_normal(C).astype(F)).requires_grad_(True)
    y2 = ops.deconv_bilinear(T(gpu, x), k, s, bias=bias, relu=True)
    y2.sum().backward()
    assert bias.grad is not None and torch.isfinite(bias.grad).all()
    assert np.allclose(y2.detach().cpu().numpy(), oracle.deconv_bilinear(x, k, s, None, None, bias.detach().cpu().numpy(), True), atol=1e-6)


@pytest.mark.parametrize("n,sigma", [(1000, 1.0), (16 * 60 * 80 * 66, 1.0), (300001, 3.0), (1, 1.0)])
def test_smooth_l1_vertex_forward_backward(gpu, n, sigma):
    import torch
    from posecnn_amd import ops
    rng = np.random.default_rng(52)
    p = (rng.standard_normal(n) * 2).astype(F); t = (rng.standard_normal(n) * 2).astype(F)
    w = (rng.random(n) < 0.3).astype(F)
    out, grad = oracle.smooth_l1_vertex(p, t, w, sigma)
    pt = T(gpu, p).requires_grad_(True)
    loss = ops.smooth_l1_loss_vertex(pt, T(gpu, t), T(gpu, w), sigma)
    same(loss.detach().cpu().numpy().reshape(1), out[:1], "smooth l1 loss")
    (loss * 5.0).backward()                                  # VERTEX_W = 5 upstream
    same(pt.grad.cpu().numpy(), (grad * F(5.0)).astype(F), "smooth l1 grad")
    with pytest.raises(ValueError):
        ops.smooth_l1_loss_vertex(pt, T(gpu, t[:-1]) if n > 1 else T(gpu, np.zeros(2, F)), T(gpu, w), sigma)


def training_feed(gpu, B, H, W, seed):
    import torch
    K = config.DEMO_INTRINSICS.copy(); K[:2] *= W / 640.0
    label, vertex, fr = synth.make_batch(seed, B, H=H, W=W, C=22, n_obj=2, K=K)
    rng = np.random.default_rng(seed)
    data = (rng.integers(0, 256, (B, H, W, 3)).astype(F) - config.PIXEL_MEANS).astype(F)
    weights = np.zeros_like(vertex)
    for c in range(1, 22):
        weights[..., 3 * c:3 * c + 3] = (label == c)[..., None]
    rows = []
    for n in range(B):
        for (cls, cx, cy, z) in fr[n]["objects"]:
            q = synth.random_unit_quats(rng, 1)[0]
            rows.append([n, cls, 0, 0, 0, 0, q[0], q[1], q[2], q[3], (cx - K[0, 2]) / K[0, 0] * z, (cy - K[1, 2]) / K[1, 1] * z, z])
    meta = np.stack([config.make_meta_data(K)] * B).reshape(B, 1, 1, 48)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu)
    return {"data": t(data), "gt_label_2d": t(label.astype(np.int32)), "keep_prob": 1.0,
            "vertex_targets": t(vertex), "vertex_weights": t(weights), "poses": t(np.array(rows, F)),
            "extents": t(config.LOV_EXTENTS), "meta_data": t(meta), "points": t(synth.make_model_points(22, 64)),
            "symmetry": t(config.LOV_SYMMETRY)}


def test_one_training_step_of_the_full_graph(gpu):
    """vgg16_convs(is_train=True) + the losses of train.py:488-519 + momentum SGD: every trainable
    variable receives a finite gradient and repeating the step on the same batch lowers the loss."""
    import torch
    from posecnn_amd import train
    from posecnn_amd.networks import vgg16_convs
    torch.manual_seed(0)
    net = vgg16_convs("COLOR", 22, 64, (1.0,), 1.0, -1.0, vertex_reg_2d=True, pose_reg=True, trainable=True,
                      is_train=True, device=gpu, seed=3, init="he")
    assert not net.fused_heads                       # literal op order: everything is differentiable
    feed = training_feed(gpu, 1, 160, 208, 77)

    class Cfg(train.TrainConfig):
        LEARNING_RATE = 1e-6

    solver = train.SolverWrapper(net, Cfg)
    first = solver.train_step(feed)
    for k in ("loss", "loss_cls", "loss_vertex", "loss_pose", "loss_regu"):
        assert np.isfinite(first[k]), (k, first)
    assert first["loss_vertex"] > 0 and first["loss_cls"] > 0
    assert net.get_output("rois").shape[0] >= 2     # is_train: one ROI per gt object (+ jittered copies)
    trainable = {k for k, v in net.vars.items() if v.requires_grad}
    assert {"conv1_1/weights", "conv5_3/biases", "score/weights", "vertex_pred/weights", "fc6/weights", "fc8/biases"} <= trainable
    assert not any(k.startswith("upscore") for k in trainable)        # deconv(..., trainable=False)
    for k in sorted(trainable):
        g = net.vars[k].grad
        assert g is not None and torch.isfinite(g).all(), k
    for k in ("conv1_1/weights", "conv4_3/weights", "vertex_pred/weights", "score/weights", "fc7/weights"):
        assert float(net.vars[k].grad.abs().max()) > 0, k
    later = first
    for _ in range(3):
        later = solver.train_step(feed)
    # tiny learning rate: the data terms must not grow (MIOpen's backward kernels are not bitwise
    # reproducible, so allow rounding-level noise on a loss dominated by the constant weight decay)
    assert later["loss"] <= first["loss"] * (1 + 1e-5), (first, later)
    assert later["loss_vertex"] + later["loss_cls"] <= (first["loss_vertex"] + first["loss_cls"]) * (1 + 1e-4), (first, later)
    # snapshot / restore round trip
    import os, tempfile
    path = os.path.join(tempfile.mkdtemp(), "snap.pt")
    solver.snapshot(path)
    before = net.vars["conv3_1/weights"].detach().clone()
    solver.train_step(feed)
    assert not torch.equal(before, net.vars["conv3_1/weights"])
    solver.restore(path)
    assert torch.equal(before, net.vars["conv3_1/weights"]) and solver.iter == 4
