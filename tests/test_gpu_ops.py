"""GPU parity tests for roi_pool, hard_label, average_distance_loss, backproject and the label-head
epilogue: libposecnn_hip.so via posecnn_amd.ops vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

import oracle
from posecnn_amd import config, synth

pytestmark = pytest.mark.gpu
F = np.float32


def T(gpu, a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(gpu)


def N(t):
    return t.detach().cpu().numpy()


def same(got, want, name=""):
    got, want = np.ascontiguousarray(got), np.ascontiguousarray(want)
    assert got.shape == want.shape, name
    if got.dtype.kind == "f":
        ok = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    else:
        ok = got == want
    assert ok.all(), "%s: %d mismatches, first at %s: gpu %s oracle %s" % (
        name, (~ok).sum(), np.argwhere(~ok)[0], got[~ok][:3], want[~ok][:3])


def random_rois(rng, R, B, C, W, H, cols=7):
    rois = np.zeros((R, cols), F)
    rois[:, 0] = rng.integers(0, B, R)
    rois[:, 1] = rng.integers(0, C, R)
    x1 = rng.uniform(-60, W - 20, R); y1 = rng.uniform(-60, H - 20, R)
    rois[:, 2], rois[:, 3] = x1, y1
    rois[:, 4] = x1 + rng.uniform(-30, 0.6 * W, R)
    rois[:, 5] = y1 + rng.uniform(-30, 0.6 * H, R)
    rois[:4, 2:6] = np.round(rois[:4, 2:6] / 16) * 16 + 8  # x.5 after the 1/16 scale
    return rois


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape,scale", [((2, 30, 40, 512), 1 / 16.0), ((2, 60, 80, 512), 1 / 8.0), ((3, 15, 20, 64), 1 / 16.0)])
def test_roi_pool_forward(gpu, shape, scale):
    from posecnn_amd import ops
    rng = np.random.default_rng(11)
    B, H, W, C = shape
    data = rng.standard_normal(shape).astype(F)
    rois = random_rois(rng, 23, B, 22, 640, 480)
    top, arg = ops.roi_pool(T(gpu, data), T(gpu, rois), 7, 7, scale, 0)
    wt, wa = oracle.roi_pool(data, rois, 7, 7, scale, 0)
    same(N(top), wt, "top")
    same(N(arg), wa, "argmax")


def test_roi_pool_generic_paths(gpu):
    from posecnn_amd import ops
    rng = np.random.default_rng(12)
    data = rng.standard_normal((2, 12, 17, 22)).astype(F)  # C % 4 != 0 -> scalar kernel
    data[0, 3, 4, :] = np.nan
    data[1, 5, 6, :] = -np.inf
    rois = random_rois(rng, 31, 2, 22, 17 * 4, 12 * 4)
    rois[5, 0] = 9    # bad batch index
    rois[6, 1] = 40   # bad class (matters only for pool_channel)
    for pc in (0, 1):
        for (ph, pw) in ((7, 7), (3, 5), (1, 1)):
            top, arg = ops.roi_pool(T(gpu, data), T(gpu, rois), ph, pw, 0.25, pc)
            wt, wa = oracle.roi_pool(data, rois, ph, pw, 0.25, pc)
            same(N(top), wt, "top pc=%d" % pc)
            same(N(arg), wa, "argmax pc=%d" % pc)
    # 5-column-plus layout of the stale reference test is rejected: needs >= 6 columns
    with pytest.raises(ValueError):
        ops.roi_pool(T(gpu, data), T(gpu, rois[:, :5]), 7, 7, 0.25, 0)
    # zero ROIs
    top, arg = ops.roi_pool(T(gpu, data), T(gpu, rois[:0]), 7, 7, 0.25, 0)
    assert top.shape == (0, 7, 7, 22)


def test_roi_pool_add2_equals_two_pools(gpu):
    from posecnn_amd import ops
    rng = np.random.default_rng(13)
    c5 = rng.standard_normal((2, 30, 40, 512)).astype(F)
    c4 = rng.standard_normal((2, 60, 80, 512)).astype(F)
    rois = random_rois(rng, 19, 2, 22, 640, 480)
    out = ops.roi_pool_add2(T(gpu, c5), 1 / 16.0, T(gpu, c4), 1 / 8.0, T(gpu, rois))
    a, _ = oracle.roi_pool(c5, rois, 7, 7, 1 / 16.0, 0)
    b, _ = oracle.roi_pool(c4, rois, 7, 7, 1 / 8.0, 0)
    same(N(out), a + b, "pool_score")


def test_roi_pool_backward(gpu):
    import torch
    from posecnn_amd import ops
    rng = np.random.default_rng(14)
    B, H, W, C = 2, 10, 12, 8
    data = rng.standard_normal((B, H, W, C)).astype(F)
    rois = random_rois(rng, 9, B, C, W * 8, H * 8)
    for pc in (0, 1):
        d = T(gpu, data).requires_grad_(True)
        top, arg = ops.roi_pool(d, T(gpu, rois), 3, 3, 0.125, pc)
        g = rng.standard_normal(tuple(top.shape)).astype(F)
        top.backward(T(gpu, g))
        want = oracle.roi_pool_bwd(g, rois, N(arg), B, H, W, C, 3, 3, 0.125, pc)
        same(N(d.grad), want, "bottom_diff pc=%d" % pc)



def test_roi_pool_ties_wide_rois_and_long_roi_lists(gpu):
    """Cases aimed at the staged/binned kernels: heavy value ties (first maximum in (h, w) order must
    win), ROIs wider than the LDS column window (direct fallback), malformed ROIs (end < start: the
    forward pools one cell, the backward's rectangle test drops it), a ROI table longer than the
    backward's LDS list (several compaction passes), C not a multiple of the chunk."""
    import torch
    from posecnn_amd import ops
    rng = np.random.default_rng(15)
    B, H, W, C = 2, 9, 150, 40
    data = rng.integers(-2, 3, (B, H, W, C)).astype(F)      # 5 distinct values -> ties everywhere
    R = 1200
    rois = random_rois(rng, R, B, C, W * 4, H * 4)
    rois[:40, 2] = 0; rois[:40, 4] = W * 4 - 1                # full-width ROIs: 150 columns
    rois[40:60, 4] = rois[40:60, 2] - 13                      # x2 < x1
    rois[60:80, 5] = rois[60:80, 3] - 9                       # y2 < y1
    rois[80:700, 0] = 1                                       # > RB_LIST ROIs on one image ...
    rois[80:700, 2:6] = [0, 0, W * 4 - 1, H * 4 - 1]          # ... all touching every tile
    for pc in (0, 1):
        d = T(gpu, data).requires_grad_(True)
        top, arg = ops.roi_pool(d, T(gpu, rois), 4, 6, 0.25, pc)
        wt, wa = oracle.roi_pool(data, rois, 4, 6, 0.25, pc)
        same(N(top), wt, "top pc=%d" % pc)
        same(N(arg), wa, "argmax pc=%d" % pc)
        g = rng.integers(-3, 4, tuple(top.shape)).astype(F) * F(0.37)
        top.backward(T(gpu, g))
        want = oracle.roi_pool_bwd(g, rois, wa, B, H, W, C, 4, 6, 0.25, pc)
        same(N(d.grad), want, "bottom_diff pc=%d" % pc)
    # fused two-tensor entry on ROIs wider than the shared window
    a = rng.integers(-2, 3, (B, H, W, C)).astype(F)
    b = rng.integers(-2, 3, (B, 2 * H, 2 * W, C)).astype(F)
    out = ops.roi_pool_add2(T(gpu, a), 0.25, T(gpu, b), 0.5, T(gpu, rois[:120]))
    pa, _ = oracle.roi_pool(a, rois[:120], 7, 7, 0.25, 0)
    pb, _ = oracle.roi_pool(b, rois[:120], 7, 7, 0.5, 0)
    same(N(out), pa + pb, "pool_score wide")

# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(1, 480, 640, 22), (2, 33, 47, 22), (1, 5, 7, 3), (1, 1, 1, 16)])
def test_hard_label(gpu, shape):
    from posecnn_amd import ops
    rng = np.random.default_rng(15)
    prob = rng.random(shape).astype(F)
    gt = rng.integers(-1, shape[3], shape[:3]).astype(np.int32)
    k = min(3, gt.size)
    gt.ravel()[:k] = [-5, shape[3] + 2, -1][:k]  # out-of-range labels are ignored
    for thr in (0.3, 1.0):
        out = ops.hard_label(T(gpu, prob), T(gpu, gt), thr)
        same(N(out), oracle.hard_label(prob, gt, thr), "hard_label")
    with pytest.raises(ValueError):
        ops.hard_label(T(gpu, prob), T(gpu, gt), 0.0)


def test_hard_label_one_hot_property_full_batch(gpu):
    """Size-independent property at bench size (16 frames): each pixel has at most one 1, exactly
    one when gt > 0, and the checksum equals the count predicted from (gt, prob)."""
    import torch
    from posecnn_amd import ops
    g = torch.Generator(device=gpu).manual_seed(0)
    prob = torch.rand((16, 480, 640, 22), device=gpu, generator=g)
    gt = torch.randint(-1, 22, (16, 480, 640), device=gpu, generator=g, dtype=torch.int32)
    out = ops.hard_label(prob, gt, 0.5)
    s = out.sum(-1)
    assert float(s.max()) <= 1.0
    assert bool((s[gt > 0] == 1).all()) and bool((s[gt == -1] == 0).all())
    p0 = prob[..., 0]
    assert float(s[gt == 0].sum()) == float(((p0 < 0.5) & (gt == 0)).sum())
    assert bool((out.gather(-1, gt.clamp(min=0).long().unsqueeze(-1)).squeeze(-1) == s).all())


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(1, 480, 640, 22), (2, 17, 31, 22), (1, 9, 9, 40), (3, 4, 4, 2)])
def test_softmax_argmax(gpu, shape):
    from posecnn_amd import ops
    rng = np.random.default_rng(16)
    score = np.maximum(rng.standard_normal(shape) * 4, 0).astype(F)  # ReLU'd scores, many ties at 0
    score.reshape(-1, shape[3])[0] = 0
    score.reshape(-1, shape[3])[1] = 200  # exp overflow guard: max-subtracted
    prob, lab = ops.softmax_argmax(T(gpu, score))
    wp, wl = oracle.softmax_argmax(score)
    same(N(lab), wl, "label_2d")
    same(N(prob), wp, "prob_normalized")
    _, lab2 = ops.softmax_argmax(T(gpu, score), want_prob=False)
    same(N(lab2), wl, "label only")


# ------------------------------------------------------------------------------------------------
def adl_case(rng, R, C, P, sym_classes=(16, 21)):
    pts = synth.make_model_points(C, P)
    sym = np.zeros(C, F)
    for c in sym_classes:
        if c < C:
            sym[c] = 1
    pred = np.zeros((R, 4 * C), F); tgt = np.zeros((R, 4 * C), F); wgt = np.zeros((R, 4 * C), F)
    classes = [c for c in (1, 16, 5, 21, 2, 16, 7, 21) if c < C] or [1]
    for n in range(R):
        if n % 5 == 4:
            continue  # rows without a class are skipped (index_cls == -1)
        c = classes[n % len(classes)]
        pred[n, 4 * c:4 * c + 4] = np.tanh(rng.standard_normal(4)).astype(F)
        tgt[n, 4 * c:4 * c + 4] = synth.random_unit_quats(rng, 1)[0]
        wgt[n, 4 * c:4 * c + 4] = 1
    return pred, tgt, wgt, pts, sym


@pytest.mark.parametrize("R,C,P,margin", [(7, 22, 2620, 0.01), (3, 22, 300, 0.0), (12, 5, 1025, 0.01), (1, 22, 64, 0.01)])
def test_average_distance_forward(gpu, R, C, P, margin):
    from posecnn_amd import ops
    rng = np.random.default_rng(17)
    pred, tgt, wgt, pts, sym = adl_case(rng, R, C, P, sym_classes=(16, 21) if C == 22 else (2,))
    loss, diff = ops.average_distance_loss(T(gpu, pred), T(gpu, tgt), T(gpu, wgt), T(gpu, pts), T(gpu, sym), margin)
    wl, wd = oracle.average_distance(pred, tgt, wgt, pts, sym, margin)
    assert wl[0] > 0
    same(N(loss), wl, "loss")
    same(N(diff), wd, "bottom_diff")


def test_average_distance_edge_cases_and_backward(gpu):
    import torch
    from posecnn_amd import ops
    rng = np.random.default_rng(18)
    pred, tgt, wgt, pts, sym = adl_case(rng, 6, 22, 500)
    # no row has a class -> loss 0, zero gradient
    loss, diff = ops.average_distance_loss(T(gpu, pred), T(gpu, tgt), T(gpu, wgt * 0), T(gpu, pts), T(gpu, sym), 0.01)
    assert float(loss) == 0 and float(diff.abs().sum()) == 0
    # zero rows
    loss, diff = ops.average_distance_loss(T(gpu, pred[:0]), T(gpu, tgt[:0]), T(gpu, wgt[:0]), T(gpu, pts), T(gpu, sym), 0.01)
    assert float(loss) == 0 and diff.shape == (0, 88)
    # AveragedistanceGrad: out = grad[0] * bottom_diff
    p = T(gpu, pred).requires_grad_(True)
    loss, diff = ops.average_distance_loss(p, T(gpu, tgt), T(gpu, wgt), T(gpu, pts), T(gpu, sym), 0.01)
    (loss * 3.0).sum().backward()
    want = oracle.average_distance_bwd(np.array([3.0], F), N(diff))
    same(N(p.grad), want, "grad")
    with pytest.raises(ValueError):
        ops.average_distance_loss(T(gpu, pred), T(gpu, tgt), T(gpu, wgt), T(gpu, pts), T(gpu, sym), -0.1)


# ------------------------------------------------------------------------------------------------
def backproject_case(rng, B, H, W, Cd, Cl, G):
    data = rng.standard_normal((B, H, W, Cd)).astype(F)
    label = rng.random((B, H, W, Cl)).astype(F)
    yy, xx = np.mgrid[0:H, 0:W]
    depth = (1.6 + 0.3 * np.sin(xx / 9.0) + 0.2 * np.cos(yy / 7.0) + 0.01 * rng.standard_normal((B, H, W))).astype(F)
    label3d = rng.random((B, G, G, G, Cl)).astype(F)
    K = np.array([[W * 0.9, 0, W / 2.0], [0, W * 0.9, H / 2.0], [0, 0, 1]])
    a = 0.05
    w2l = np.array([[np.cos(a), 0, np.sin(a), 0.01], [0, 1, 0, -0.02], [-np.sin(a), 0, np.cos(a), 0.03]], F)
    l2w = np.array([[np.cos(a), 0, -np.sin(a), -0.01], [0, 1, 0, 0.02], [np.sin(a), 0, np.cos(a), -0.03]], F)
    step = (2.4 / G, 2.0 / G, 1.2 / G)
    meta = np.stack([config.make_meta_data(K, voxel_step=step, voxel_min=(-1.2, -1.0, 1.1),
                                           pose_world2live=w2l, pose_live2world=l2w)] * B)
    return data, label, depth[..., None], meta, label3d


@pytest.mark.parametrize("B,H,W,Cd,Cl,G,k", [(1, 48, 64, 64, 22, 24, 3), (2, 20, 28, 6, 3, 9, 1), (1, 16, 16, 5, 2, 8, 0),
                                            # fused one-scan-per-voxel kernel: every lanes-per-voxel variant, a voxel
                                            # count that is no multiple of 64, two images; and k = 4 (81-pixel window
                                            # > 64 bits) which must take the per-channel kernels
                                            (2, 24, 32, 128, 5, 7, 3), (1, 24, 32, 32, 22, 9, 2), (1, 20, 24, 16, 4, 6, 3),
                                            (2, 20, 24, 8, 3, 5, 1), (1, 20, 24, 4, 3, 5, 3), (1, 24, 32, 64, 22, 10, 4)])
def test_backproject_forward(gpu, B, H, W, Cd, Cl, G, k):
    from posecnn_amd import ops
    rng = np.random.default_rng(19)
    data, label, depth, meta, label3d = backproject_case(rng, B, H, W, Cd, Cl, G)
    m4 = meta.reshape(B, 1, 1, 48)
    td, tl, tf = ops.backproject(T(gpu, data), T(gpu, label), T(gpu, depth), T(gpu, m4), T(gpu, label3d), G, k, 0.05)
    wd, wl, wf = oracle.backproject(data, label, depth, meta, label3d, G, k, 0.05)
    assert wf.sum() > 0
    same(N(td), wd, "top_data")
    same(N(tf), wf, "top_flag")
    same(N(tl), wl, "top_label")


def test_backproject_degenerate_projection_and_backward(gpu):
    import torch
    from posecnn_amd import ops
    rng = np.random.default_rng(20)
    B, H, W, Cd, Cl, G = 1, 16, 20, 4, 3, 6
    data, label, depth, meta, label3d = backproject_case(rng, B, H, W, Cd, Cl, G)
    meta0 = meta.copy()
    meta0[:, 18:30] = 0  # zero pose (as lib/fcn/test.py leaves it): x3 = 0 -> 0/0 -> px = py = 0
    m4 = meta0.reshape(B, 1, 1, 48)
    td, tl, tf = ops.backproject(T(gpu, data), T(gpu, label), T(gpu, depth), T(gpu, m4), T(gpu, label3d), G, 2, 5.0)
    wd, wl, wf = oracle.backproject(data, label, depth, meta0, label3d, G, 2, 5.0)
    same(N(td), wd, "top_data"); same(N(tf), wf, "top_flag"); same(N(tl), wl, "top_label")
    # backward (pixel -> voxel gather)
    m4 = meta.reshape(B, 1, 1, 48)
    d = T(gpu, data).requires_grad_(True)
    td, tl, tf = ops.backproject(d, T(gpu, label), T(gpu, depth), T(gpu, m4), T(gpu, label3d), G, 1, 0.05)
    g = rng.standard_normal(tuple(td.shape)).astype(F)
    td.backward(T(gpu, g))
    want = oracle.backproject_bwd(g, depth, meta, B, H, W, Cd, G)
    assert np.abs(want).sum() > 0
    same(N(d.grad), want, "bottom_diff")


# ------------------------------------------------------------------------------------------------
def test_device_math_is_ieee(gpu):
    """The exactness argument rests on IEEE-correct f32 divide/sqrt and on the canonical exp; check
    them end to end through an op whose output exposes them: softmax over 2 channels gives
    exp(x)/(1+exp(x)) ... and hough's project_box; here: softmax on wide-range inputs."""
    from posecnn_amd import ops
    rng = np.random.default_rng(21)
    x = np.zeros((200000, 2), F)
    x[:, 1] = -np.abs(rng.uniform(0, 100, 200000)).astype(F)
    x[:1000, 1] = -np.linspace(0, 104, 1000).astype(F)
    prob, lab = ops.softmax_argmax(T(gpu, x))
    wp, wl = oracle.softmax_argmax(x)
    same(N(prob), wp, "softmax wide range")


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape,k,s", [((2, 30, 40, 64), 4, 2), ((1, 60, 80, 128), 16, 8), ((2, 7, 9, 66), 16, 8),
                                       ((1, 5, 6, 22), 4, 2), ((1, 3, 4, 5), 2, 2), ((1, 4, 4, 3), 6, 2)])
def test_deconv_bilinear(gpu, shape, k, s):
    from posecnn_amd import ops
    rng = np.random.default_rng(22)
    x = rng.standard_normal(shape).astype(F)
    B, H, W, C = shape
    out = ops.deconv_bilinear(T(gpu, x), k, s)
    same(N(out), oracle.deconv_bilinear(x, k, s), "deconv")
    a1 = rng.standard_normal((B, H * s, W * s, C)).astype(F)
    a2 = rng.standard_normal((B, H * s, W * s, C)).astype(F)
    bias = rng.standard_normal(C).astype(F)
    out = ops.deconv_bilinear(T(gpu, x), k, s, add1=T(gpu, a1), add2=T(gpu, a2), bias=T(gpu, bias), relu=True)
    same(N(out), oracle.deconv_bilinear(x, k, s, a1, a2, bias, True), "deconv + adds + bias + relu")
    # interior pixels of the stride-s bilinear kernel interpolate: a constant image stays constant
    if k == 2 * s and H > 2 and W > 2:
        ones = np.ones(shape, F)
        y = N(ops.deconv_bilinear(T(gpu, ones), k, s))
        assert np.all(y[:, s:-s, s:-s] == 1.0)


@pytest.mark.parametrize("shape,relu", [((2, 60, 80, 22), True), ((1, 7, 9, 22), True), ((1, 5, 5, 40), False), ((1, 30, 33, 3), True)])
def test_upscore_softmax_argmax(gpu, shape, relu):
    from posecnn_amd import ops
    rng = np.random.default_rng(23)
    z = (rng.standard_normal(shape) * 3).astype(F)
    bias = rng.standard_normal(shape[3]).astype(F)
    score, prob, label = ops.upscore_softmax_argmax(T(gpu, z), T(gpu, bias), 16, 8, relu=relu, want_score=True)
    ws, wp, wl = oracle.upscore_softmax_argmax(z, bias, 16, 8, relu)
    same(N(score), ws, "score")
    same(N(label), wl, "label_2d")
    same(N(prob), wp, "prob_normalized")
    s2, p2, l2 = ops.upscore_softmax_argmax(T(gpu, z), T(gpu, bias), 16, 8, relu=relu, want_score=False, want_prob=False)
    assert s2 is None and p2 is None
    same(N(l2), wl, "label only")


@pytest.mark.parametrize("shape", [(2, 30, 40, 512), (1, 7, 9, 22), (3, 5, 5, 64)])
def test_bias_act_inplace(gpu, shape):
    from posecnn_amd import ops
    rng = np.random.default_rng(24)
    x = rng.standard_normal(shape).astype(F)
    b = rng.standard_normal(shape[-1]).astype(F)
    for relu in (True, False):
        t = T(gpu, x)
        out = ops.bias_act_(t, T(gpu, b), relu)
        assert out.data_ptr() == t.data_ptr()
        want = x + b
        if relu:
            want = np.maximum(want, 0)
        same(N(out), want.astype(F), "bias_act relu=%s" % relu)


@pytest.mark.parametrize("shape", [(2, 48, 64, 64), (1, 6, 10, 3), (3, 2, 2, 128), (1, 30, 40, 256)])
def test_bias_relu_pool2_equals_pool_of_bias_act(gpu, shape):
    """conv -> max_pool pairs of the VGG trunk: one pass from the raw conv output, same bits as
    max_pool_2x2(ReLU(x + b)) (numpy restatement; network.py:181-187 + :189-196)."""
    from posecnn_amd import ops
    rng = np.random.default_rng(31)
    x = (rng.standard_normal(shape) * 3).astype(F)
    b = rng.standard_normal(shape[-1]).astype(F)
    B, H, W, C = shape
    for relu in (True, False):
        act = x + b
        if relu:
            act = np.maximum(act, 0)
        want = act.reshape(B, H // 2, 2, W // 2, 2, C).max(axis=(2, 4)).astype(F)
        got = ops.bias_relu_pool2(T(gpu, x), T(gpu, b), relu)
        assert got.shape == (B, H // 2, W // 2, C)
        same(N(got), want, "bias_relu_pool2 relu=%s" % relu)
    with pytest.raises(ValueError):
        ops.bias_relu_pool2(T(gpu, x[:, :H - 1]), T(gpu, b))


@pytest.mark.parametrize("shape,cout", [((2, 48, 64), 64), ((1, 5, 131), 64), ((1, 33, 7), 128), ((1, 1, 1), 64), ((1, 17, 300), 64)])
def test_conv3x3_c3_bias_relu(gpu, shape, cout):
    """conv1_1 fused kernel against a float64 restatement (tolerance: f32 accumulation of 27 products)
    and against the library convolution the unfused path uses."""
    import torch
    from posecnn_amd import ops
    rng = np.random.default_rng(41)
    B, H, W = shape
    x = (rng.standard_normal((B, H, W, 3)) * 50).astype(F)          # mean-subtracted pixels: O(100)
    w = (rng.standard_normal((3, 3, 3, cout)) * 0.3).astype(F)      # (ky, kx, ci, co)
    b = rng.standard_normal(cout).astype(F)
    xp = np.zeros((B, H + 2, W + 2, 3), np.float64); xp[:, 1:-1, 1:-1] = x
    want = np.zeros((B, H, W, cout), np.float64)
    for ky in range(3):
        for kx in range(3):
            want += np.einsum("bhwc,oc->bhwo", xp[:, ky:ky + H, kx:kx + W], w[ky, kx].T.astype(np.float64))
    want += b
    for relu in (True, False):
        ref = np.maximum(want, 0) if relu else want
        got = N(ops.conv3x3_c3(T(gpu, x), T(gpu, w), T(gpu, b), relu))
        assert got.shape == (B, H, W, cout)
        assert np.abs(got - ref).max() <= 2e-5 * np.abs(want).max()
    lib_y = torch.nn.functional.conv2d(T(gpu, x).permute(0, 3, 1, 2), T(gpu, w).permute(3, 2, 0, 1), T(gpu, b), padding=1)
    assert np.abs(N(lib_y.permute(0, 2, 3, 1)) - want).max() <= 2e-5 * np.abs(want).max()
    with pytest.raises(ValueError):
        ops.conv3x3_c3(T(gpu, x), T(gpu, w[..., :48]), T(gpu, b[:48]))


# ---- Winograd F(2x2,3x3) transforms (csrc/winograd.hip) -----------------------------------------------
BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)
GM = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)


def np_wino_input(x):
    """f32 restatement of wino_input_kernel: rows first, then columns (every entry of B^T is 0 / +-1 and
    every row has two non-zeros, so each output is ONE f32 add or subtract — no order ambiguity)."""
    B, H, W, C = x.shape
    xp = np.zeros((B, H + 2, W + 2, C), F); xp[:, 1:-1, 1:-1] = x
    Ht, Wt = H // 2, W // 2
    d = np.empty((4, 4, B, Ht, Wt, C), F)
    for r in range(4):
        for s in range(4):
            d[r, s] = xp[:, r:r + H:2, s:s + W:2][:, :Ht, :Wt]
    t = np.stack([d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]])            # [4(i), 4(s), ...]
    v = np.stack([t[:, 0] - t[:, 2], t[:, 1] + t[:, 2], t[:, 2] - t[:, 1], t[:, 1] - t[:, 3]], axis=1)  # [i, j, ...]
    return v.reshape(16, B * Ht * Wt, C)


def np_wino_output(m, bias, B, H, W, relu):
    C = m.shape[2]
    q = m.reshape(4, 4, B, H // 2, W // 2, C)
    t0 = q[0] + q[1] + q[2]; t1 = q[1] - q[2] - q[3]                               # [j, ...]
    o = np.empty((2, 2, B, H // 2, W // 2, C), F)
    for a, t in enumerate((t0, t1)):
        o[a, 0] = t[0] + t[1] + t[2] + bias
        o[a, 1] = t[1] - t[2] - t[3] + bias
    if relu:
        o = np.maximum(o, 0)
    y = np.empty((B, H, W, C), F)
    for a in range(2):
        for b_ in range(2):
            y[:, a::2, b_::2] = o[a, b_]
    return y


@pytest.mark.parametrize("shape", [(2, 8, 12, 16), (1, 30, 40, 512), (1, 2, 2, 4), (3, 6, 4, 260)])
def test_winograd_transforms_bit_exact(gpu, shape):
    from posecnn_amd import ops
    rng = np.random.default_rng(61)
    B, H, W, C = shape
    x = rng.standard_normal(shape).astype(F)
    same(N(ops.winograd_input(T(gpu, x))), np_wino_input(x), "winograd input transform")
    m = rng.standard_normal((16, B * (H // 2) * (W // 2), C)).astype(F)
    bias = rng.standard_normal(C).astype(F)
    for relu in (True, False):
        want = np_wino_output(m, bias, B, H, W, relu)
        same(N(ops.winograd_output(T(gpu, m), T(gpu, bias), B, H, W, relu, pool=False)), want, "winograd output transform")
        pooled = want.reshape(B, H // 2, 2, W // 2, 2, C).max(axis=(2, 4))
        same(N(ops.winograd_output(T(gpu, m), T(gpu, bias), B, H, W, relu, pool=True)), pooled, "winograd output + pool")
    with pytest.raises(ValueError):
        ops.winograd_input(T(gpu, x[:, :H - 1])) if H > 2 else ops.winograd_input(T(gpu, x[:, :1]))


@pytest.mark.parametrize("shape,cout", [((2, 12, 16, 64), 32), ((1, 30, 40, 256), 512), ((1, 6, 6, 512), 512)])
def test_winograd_convolution_matches_direct(gpu, shape, cout):
    """F(2x2,3x3) end to end (transform kernels + library batched GEMM) against a float64 direct
    convolution: all-f32 arithmetic, error of the order of the direct f32 convolution's own."""
    import torch
    from posecnn_amd import ops
    torch.backends.cuda.matmul.allow_tf32 = False
    rng = np.random.default_rng(62)
    B, H, W, C = shape
    x = np.maximum(rng.standard_normal(shape), 0).astype(F)                  # post-ReLU activations
    w = (rng.standard_normal((cout, C, 3, 3)) * np.sqrt(2.0 / (9 * C))).astype(F)
    b = rng.standard_normal(cout).astype(F)
    xt, wt, bt = T(gpu, x), T(gpu, w), T(gpu, b)
    u = ops.winograd_filter(wt)
    # G g G^T of the filter, float64 (the filter transform is host-side plumbing)
    assert np.allclose(N(u).reshape(4, 4, C, cout)[1, 2], np.einsum("r,ocrs,s->co", GM[1], w.astype(np.float64), GM[2]), atol=1e-6)
    y = N(ops.conv3x3_winograd(xt, u, bt, relu=True))
    ref = torch.nn.functional.conv2d(xt.double().permute(0, 3, 1, 2), wt.double(), bt.double(), padding=1).permute(0, 2, 3, 1)
    ref = np.maximum(ref.cpu().numpy(), 0)
    lib = np.maximum(N(torch.nn.functional.conv2d(xt.permute(0, 3, 1, 2), wt, bt, padding=1).permute(0, 2, 3, 1)), 0)
    scale = np.abs(ref).max()
    err_w, err_l = np.abs(y - ref).max() / scale, np.abs(lib - ref).max() / scale
    assert err_w < 2e-5, (err_w, err_l)
    assert err_w < 20 * max(err_l, 1e-7), (err_w, err_l)
    yp = N(ops.conv3x3_winograd(xt, u, bt, relu=True, pool=True))
    same(yp, y.reshape(B, H // 2, 2, W // 2, 2, cout).max(axis=(2, 4)), "pooled winograd conv")


# ---- F(4x4,3x3) ---------------------------------------------------------------------------------------
def np_bt6(d):
    f4, f5, f2 = F(4), F(5), F(2)
    a = d[4] - f4 * d[2]; b = d[3] - f4 * d[1]
    c = d[4] - d[2]; e = f2 * (d[3] - d[1])
    return [(f4 * d[0] - f5 * d[2]) + d[4], a + b, a - b, c + e, c - e, (f4 * d[1] - f5 * d[3]) + d[5]]


def np_at6(m):
    s = m[1] + m[2]; d = m[1] - m[2]; S = m[3] + m[4]; D = m[3] - m[4]
    return [(m[0] + s) + S, d + F(2) * D, s + F(4) * S, (d + F(8) * D) + m[5]]


def np_wino43_input(x):
    B, H, W, C = x.shape
    Ht, Wt = (H + 3) // 4, (W + 3) // 4
    xp = np.zeros((B, 4 * Ht + 2, 4 * Wt + 2, C), F); xp[:, 1:H + 1, 1:W + 1] = x
    d = [[xp[:, r:r + 4 * Ht:4, s:s + 4 * Wt:4] for s in range(6)] for r in range(6)]       # d[r][s]: [B,Ht,Wt,C]
    tmp = [np_bt6([d[r][s] for r in range(6)]) for s in range(6)]                           # tmp[s][i]
    v = [[None] * 6 for _ in range(6)]
    for i in range(6):
        row = np_bt6([tmp[s][i] for s in range(6)])
        for j in range(6):
            v[i][j] = row[j]
    return np.stack([v[i][j] for i in range(6) for j in range(6)]).reshape(36, B * Ht * Wt, C)


def np_wino43_output(m, bias, B, H, W, relu):
    C = m.shape[2]
    Ht, Wt = (H + 3) // 4, (W + 3) // 4
    q = m.reshape(6, 6, B, Ht, Wt, C)
    tmp = [np_at6([q[i, j] for i in range(6)]) for j in range(6)]                           # tmp[j][a]
    y = np.zeros((B, 4 * Ht, 4 * Wt, C), F)
    for a in range(4):
        row = np_at6([tmp[j][a] for j in range(6)])
        for e in range(4):
            o = row[e] + bias
            y[:, a::4, e::4] = np.maximum(o, 0) if relu else o
    return y[:, :H, :W]


@pytest.mark.parametrize("shape", [(2, 8, 12, 16), (1, 30, 40, 512), (1, 6, 10, 4), (2, 5, 7, 36), (1, 1, 1, 8),
                                   (2, 22, 38, 64), (1, 17, 9, 128)])   # C % 64 == 0: the blocked tile map
def test_winograd43_transforms_bit_exact(gpu, shape):
    from posecnn_amd import ops
    rng = np.random.default_rng(63)
    B, H, W, C = shape
    x = rng.standard_normal(shape).astype(F)
    same(N(ops.winograd_input(T(gpu, x), tile=4)), np_wino43_input(x), "F(4,3) input transform")
    Tn = B * ((H + 3) // 4) * ((W + 3) // 4)
    m = rng.standard_normal((36, Tn, C)).astype(F)
    bias = rng.standard_normal(C).astype(F)
    for relu in (True, False):
        want = np_wino43_output(m, bias, B, H, W, relu)
        same(N(ops.winograd_output(T(gpu, m), T(gpu, bias), B, H, W, relu, pool=False, tile=4)), want, "F(4,3) output transform")
        if H % 2 == 0 and W % 2 == 0:
            pooled = want.reshape(B, H // 2, 2, W // 2, 2, C).max(axis=(2, 4))
            same(N(ops.winograd_output(T(gpu, m), T(gpu, bias), B, H, W, relu, pool=True, tile=4)), pooled, "F(4,3) output + pool")
            y2, yp2 = ops.winograd43_output_both(T(gpu, m), T(gpu, bias), B, H, W, relu)
            same(N(y2), want, "F(4,3) output (both)"); same(N(yp2), pooled, "F(4,3) pooled (both)")
    if H % 2:
        with pytest.raises(ValueError):
            ops.winograd_output(T(gpu, m), T(gpu, bias), B, H, W, True, pool=True, tile=4)
        with pytest.raises(ValueError):
            ops.winograd43_output_both(T(gpu, m), T(gpu, bias), B, H, W, True)


@pytest.mark.parametrize("shape,cout", [((2, 12, 16, 64), 32), ((1, 30, 40, 256), 512), ((1, 6, 6, 512), 512), ((1, 60, 80, 128), 128)])
def test_winograd43_convolution_accuracy(gpu, shape, cout):
    """F(4x4,3x3) end to end against a float64 direct convolution; its f32 error is reported next to
    F(2x2,3x3)'s and the library direct convolution's, and bounded."""
    import torch
    from posecnn_amd import ops
    torch.backends.cuda.matmul.allow_tf32 = False
    rng = np.random.default_rng(64)
    B, H, W, C = shape
    x = np.maximum(rng.standard_normal(shape), 0).astype(F)
    w = (rng.standard_normal((cout, C, 3, 3)) * np.sqrt(2.0 / (9 * C))).astype(F)
    b = rng.standard_normal(cout).astype(F)
    xt, wt, bt = T(gpu, x), T(gpu, w), T(gpu, b)
    ref = torch.nn.functional.conv2d(xt.double().permute(0, 3, 1, 2), wt.double(), bt.double(), padding=1).permute(0, 2, 3, 1)
    ref = np.maximum(ref.cpu().numpy(), 0)
    scale = np.abs(ref).max()
    errs = {}
    for tile in (2, 4):
        y = N(ops.conv3x3_winograd(xt, ops.winograd_filter(wt, tile), bt, relu=True, tile=tile))
        errs[tile] = np.abs(y - ref).max() / scale
    lib = np.maximum(N(torch.nn.functional.conv2d(xt.permute(0, 3, 1, 2), wt, bt, padding=1).permute(0, 2, 3, 1)), 0)
    errs["direct"] = np.abs(lib - ref).max() / scale
    print("relative max error vs float64:", errs)
    assert errs[4] < 5e-5, errs
    yp = N(ops.conv3x3_winograd(xt, ops.winograd_filter(wt, 4), bt, relu=True, pool=True, tile=4))
    y4 = N(ops.conv3x3_winograd(xt, ops.winograd_filter(wt, 4), bt, relu=True, tile=4))
    same(yp, y4.reshape(B, H // 2, 2, W // 2, 2, cout).max(axis=(2, 4)), "pooled F(4,3) conv")


@pytest.mark.parametrize("shape,cout", [((2, 16, 32), 64), ((1, 30, 50), 64), ((1, 5, 7), 128), ((1, 480, 640), 64)])
def test_first_conv_fused_into_winograd_input_transform(gpu, shape, cout):
    """conv1_1 + bias + ReLU evaluated inside conv1_2's F(4x4,3x3) input transform: V must equal
    winograd_input(conv3x3_c3(x)) bit for bit (same fma order, same transform expressions)."""
    from posecnn_amd import ops
    rng = np.random.default_rng(71)
    B, H, W = shape
    x = (rng.standard_normal((B, H, W, 3)) * 50).astype(F)
    w = (rng.standard_normal((3, 3, 3, cout)) * 0.1).astype(F)
    b = rng.standard_normal(cout).astype(F)
    for relu in (True, False):
        want = ops.winograd_input(ops.conv3x3_c3(T(gpu, x), T(gpu, w), T(gpu, b), relu), tile=4)
        got = ops.conv3x3_c3_winograd43(T(gpu, x), T(gpu, w), T(gpu, b), relu)
        same(N(got), N(want), "fused conv1_1 -> V (relu=%s)" % relu)
    with pytest.raises(ValueError):
        ops.conv3x3_c3_winograd43(T(gpu, x), T(gpu, w[..., :48]), T(gpu, b[:48]))


def test_network_first_conv_fusion_is_transparent(gpu):
    """The DSL path: conv1_1 stays a pending layer that conv1_2 consumes; the outputs of conv1_2 and a
    fetch of conv1_1 by name are the same bits as without the fusion."""
    import torch
    from posecnn_amd.networks import vgg16_convs
    rng = np.random.default_rng(72)
    x = T(gpu, (rng.standard_normal((1, 32, 48, 3)) * 50).astype(F))
    outs = []
    nets = []
    for fuse in (True, False):
        net = vgg16_convs("COLOR", 22, 64, (1.0,), 1.0, -1.0, trainable=False, is_train=False, device=gpu)
        net.fuse_first_conv_into_winograd = fuse
        if nets:
            net.vars = nets[0].vars
        nets.append(net)
        net.layers = {"data": x}
        with torch.no_grad():
            (net.feed("data").conv(3, 3, 64, 1, 1, name="conv1_1", c_i=3).conv(3, 3, 64, 1, 1, name="conv1_2", c_i=64)
                .max_pool(2, 2, 2, 2, name="pool1"))
            outs.append((N(net.get_output("pool1")), N(net.get_output("conv1_1")), N(net.get_output("conv1_2"))))
    for a, b_ in zip(outs[0], outs[1]):
        same(a, b_, "fused vs unfused first conv")


@pytest.mark.parametrize("shape,cout,pool,groups", [
    ((1, 8, 16, 64), 64, 0, 1), ((2, 12, 20, 64), 128, 1, 1), ((1, 30, 44, 64), 128, 0, 1), ((1, 10, 6, 64), 256, 1, 1),
    ((3, 5, 7, 64), 64, 0, 1), ((1, 120, 160, 64), 64, 1, 1), ((2, 30, 44, 128), 192, 2, 1), ((1, 34, 18, 128), 64, 0, 1),
    ((5, 9, 9, 64), 64, 0, 1), ((2, 60, 80, 256), 256, 0, 1), ((2, 30, 40, 512), 512, 2, 2), ((4, 16, 16, 512), 512, 0, 2),
    ((2, 24, 20, 256), 512, 1, 2), ((6, 10, 14, 128), 128, 0, 3), ((2, 64, 64, 64), 64, 1, 2),
    # enough tile blocks for the 64-tile / 8-wave variant (>= 1024 workgroups), single and grouped
    ((2, 480, 640, 64), 128, 1, 1), ((2, 478, 638, 64), 128, 0, 2), ((4, 240, 320, 128), 256, 2, 2),
    # launches small enough for the Cin split (S = 8 / 4 / 2), ragged tiles, every pool mode, grouped
    ((1, 14, 18, 512), 512, 1, 1), ((2, 30, 38, 256), 256, 2, 2), ((1, 6, 6, 128), 64, 0, 1), ((2, 10, 14, 512), 128, 0, 2)])
def test_winograd43_mfma_conv_kernel(gpu, shape, cout, pool, groups):
    """The 36 Winograd-domain contractions + output transform in one fp32-MFMA kernel (csrc/wino_mfma.hip)
    against (a) the unfused pair — library batched GEMM + wino43_output_kernel — and (b) a float64 direct
    convolution, per group. Only the summation orders differ (K order inside the MFMA chain, column-wise
    output transform); A^T . A amplifies that rounding difference by up to 8 x 8, so the f32 paths agree to
    ~1e-5 of the output range, each within F(4x4,3x3)'s usual error of the float64 reference. Covers every
    Cin of the trunk, partial tiles, tile counts that are no multiple of the 64-tile workgroup, all three
    pooling modes and grouped (two-tower) launches."""
    import torch
    from posecnn_amd import ops
    torch.backends.cuda.matmul.allow_tf32 = False
    rng = np.random.default_rng(81)
    B, H, W, C = shape
    x = np.maximum(rng.standard_normal(shape), 0).astype(F)
    w = (rng.standard_normal((groups, cout, C, 3, 3)) * np.sqrt(2.0 / (9 * C))).astype(F)
    b = rng.standard_normal((groups, cout)).astype(F)
    xt, wt, bt = T(gpu, x), T(gpu, w), T(gpu, b)
    v = ops.winograd_input(xt, 4)
    ut = torch.stack([ops.winograd_filter(wt[g], 4).transpose(1, 2) for g in range(groups)]).contiguous()
    out = ops.winograd43_conv(v, ut, bt, B, H, W, True, pool, groups)
    got_full, got_pool = (N(out[0]), N(out[1])) if pool == 2 else ((None, N(out)) if pool == 1 else (N(out), None))
    Bg = B // groups
    for g in range(groups):
        sl = slice(g * Bg, (g + 1) * Bg)
        ref = torch.nn.functional.conv2d(xt[sl].double().permute(0, 3, 1, 2), wt[g].double(), bt[g].double(), padding=1).permute(0, 2, 3, 1)
        ref = np.maximum(ref.cpu().numpy(), 0)
        vg = ops.winograd_input(xt[sl].contiguous(), 4)
        want = N(ops.winograd_output(torch.bmm(vg, ops.winograd_filter(wt[g], 4)), bt[g], Bg, H, W, True, False, 4))
        scale = np.abs(ref).max()
        if got_full is not None:
            assert got_full.shape == (B, H, W, cout)
            assert np.abs(got_full[sl] - ref).max() <= 5e-5 * scale, (g, np.abs(got_full[sl] - ref).max() / scale)
            assert np.abs(got_full[sl] - want).max() <= 3e-5 * scale
            assert (np.abs(got_full[sl] - want) > 1e-3 * scale).sum() == 0   # no isolated wrong values
        if got_pool is not None:
            refp = ref.reshape(Bg, H // 2, 2, W // 2, 2, cout).max(axis=(2, 4))
            assert got_pool.shape == (B, H // 2, W // 2, cout)
            assert np.abs(got_pool[sl] - refp).max() <= 5e-5 * scale, (g, np.abs(got_pool[sl] - refp).max() / scale)
    if pool == 2:   # the pooled tensor is exactly the max-pool of the full one
        same(got_pool, got_full.reshape(B, H // 2, 2, W // 2, 2, cout).max(axis=(2, 4)), "pool of own output")
    with pytest.raises(ValueError):   # channels must be multiples of 64
        ops.winograd43_conv(v[:, :, :32].contiguous(), ut[:, :, :, :32].contiguous(), bt, B, H, W)
