"""Known-answer tests for the CPU oracle, hand-derived from the reference kernel sources
(SURVEY.md §8a "Hand-derived known answers"). The reference ships no tests or golden vectors
(SURVEY.md §4), so these pin the oracle to the source text."""
import numpy as np
import pytest

import oracle
from posecnn_amd import config

F = np.float32
META = config.make_meta_data(config.DEMO_INTRINSICS)
EXT = config.LOV_EXTENTS


def test_project_box_known_answers():
    # hough_voting_gpu_op.cu.cc:84-120 with the demo intrinsics (tools/demo.py:100), class 1 extents
    assert oracle.project_box(1, EXT, META, 0.6) == pytest.approx(128.3848, abs=2e-4)
    assert oracle.project_box(1, EXT, META, 1.0) == pytest.approx(73.2118, abs=2e-4)
    assert oracle.project_box(1, EXT, META, 1.2) == pytest.approx(60.3194, abs=2e-4)
    assert oracle.project_box(11, EXT, META, 0.8) == pytest.approx(157.199, abs=2e-3)


def test_project_box_by_hand():
    # one value recomputed step by step in float32
    e = EXT[1]
    hx, hy, hz = F(e[0]) * F(0.5), F(e[1]) * F(0.5), F(e[2]) * F(0.5)
    d = F(1.0)
    fx, px, fy, py = F(META[0]), F(META[2]), F(META[4]), F(META[5])
    xs, ys = [], []
    for Z in (hz + d, -hz + d):
        for sy in (1, -1):
            for sx in (1, -1):
                xs.append(F(fx * F(F(sx) * hx / Z)) + px)
                ys.append(F(fy * F(F(sy) * hy / Z)) + py)
    w = F(max(xs) - min(xs)) + F(1)
    h = F(max(ys) - min(ys)) + F(1)
    assert oracle.project_box(1, EXT, META, 1.0) == F(max(w, h) * F(0.6))


def test_expf_matches_correctly_rounded_double():
    xs = np.concatenate([np.linspace(-20, 20, 4001), [-104, -103.5, -87.4, -87.3, 0.0, 88.7, 88.8, 130, 200, -200]]).astype(F)
    got = oracle.expf(xs)
    with np.errstate(over="ignore", under="ignore"):
        want = np.exp(xs.astype(np.float64)).astype(F)
    # double exp + one rounding: identical except (rarely) at a double-rounding boundary
    mism = np.flatnonzero(got != want)
    assert len(mism) <= 2
    assert np.all(np.abs(got[mism].astype(np.float64) - want[mism]) <= np.spacing(want[mism]))
    assert np.isnan(oracle.expf(np.array([np.nan], F))[0])
    assert oracle.expf(np.array([200], F))[0] == np.inf
    assert oracle.expf(np.array([-200], F))[0] == 0


def test_exp_softmax_canonical_sequence():
    """The all-f32 exp of the softmax layers: the C oracle and the numpy restatement agree bit for
    bit, the value is within 2 ulp of the correctly rounded exp on the softmax domain (x <= 0) and
    the edge cases behave (NaN passes through, clamped tails, subnormal results)."""
    import np_ref
    rng = np.random.default_rng(5)
    xs = np.concatenate([-rng.exponential(4.0, 60000), np.linspace(-110, 0, 20001), rng.uniform(-1, 1, 5000),
                         [0.0, -0.0, -87.3, -87.4, -88.0, -100.0, -103.9, -104.0, -150.0, 1.0, 88.0, 100.0]]).astype(F)
    got = oracle.exp_softmax(xs)
    with np.errstate(all="ignore"):
        same_bits = got.view(np.uint32) == np_ref.exp_softmax_f32(xs).view(np.uint32)
        want = np.exp(np.clip(xs.astype(np.float64), -104, 88))
    assert same_bits.all()
    normal = want >= 1.2e-38
    ulp = np.spacing(want.astype(F)).astype(np.float64)
    assert np.max(np.abs(got[normal].astype(np.float64) - want[normal]) / ulp[normal]) <= 2.0
    assert np.all(np.abs(got[~normal].astype(np.float64) - want[~normal]) <= 2 * 1.4e-45 + 4e-7 * want[~normal])
    assert np.all(got[xs == 0] == 1.0)
    assert np.isnan(oracle.exp_softmax(np.array([np.nan], F))[0])
    assert np.all(np.diff(oracle.exp_softmax(np.linspace(-104, 0, 4001).astype(F))) >= 0)   # monotone on a grid


def _one_object(H=64, W=96, C=3, cls=1, z=1.0, label_thr_pixels=600):
    label = np.zeros((1, H, W), np.int32)
    label[0, 20:50, 30:70] = cls  # 1200 px
    vertex = np.zeros((1, H, W, 3 * C), F)
    cy, cx = 35.0, 50.0
    yy, xx = np.mgrid[0:H, 0:W]
    ang = np.arctan2(cy - yy, cx - xx)
    vertex[0, :, :, 3 * cls] = np.cos(ang)
    vertex[0, :, :, 3 * cls + 1] = np.sin(ang)
    vertex[0, :, :, 3 * cls + 2] = np.log(z)
    return label, vertex


def test_hough_single_object_center_and_pose():
    label, vertex = _one_object()
    ext = np.zeros((3, 3), F)
    ext[1] = (0.1, 0.1, 0.1)
    box, pose, target, weight, domain = oracle.hough_voting(label, vertex, ext, META[None], None, 0, -1.0, 0.02, 10)
    assert box.shape == (1, 7)
    assert box[0, 0] == 0 and box[0, 1] == 1
    cxs, cys = (box[0, 2] + box[0, 4]) / 2, (box[0, 3] + box[0, 5]) / 2
    assert abs(cxs - 50) <= 1 and abs(cys - 35) <= 1
    # pose = (1,0,0,0, rx*d, ry*d, d) with d = mean exp(log z) = 1  (.cu.cc:567-573)
    assert pose[0, 0] == 1 and np.all(pose[0, 1:4] == 0)
    assert pose[0, 6] == pytest.approx(1.0, abs=1e-6)
    assert pose[0, 4] == pytest.approx((cxs - META[2]) / META[0], abs=1e-3)
    assert np.all(target == 0) and np.all(weight == 0) and np.all(domain == 0)


def test_hough_pixel_never_votes_for_its_own_cell():
    # angle_distance at dx=dy=0 is 0/0 = NaN -> comparison false (.cu.cc:32-42,283)
    H, W, C = 8, 8, 2
    label = np.zeros((H, W), np.int32)
    label[3, 4] = 1
    vertex = np.zeros((H, W, 3 * C), F)
    vertex[3, 4, 3] = 1.0  # u = 1: points to +x
    ext = np.zeros((C, 3), F); ext[1] = 0.1
    hs, hd, m = oracle.hough_space(label, vertex, ext, META, 1, 1)
    assert m == 1
    assert hs[3, 4] == 0
    assert hs[3, 5] == 1 and hs[3, 7] == 1       # straight ahead: cos = 1
    assert hs[3, 3] == 0                          # behind: cos = -1
    assert hs[4, 5] == 0                          # 45 degrees: cos = 0.707 < 0.9
    assert hs[4, 7] == 1                          # atan(1/3) = 18.4 deg: cos = 0.949 > 0.9
    # hough_data = (mean depth, 2*bb_height, 2*bb_width): one voter at |dx|=3,|dy|=1 for cell (4,7)
    assert tuple(hd[4, 7]) == (1.0, 2.0, 6.0)
    assert tuple(hd[3, 4]) == (0.0, 0.0, 0.0)


def test_hough_below_label_threshold_gives_dummy_row():
    label, vertex = _one_object()
    label[0, 20:50, 30:70] = 0
    label[0, 20:30, 30:70] = 1  # 400 px <= labelThreshold 500 (hough_voting_gpu_op.cc:357)
    ext = np.zeros((3, 3), F); ext[1] = 0.1
    out = oracle.hough_voting(label, vertex, ext, META[None], None, 0, -1.0, 0.02, 10, padded=True)
    assert tuple(out[5]) == (1, 0)            # one dummy row, zero true rows (.cc:381-383)
    assert np.all(out[0][0] == 0)


def test_hough_empty_space_emits_origin_roi():
    # no pixel can vote (zero direction -> NaN): max_element returns cell 0, ROI (0,0,0,0), score 0
    H, W, C = 32, 32, 2
    label = np.ones((1, H, W), np.int32)
    vertex = np.zeros((1, H, W, 3 * C), F)
    ext = np.zeros((C, 3), F); ext[1] = 0.1
    box, pose, *_ = oracle.hough_voting(label, vertex, ext, META[None], None, 0, -1.0, 0.02, 10)
    assert box.shape[0] == 1
    assert tuple(box[0]) == (0, 1, 0, 0, 0, 0, 0)
    assert tuple(pose[0]) == (1, 0, 0, 0, 0, 0, 0)


def test_hough_capacity_is_max_roi_over_batch():
    # index_size = MAX_ROI / batch_size (.cu.cc:733): with B=64 only 2 maxima per image survive
    B, H, W, C = 64, 32, 32, 5
    label = np.zeros((B, H, W), np.int32)
    for c in range(1, 5):
        label[:, (c - 1) * 8:(c - 1) * 8 + 8, :] = c  # 256 px each
    vertex = np.zeros((B, H, W, 3 * C), F)
    ext = np.full((C, 3), 0.1, F)
    meta = np.tile(META, (B, 1))
    out = oracle.hough_voting(label, vertex, ext, meta, None, 0, -1.0, 0.02, 10, label_thr=100, padded=True)
    assert tuple(out[5]) == (128, 128)
    assert list(out[0][:4, 1]) == [1, 2, 1, 2] and list(out[0][:4, 0]) == [0, 0, 1, 1]


def test_hough_train_rows_and_targets():
    label, vertex = _one_object()
    ext = np.zeros((3, 3), F); ext[1] = (0.04, 0.03, 0.02)  # projects to ~44x33 px at 1 m: IoU > 0.2
    # gt = (batch, cls, box4, quat wxyz, trans3): object at the detected centre, identity rotation
    rx, ry = (50 - META[2]) / META[0], (35 - META[5]) / META[4]
    gt = np.array([[0, 1, 0, 0, 0, 0, 1, 0, 0, 0, rx, ry, 1.0],
                   [0, 2, 0, 0, 0, 0, 0, 1, 0, 0, rx, ry, 1.0]], F)
    box, pose, target, weight, domain = oracle.hough_voting(label, vertex, ext, META[None], gt, 1, -1.0, 0.02, 10)
    assert box.shape == (9, 7)
    ww, hh = box[0, 4] - box[0, 2], box[0, 5] - box[0, 3]
    # jitter order (.cu.cc:476-554)
    signs = [(-1, -1), (1, -1), (-1, 1), (1, 1), (0, -1), (-1, 0), (0, 1), (1, 0)]
    for j, (sx, sy) in enumerate(signs):
        assert box[1 + j, 2] == F(np.float64(box[0, 2]) + sx * 0.05 * np.float64(ww))
        assert box[1 + j, 3] == F(np.float64(box[0, 3]) + sy * 0.05 * np.float64(hh))
        assert box[1 + j, 4] == F(box[1 + j, 2] + ww)
    assert np.all(weight[:, 4:8] == 1) and np.all(target[:, 4:8] == (1, 0, 0, 0))
    assert np.all(weight[:, :4] == 0) and np.all(weight[:, 8:] == 0)
    assert np.all(domain == 0)
    # no gt at all -> domain label 1 (.cu.cc:433-436)
    *_, domain2 = oracle.hough_voting(label, vertex, ext, META[None], None, 1, -1.0, 0.02, 10)
    assert np.all(domain2 == 1)


def test_roi_pool_rounding_known_answers():
    # ROI (100.4, 50.5, 300.6, 250.5): scale 1/16 -> start (6,3), end (19,16), 14x14 bins of 2.0;
    # scale 1/8 -> start (13,6), end (38,31), 26x26, bin 3.7142856 (roundf = half away from zero)
    H, W, C = 30, 40, 1
    data = np.arange(H * W * C, dtype=F).reshape(1, H, W, C)  # value = index: max sits at bin's last cell
    rois = np.array([[0, 3, 100.4, 50.5, 300.6, 250.5, 0.9]], F)
    top, arg = oracle.roi_pool(data, rois, 7, 7, 1.0 / 16, 0)
    for ph in range(7):
        for pw in range(7):
            h_last, w_last = 3 + 2 * ph + 1, 6 + 2 * pw + 1
            assert arg[0, ph, pw, 0] == h_last * W + w_last
            assert top[0, ph, pw, 0] == h_last * W + w_last
    H2, W2 = 60, 80
    data2 = np.arange(H2 * W2, dtype=F).reshape(1, H2, W2, 1)
    top2, arg2 = oracle.roi_pool(data2, rois, 7, 7, 1.0 / 8, 0)
    bin_ = F(26) / F(7)
    for ph in range(7):
        for pw in range(7):
            he = int(np.ceil(F(ph + 1) * bin_)) + 6
            we = int(np.ceil(F(pw + 1) * bin_)) + 13
            assert arg2[0, ph, pw, 0] == (he - 1) * W2 + (we - 1)


def test_roi_pool_edge_cases():
    H, W, C = 6, 6, 2
    data = np.random.default_rng(0).standard_normal((2, H, W, C)).astype(F)
    rois = np.array([
        [0, 1, -50, -50, -20, -20, 0],   # fully outside: every bin empty -> 0 / -1
        [1, 1, 4, 4, 2, 2, 0],           # malformed (end < start) forced to 1x1
        [1, 0, 0, 0, 100, 100, 0],       # oversized: clipped to the image
        [0, 1, 2.5, 2.5, 2.5, 2.5, 0],   # .5 rounds away from zero -> 3
    ], F)
    top, arg = oracle.roi_pool(data, rois, 2, 2, 1.0, 0)
    assert np.all(top[0] == 0) and np.all(arg[0] == -1)
    assert np.all(top[1] == data[1, 4, 4]) and np.all(arg[1][..., 0] == (4 * W + 4) * C)
    # roi 101 px wide: bin 50.5 -> bin (0,0) = [0,51) clipped to the whole image, bin (1,1) = [50,101) -> empty
    assert top[2, 0, 0, 0] == data[1, :, :, 0].max() and top[2, 0, 0, 1] == data[1, :, :, 1].max()
    assert np.all(top[2, 1, 1] == 0) and np.all(arg[2, 1, 1] == -1)
    assert np.all(top[3] == data[0, 3, 3])
    # pool_channel=1 pools only channel roi_cls (roi_pooling_op_gpu.cu.cc:87-88)
    top_c, arg_c = oracle.roi_pool(data, rois[1:2], 2, 2, 1.0, 1)
    assert top_c.shape == (1, 2, 2, 1) and np.all(top_c == data[1, 4, 4, 1])


def test_hard_label_truth_table():
    # hard_label_op_gpu.cu.cc:25-27: out[g] = 1 iff g != -1 and (g > 0 or prob[g] < threshold)
    C = 3
    prob = np.array([[0.2, 0.5, 0.3],   # g=0, p=0.2 <  thr -> 1
                     [0.9, 0.05, 0.05],  # g=0, p=0.9 >= thr -> 0
                     [0.9, 0.05, 0.05],  # g=2 (foreground) -> always 1
                     [0.1, 0.1, 0.8],    # g=-1 -> nothing
                     [0.5, 0.25, 0.25]], F)  # g=0, p == thr -> not < -> 0
    gt = np.array([0, 0, 2, -1, 0], np.int32)
    out = oracle.hard_label(prob, gt, 0.5)
    want = np.zeros((5, C), F)
    want[0, 0] = 1
    want[2, 2] = 1
    assert np.array_equal(out, want)


def _points(C=3, P=40, seed=1):
    rng = np.random.default_rng(seed)
    return rng.uniform(-0.1, 0.1, size=(C, P, 3)).astype(F)


def test_average_distance_identity_and_rotation():
    C, P = 3, 40
    pts = _points(C, P)
    sym = np.zeros(C, F)
    pred = np.zeros((2, 4 * C), F); tgt = np.zeros((2, 4 * C), F); wgt = np.zeros((2, 4 * C), F)
    pred[0, 4:8] = tgt[0, 4:8] = (1, 0, 0, 0); wgt[0, 4:8] = 1
    # identical poses -> distance 0 < margin -> nothing contributes
    loss, diff = oracle.average_distance(pred, tgt, wgt, pts, sym, 0.01)
    assert loss[0] == 0 and np.all(diff == 0)
    # 180 degrees about z for class 1: x2 = (-x,-y,z) -> dist = 4(x^2+y^2); margin 0
    tgt[0, 4:8] = (0, 0, 0, 1)
    loss, diff = oracle.average_distance(pred, tgt, wgt, pts, sym, 0.0)
    R = 2
    want = F(0)
    for p in range(P):
        x, y, z = pts[1, p]
        ex, ey, ez = F(x - (-x)), F(y - (-y)), F(z - z)
        d = F(F(ex * ex) + F(ey * ey)) + F(ez * ez)
        want = F(want + F(np.float64(d) / (2.0 * R * P)))
    assert loss[0] == want
    assert np.all(diff[1] == 0) and np.all(diff[0, :4] == 0) and np.any(diff[0, 4:8] != 0)


def test_average_distance_symmetric_uses_nearest_point():
    C, P = 2, 6
    pts = np.zeros((C, P, 3), F)
    pts[1] = [[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]]
    pred = np.zeros((1, 8), F); tgt = np.zeros((1, 8), F); wgt = np.zeros((1, 8), F)
    pred[0, 4:8] = (1, 0, 0, 0); tgt[0, 4:8] = (0, 0, 0, 1); wgt[0, 4:8] = 1  # 180 deg about z maps the set onto itself
    loss_sym, _ = oracle.average_distance(pred, tgt, wgt, pts, np.array([0, 1], F), 0.0)
    loss_plain, _ = oracle.average_distance(pred, tgt, wgt, pts, np.array([0, 0], F), 0.0)
    assert loss_sym[0] == 0
    assert loss_plain[0] == pytest.approx(4 * 4 / (2.0 * 1 * P), rel=1e-6)  # four points move by 2


def test_backproject_surface_and_empty_voxels():
    B, H, W, Cd, Cl, G = 1, 8, 8, 2, 3, 4
    rng = np.random.default_rng(2)
    data = rng.standard_normal((B, H, W, Cd)).astype(F)
    label = rng.random((B, H, W, Cl)).astype(F)
    depth = np.full((B, H, W, 1), 2.0, F)
    label3d = rng.random((B, G, G, G, Cl)).astype(F)
    K = np.array([[4.0, 0, 4.0], [0, 4.0, 4.0], [0, 0, 1]])
    ident = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]], F)
    # voxel (d,h,w) -> X=(d-2)*0.5+0.25.., Z = w*0.5 + 1.0: only w=2 gives Z1 = 2.0 (on the surface)
    meta = config.make_meta_data(K, voxel_step=(0.5, 0.5, 0.5), voxel_min=(-1.0, -1.0, 1.0),
                                 pose_world2live=ident, pose_live2world=ident)
    td, tl, tf = oracle.backproject(data, label, depth, meta[None], label3d, G, 0, 0.02)
    for d in range(G):
        for h in range(G):
            for w in range(G):
                if w == 2:
                    X, Y, Z = d * 0.5 - 1.0, h * 0.5 - 1.0, 2.0
                    px = int(np.floor((4 * X + 4 * Z) / Z + 0.5)); py = int(np.floor((4 * Y + 4 * Z) / Z + 0.5))
                    assert np.all(tf[0, d, h, w] == 1)
                    assert np.array_equal(td[0, d, h, w], data[0, py, px])
                    assert np.array_equal(tl[0, d, h, w], label[0, py, px])
                else:
                    assert np.all(tf[0, d, h, w] == 0) and np.all(td[0, d, h, w] == 0)
                    assert np.array_equal(tl[0, d, h, w], label3d[0, d, h, w])


def test_softmax_argmax_first_maximum():
    score = np.array([[0, 0, 0, 0], [0, 3, 3, 1], [5, 1, 5, 5], [-1, -2, -0.5, -3]], F)
    prob, lab = oracle.softmax_argmax(score)
    assert list(lab) == [0, 1, 0, 2]
    assert np.allclose(prob.sum(-1), 1, atol=1e-6)
    assert np.all(prob[0] == 0.25)


def test_cpu_kernel_semantics_hough_finds_the_same_objects():
    """H7 (hough_voting_gpu_op.cc:486-758, ray marching) is a different algorithm and no parity
    target; on a clean synthetic frame it must still localise the objects the GPU-kernel semantics
    find (same classes, centres within a few pixels)."""
    import time
    from posecnn_amd import config, synth
    K = config.DEMO_INTRINSICS.copy()
    label, vertex, fr = synth.make_batch(900, 1, H=480, W=640, C=22, n_obj=4, K=K)
    meta = np.stack([config.make_meta_data(K)])
    t0 = time.perf_counter()
    rows = oracle.hough_cpu_kernel(label, vertex, config.LOV_EXTENTS, meta)
    dt = time.perf_counter() - t0
    boxes, poses = oracle.hough_voting(label, vertex, config.LOV_EXTENTS, meta, None, 0, -1.0, 0.02, 10)[:2]
    assert sorted(int(r[1]) for r in rows) == sorted(int(b[1]) for b in boxes)
    by_cls = {int(b[1]): (b, p) for b, p in zip(boxes, poses)}
    for r in rows:
        b, p = by_cls[int(r[1])]
        assert abs((r[2] + r[4]) / 2 - (b[2] + b[4]) / 2) < 12 and abs((r[3] + r[5]) / 2 - (b[3] + b[5]) / 2) < 12
        assert abs(r[13] - p[6]) < 0.1           # mean inlier depth
    assert dt < 5.0
