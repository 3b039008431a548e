"""GPU parity tests for Hough voting: libposecnn_hip.so (through the C-ABI, via posecnn_amd.ops)
against the CPU oracle on the same seeded inputs. The bar is BIT-EXACT outputs for all five
tensors + the row counts (integer/index work and f32 built from identical IEEE operations)."""
import numpy as np
import pytest

import oracle
from posecnn_amd import config, synth

pytestmark = pytest.mark.gpu
F = np.float32
NAMES = ("top_box", "top_pose", "top_target", "top_weight", "top_domain", "num_rois")


def run_gpu(gpu, label, vertex, ext, meta, gt, is_train, vote_thr, per_thr, skip, **kw):
    import torch
    from posecnn_amd import ops
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(gpu)
    out = ops.hough_voting_gpu_padded(t(label), t(vertex), t(ext), t(meta), t(gt), is_train, vote_thr, per_thr, skip, **kw)
    torch.cuda.synchronize()
    return [o.cpu().numpy() for o in out]


def compare(got, want):
    for name, g, w in zip(NAMES, got, want):
        w = np.ascontiguousarray(w)
        assert g.shape == w.shape, name
        gb, wb = g.view(np.uint32), w.view(np.uint32)
        if not np.array_equal(gb, wb):
            bad = np.argwhere(gb != wb)
            raise AssertionError("%s differs at %s: gpu %s oracle %s (num_rois gpu %s oracle %s)" % (
                name, bad[:4].tolist(), g[tuple(bad[0])], w[tuple(bad[0])], got[5], want[5]))


def both(gpu, label, vertex, ext, meta, gt=None, is_train=0, vote_thr=-1.0, per_thr=0.02, skip=10, label_thr=500):
    want = oracle.hough_voting(label, vertex, ext, meta, gt, is_train, vote_thr, per_thr, skip,
                               label_thr=label_thr, padded=True)
    got = run_gpu(gpu, label, vertex, ext, meta, gt, is_train, vote_thr, per_thr, skip, label_threshold=label_thr)
    compare(got, want)
    return got


def frames(first, B, H=480, W=640, C=22, n_obj=5):
    K = config.DEMO_INTRINSICS.copy()
    K[:2] *= W / 640.0
    label, vertex, fr = synth.make_batch(first, B, H=H, W=W, C=C, n_obj=n_obj, K=K)
    meta = np.stack([config.make_meta_data(K)] * B)
    return label, vertex, meta, fr


def test_full_size_single_frame(gpu):
    label, vertex, meta, fr = frames(0, 1)
    got = both(gpu, label, vertex, config.LOV_EXTENTS, meta)
    n = int(got[5][1])
    assert n == 5
    # one detection per generated object; unoccluded objects are localised to a few pixels
    assert sorted(int(r[1]) for r in got[0][:n]) == sorted(o[0] for o in fr[0]["objects"])
    near = [abs((r[2] + r[4]) / 2 - o[1]) < 8 and abs((r[3] + r[5]) / 2 - o[2]) < 8
            for r in got[0][:n] for o in fr[0]["objects"] if o[0] == int(r[1])]
    assert sum(near) >= 3


def test_full_size_batch4_row_order(gpu):
    label, vertex, meta, _ = frames(20, 4)
    got = both(gpu, label, vertex, config.LOV_EXTENTS, meta)
    n = int(got[5][1])
    keys = [(int(r[0]), int(r[1])) for r in got[0][:n]]
    assert keys == sorted(keys)  # canonical (image, class slot) order


def test_batch16_capacity_8_per_image(gpu):
    label, vertex, meta, _ = frames(40, 16, H=240, W=320, n_obj=9)
    got = both(gpu, label, vertex, config.LOV_EXTENTS, meta, label_thr=150)
    per_image = np.bincount(got[0][:int(got[5][1]), 0].astype(int), minlength=16)
    assert per_image.max() <= 8  # index_size = MAX_ROI / batch_size (.cu.cc:733)


@pytest.mark.parametrize("skip", [1, 7, 10])
def test_skip_pixels(gpu, skip):
    label, vertex, meta, _ = frames(60, 2, H=120, W=160, C=8, n_obj=3)
    both(gpu, label, vertex, config.LOV_EXTENTS[:8] * 1.3, meta, skip=skip, label_thr=100)


@pytest.mark.parametrize("hw", [(61, 77), (33, 2049), (480, 64), (17, 19)])
def test_ragged_sizes(gpu, hw):
    H, W = hw
    label, vertex, meta, _ = frames(70, 2, H=H, W=W, C=6, n_obj=2)
    both(gpu, label, vertex, config.LOV_EXTENTS[:6] * 2, meta, skip=3, label_thr=30)


@pytest.mark.parametrize("vote_thr,per_thr", [(3.0, 0.0005), (20.0, 0.02), (1.0, 0.0)])
def test_vote_threshold_local_maxima(gpu, vote_thr, per_thr):
    label, vertex, meta, _ = frames(80, 2, H=120, W=160, C=8, n_obj=3)
    got = both(gpu, label, vertex, config.LOV_EXTENTS[:8] * 1.3, meta, vote_thr=vote_thr, per_thr=per_thr, skip=5, label_thr=100)
    assert int(got[5][0]) >= 1


def test_vote_threshold_full_size(gpu):
    label, vertex, meta, _ = frames(90, 2)
    got = both(gpu, label, vertex, config.LOV_EXTENTS, meta, vote_thr=50.0, per_thr=0.002)
    assert int(got[5][1]) >= 5


def test_vote_threshold_many_maxima_hit_capacity(gpu):
    # a noisy direction field has hundreds of 7x7 local maxima above a low threshold; only the
    # first MAX_ROI / B in ascending (slot, cell) order may be emitted (.cu.cc:377-379, 773-774)
    H, W, C = 96, 128, 3
    rng = np.random.default_rng(5)
    label = np.zeros((2, H, W), np.int32); label[:, 10:80, 10:110] = 1; label[1, 60:90, 5:120] = 2
    ang = rng.uniform(0, 2 * np.pi, (2, H, W))
    vertex = np.zeros((2, H, W, 3 * C), F)
    for c in (1, 2):
        vertex[..., 3 * c] = np.cos(ang); vertex[..., 3 * c + 1] = np.sin(ang); vertex[..., 3 * c + 2] = np.log(0.9)
    ext = np.full((C, 3), 0.2, F)
    meta = np.stack([config.make_meta_data(config.DEMO_INTRINSICS)] * 2)
    got = both(gpu, label, vertex, ext, meta, vote_thr=2.0, per_thr=0.0, skip=4, label_thr=100)
    assert int(got[5][1]) == 128
    assert np.all(got[0][:64, 0] == 0) and np.all(got[0][64:128, 0] == 1)


def test_train_mode_targets_and_jitter(gpu):
    label, vertex, meta, fr = frames(100, 2, H=240, W=320, n_obj=4)
    rng = np.random.default_rng(0)
    gts = []
    for n in range(2):
        for (cls, cx, cy, z) in fr[n]["objects"]:
            q = synth.random_unit_quats(rng, 1)[0]
            K = fr[n]["K"]
            gts.append([n, cls, 0, 0, 0, 0, q[0], q[1], q[2], q[3], (cx - K[0, 2]) / K[0, 0] * z, (cy - K[1, 2]) / K[1, 1] * z, z])
    gts.append([1, 3, 0, 0, 0, 0, 1, 0, 0, 0, 5.0, 5.0, 1.0])  # far-away gt: IoU 0
    gt = np.array(gts, F)
    got = both(gpu, label, vertex, config.LOV_EXTENTS, meta, gt=gt, is_train=1, label_thr=150)
    n = int(got[5][1])
    assert n % 9 == 0 and n >= 18
    assert got[3][:n].sum() > 0          # some targets assigned
    assert np.all(got[4][:n] == 0)
    got2 = both(gpu, label, vertex, config.LOV_EXTENTS, meta, gt=None, is_train=1, label_thr=150)
    assert np.all(got2[4][:int(got2[5][1])] == 1)  # no gt -> domain 1


def test_degenerate_inputs(gpu):
    H, W, C = 96, 128, 5
    ext = config.LOV_EXTENTS[:C] * 2
    meta = np.stack([config.make_meta_data(config.DEMO_INTRINSICS)] * 2)
    # (a) all background -> single dummy row
    label = np.zeros((2, H, W), np.int32)
    vertex = np.random.default_rng(1).standard_normal((2, H, W, 3 * C)).astype(F)
    got = both(gpu, label, vertex, ext, meta, label_thr=50)
    assert tuple(got[5]) == (1, 0)
    # (b) classes at or just above the threshold
    label[0, :5, :10] = 1      # 50 px: not > 50
    label[1, :5, :10] = 2
    label[1, 5, 0] = 2         # 51 px
    got = both(gpu, label, vertex, ext, meta, label_thr=50, skip=2)
    assert tuple(got[5]) == (1, 1)
    # (c) out-of-range labels are ignored
    label[0, 20:40, 20:60] = 7
    label[0, 40:60, 20:60] = -3
    label[0, 60:90, 20:100] = 4
    both(gpu, label, vertex, ext, meta, label_thr=50, skip=2)
    # (d) zero direction field: no votes anywhere -> ROI at the origin with score 0
    vertex0 = np.zeros_like(vertex)
    got = both(gpu, label, vertex0, ext, meta, label_thr=50, skip=2)
    assert np.all(got[0][:int(got[5][1]), 2:] == 0)


def test_non_finite_and_extreme_vertex_values(gpu):
    H, W, C = 96, 128, 4
    rng = np.random.default_rng(2)
    label = np.zeros((1, H, W), np.int32)
    label[0, 10:50, 10:70] = 1
    label[0, 50:90, 40:120] = 2
    label[0, 5:30, 80:125] = 3
    vertex = rng.standard_normal((1, H, W, 3 * C)).astype(F)
    specials = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 1e-30, 1e-42, 1e30, 3e38, -1e-20, 1e-18, 1e18], F)
    m = rng.random((1, H, W, 3 * C)) < 0.15
    vertex[m] = rng.choice(specials, size=int(m.sum()))
    ext = config.LOV_EXTENTS[:C] * 2
    meta = config.make_meta_data(config.DEMO_INTRINSICS)[None]
    both(gpu, label, vertex, ext, meta, label_thr=100, skip=3)
    both(gpu, label, vertex, ext, meta, label_thr=100, skip=3, vote_thr=1.0, per_thr=0.0)


def test_threshold_boundary_stress(gpu):
    """Directions chosen so that many (pixel, cell) pairs sit exactly on cos = 0.9 or within an
    ulp of it: the filtered predicate must agree with the exact one."""
    H, W, C = 128, 160, 3
    label = np.zeros((1, H, W), np.int32)
    label[0, 30:100, 30:130] = 1
    vertex = np.zeros((1, H, W, 3 * C), F)
    rng = np.random.default_rng(3)
    # integer direction pairs (a, b): cells at k*(a', b') have rational cosines; mix in pairs whose
    # cosine to a grid direction is 0.9 +- few ulp: (u,v) = normalize(rot(grid_dir, +-acos(0.9f)))
    yy, xx = np.mgrid[30:100, 30:130]
    gd = rng.integers(-12, 13, size=(2,) + yy.shape).astype(np.float64)
    gd[0][(gd[0] == 0) & (gd[1] == 0)] = 1
    ang = np.arctan2(gd[1], gd[0]) + rng.choice([-1, 1], size=yy.shape) * np.arccos(np.float64(F(0.9)))
    scale = rng.choice([1.0, 0.37, 2.5, 1e-3, 1e3], size=yy.shape)
    vertex[0, 30:100, 30:130, 3] = (np.cos(ang) * scale).astype(F)
    vertex[0, 30:100, 30:130, 4] = (np.sin(ang) * scale).astype(F)
    vertex[0, ..., 5] = np.log(0.8)
    ext = np.full((C, 3), 0.15, F)
    meta = config.make_meta_data(config.DEMO_INTRINSICS)[None]
    want = oracle.hough_voting(label, vertex, ext, meta, None, 0, 1.0, 0.0, 1, label_thr=100, padded=True, want_hs=True)
    got = run_gpu(gpu, label, vertex, ext, meta, None, 0, 1.0, 0.0, 1, label_threshold=100)
    compare(got, want[:6])
    both(gpu, label, vertex, ext, meta, skip=1, label_thr=100)


def test_batch_above_max_roi_has_zero_capacity(gpu):
    B, H, W, C = 130, 32, 32, 3
    label = np.ones((B, H, W), np.int32)
    vertex = np.zeros((B, H, W, 3 * C), F)
    ext = np.full((C, 3), 0.1, F)
    meta = np.tile(config.make_meta_data(config.DEMO_INTRINSICS), (B, 1))
    got = both(gpu, label, vertex, ext, meta, label_thr=100)
    assert tuple(got[5]) == (1, 0)


def test_deterministic_and_graph_capturable(gpu):
    import torch
    from posecnn_amd import ops
    label, vertex, meta, _ = frames(120, 4, H=240, W=320, n_obj=4)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu)
    args = (t(label), t(vertex), t(config.LOV_EXTENTS), t(meta), None, 0, -1.0, 0.02, 10)
    ref = [o.clone() for o in ops.hough_voting_gpu_padded(*args, label_threshold=150)]
    for _ in range(3):
        out = ops.hough_voting_gpu_padded(*args, label_threshold=150)
        for a, b in zip(out, ref):
            assert torch.equal(a, b)
    # no host round trips / device syncs inside the call: it can be captured and replayed
    ws = ops.Workspace()
    outbuf = ops.hough_voting_gpu_padded(*args, label_threshold=150, workspace=ws)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ops.hough_voting_gpu_padded(*args, label_threshold=150, workspace=ws, out=outbuf)  # warm on s
        s.synchronize()
        with torch.cuda.graph(g, stream=s):
            ops.hough_voting_gpu_padded(*args, label_threshold=150, workspace=ws, out=outbuf)
    for o in outbuf:
        o.fill_(7)
    g.replay()
    torch.cuda.synchronize()
    for a, b in zip(outbuf, ref):
        assert torch.equal(a, b)


def test_relabel_covariance_property(gpu):
    """Size-independent property at full size: permuting class ids (labels, vertex channel triples
    and extents together) permutes the class column and nothing else."""
    label, vertex, meta, _ = frames(130, 2)
    C = 22
    perm = np.arange(C); perm[1:] = np.random.default_rng(4).permutation(np.arange(1, C))  # new id of old class
    label2 = perm[label].astype(np.int32)
    vertex2 = np.empty_like(vertex)
    ext2 = np.empty_like(config.LOV_EXTENTS)
    for c in range(C):
        vertex2[..., 3 * perm[c]:3 * perm[c] + 3] = vertex[..., 3 * c:3 * c + 3]
        ext2[perm[c]] = config.LOV_EXTENTS[c]
    a = run_gpu(gpu, label, vertex, config.LOV_EXTENTS, meta, None, 0, -1.0, 0.02, 10)
    b = run_gpu(gpu, label2, vertex2, ext2, meta, None, 0, -1.0, 0.02, 10)
    na, nb = int(a[5][1]), int(b[5][1])
    assert na == nb == 10
    ra = {(int(r[0]), int(perm[int(r[1])])): (r[2:].tobytes(), p.tobytes()) for r, p in zip(a[0][:na], a[1][:na])}
    rb = {(int(r[0]), int(r[1])): (r[2:].tobytes(), p.tobytes()) for r, p in zip(b[0][:nb], b[1][:nb])}
    assert ra == rb


def gpu_hough_space(gpu, label, vertex, ext, meta, vote_thr, skip, label_thr):
    """Run the op with threshold_vote > 0 and read the materialised Hough space out of the
    workspace (pcnn_hough_voting_debug_layout)."""
    import ctypes
    import torch
    from posecnn_amd import _lib, ops
    B, H, W = label.shape
    C = vertex.shape[3] // 3
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu)
    ws = ops.Workspace()
    out = ops.hough_voting_gpu_padded(t(label), t(vertex), t(ext), t(meta), None, 0, vote_thr, 0.0, skip,
                                      workspace=ws, label_threshold=label_thr)
    torch.cuda.synchronize()
    offs = (ctypes.c_size_t * 8)()
    _lib.check("debug_layout", _lib.lib().pcnn_hough_voting_debug_layout(B, H, W, C, float(vote_thr), skip, 0, offs))
    raw = ws._buf.cpu().numpy()
    hs = raw[offs[0]:offs[0] + 4 * B * (C - 1) * H * W].view(np.float32).reshape(B, C - 1, H * W)
    slots = raw[offs[3]:offs[3] + 4 * B * C].view(np.int32).reshape(B, C)
    nslots = raw[offs[4]:offs[4] + 4 * B].view(np.int32)
    return hs, slots, nslots, [o.cpu().numpy() for o in out]


@pytest.mark.parametrize("case", ["synthetic", "plateau", "boundary", "cone_edge"])
def test_every_hough_cell_matches_oracle(gpu, case):
    if case == "synthetic":
        label, vertex, meta, _ = frames(140, 2, H=240, W=320, n_obj=4)
        ext, skip, label_thr = config.LOV_EXTENTS, 10, 150
    elif case == "plateau":
        H, W, C = 64, 96, 3
        label = np.zeros((2, H, W), np.int32); label[:, 10:50, 10:80] = 1
        vertex = np.zeros((2, H, W, 3 * C), F); vertex[..., 3] = 1.0
        ext = np.full((C, 3), 0.2, F)
        meta = np.stack([config.make_meta_data(config.DEMO_INTRINSICS)] * 2)
        skip, label_thr = 4, 100
    elif case == "cone_edge":
        # vote cones with an edge (nearly) parallel to the Hough rows: |u| / |uv| = inlierThreshold to within a
        # few ulps, so the quadratic of the interval form degenerates (A -> 0, one root at infinity); also the
        # mirror cases |v| / |uv| = 0.9 and exact axis directions. Every cell must still equal the oracle's.
        H, W, C = 96, 128, 3
        rng = np.random.default_rng(11)
        label = np.zeros((1, H, W), np.int32); label[0, 20:76, 24:104] = 1
        vertex = np.zeros((1, H, W, 3 * C), F)
        base = []
        s19 = np.sqrt(0.19)
        for su in (1, -1):
            for sv in (1, -1):
                base += [(0.9 * su, s19 * sv), (s19 * su, 0.9 * sv)]
        base += [(1.0, 0.0), (-1.0, 0.0), (0.0, 1.0), (0.0, -1.0)]
        ys, xs = np.nonzero(label[0])
        for i, (y, x) in enumerate(zip(ys, xs)):
            u, v = base[i % len(base)]
            mag = F(rng.uniform(0.3, 3.0))
            uu, vv = F(u) * mag, F(v) * mag
            for _ in range(int(rng.integers(0, 4))):   # a few ulps off the degenerate direction, either way
                uu = np.nextafter(uu, F(np.inf) if rng.random() < 0.5 else F(-np.inf), dtype=F)
            vertex[0, y, x, 3], vertex[0, y, x, 4] = uu, vv
        vertex[0, ..., 5] = np.log(0.9)
        ext = np.full((C, 3), 0.12, F)
        meta = config.make_meta_data(config.DEMO_INTRINSICS)[None]
        skip, label_thr = 1, 100
    else:
        H, W, C = 128, 160, 3
        rng = np.random.default_rng(3)
        label = np.zeros((1, H, W), np.int32); label[0, 30:100, 30:130] = 1
        vertex = np.zeros((1, H, W, 3 * C), F)
        ang = rng.uniform(0, 2 * np.pi, (70, 100))
        vertex[0, 30:100, 30:130, 3] = np.cos(ang); vertex[0, 30:100, 30:130, 4] = np.sin(ang)
        vertex[0, ..., 5] = np.log(0.8)
        ext = np.full((C, 3), 0.15, F)
        meta = config.make_meta_data(config.DEMO_INTRINSICS)[None]
        skip, label_thr = 1, 100
    B, H, W = label.shape
    C = vertex.shape[3] // 3
    hs, slots, nslots, _ = gpu_hough_space(gpu, label, vertex, ext, meta, 1.0, skip, label_thr)
    want = oracle.hough_voting(label, vertex, ext, meta, None, 0, 1.0, 0.0, skip, label_thr=label_thr, padded=True, want_hs=True)
    whs = want[6]
    for n in range(B):
        cls_list = [c for c in range(1, C) if (label[n] == c).sum() > label_thr]
        assert int(nslots[n]) == len(cls_list) and list(slots[n, :len(cls_list)]) == cls_list
        for s, c in enumerate(cls_list):
            diff = np.flatnonzero(hs[n, s] != whs[n, c])
            assert diff.size == 0, "image %d class %d: %d cells differ, first cell %d gpu %s oracle %s" % (
                n, c, diff.size, diff[0], hs[n, s, diff[0]], whs[n, c, diff[0]])


def test_linemod_stress_size_960x1280(gpu):
    """BASELINE configs[4]: 1280x960 inputs, 13 LINEMOD classes + background (C = 14); intrinsics
    scale with the image (lib/fcn/test.py:130-131). Parity with the oracle at the full size."""
    H, W, C = 960, 1280, 14
    K = config.DEMO_INTRINSICS.copy(); K[:2] *= W / 640.0
    ext = config.LINEMOD_EXTENTS      # data/LINEMOD/extents.txt, first 13 objects (checked against the file in test_datasets.py)
    assert ext.shape == (C, 3)
    label, vertex, fr = synth.make_batch(500, 1, H=H, W=W, C=C, n_obj=4, extents=ext, K=K)
    meta = config.make_meta_data(K)[None]
    got = both(gpu, label, vertex, ext, meta)
    assert int(got[5][1]) == 4
    # Hough windows at this size exceed 600 px: also exercise the local-maximum path once
    both(gpu, label, vertex, ext, meta, vote_thr=200.0, per_thr=0.002)


# ---- fused vertex head -> Hough voting (pcnn_hough_voting_lowres_fwd) -----------
def lowres_case(first, B, H, W, C, n_obj, stride):
    """A 1/stride-resolution vertex field whose bilinear upsampling still looks like a scene:
    subsample the synthetic full-resolution field at the cell centres."""
    label, vertex, meta, _ = frames(first, B, H=H, W=W, C=C, n_obj=n_obj)
    z = np.ascontiguousarray(vertex[:, stride // 2::stride, stride // 2::stride, :])
    bias = (np.random.default_rng(first).standard_normal(3 * C) * 0.01).astype(F)
    return label, z, bias, meta


@pytest.mark.parametrize("shape,k,s,vote_thr,per_thr", [
    ((2, 480, 640, 22, 5), 16, 8, -1.0, 0.02),      # the network's configuration
    ((1, 240, 320, 22, 4), 16, 8, 20.0, 0.002),     # threshold path
    ((3, 120, 160, 8, 3), 4, 2, -1.0, 0.02),        # the other (k, s) pair the graph uses
    ((1, 96, 128, 6, 2), 4, 4, -1.0, 0.02),         # k == s: one tap per axis
])
def test_lowres_vertex_source_equals_materialised_field(gpu, shape, k, s, vote_thr, per_thr):
    """Interpolating only the sampled pixels inside the Hough kernel must give the same bits as
    deconv_bilinear -> hough_voting (both HIP), and as the oracle fed the oracle's deconv."""
    import torch
    from posecnn_amd import ops
    B, H, W, C, n_obj = shape
    label, z, bias, meta = lowres_case(500 + H, B, H, W, C, n_obj, s)
    ext = config.LOV_EXTENTS[:C]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu)
    lt = 500 if H >= 480 else 60
    got = ops.hough_voting_gpu_lowres_padded(t(label), t(z), t(bias), k, s, t(ext), t(meta), None, 0, vote_thr, per_thr, 10,
                                             label_threshold=lt)
    full = ops.deconv_bilinear(t(z), k, s, bias=t(bias))
    ref = ops.hough_voting_gpu_padded(t(label), full, t(ext), t(meta), None, 0, vote_thr, per_thr, 10, label_threshold=lt)
    torch.cuda.synchronize()
    got = [o.cpu().numpy() for o in got]
    compare(got, [o.cpu().numpy() for o in ref])
    want = oracle.hough_voting(label, oracle.deconv_bilinear(z, k, s, None, None, bias, False), ext, meta, None, 0,
                               vote_thr, per_thr, 10, label_thr=lt, padded=True)
    compare(got, want)
    assert int(got[5][1]) >= 1


def test_lowres_argument_checks(gpu):
    import torch
    from posecnn_amd import ops
    label = torch.zeros((1, 60, 80), dtype=torch.int32, device=gpu)
    ext = torch.from_numpy(config.LOV_EXTENTS[:4]).to(gpu)
    meta = torch.from_numpy(config.make_meta_data(config.DEMO_INTRINSICS)[None]).to(gpu)
    z = torch.zeros((1, 15, 20, 12), device=gpu)
    bias = torch.zeros(12, device=gpu)
    out = ops.hough_voting_gpu_lowres_padded(label, z, bias, 8, 4, ext, meta, None, 0, -1.0, 0.02, 10)
    assert out[5].tolist() == [1, 0]                      # empty scene: the single dummy row
    with pytest.raises(ValueError):                       # field does not match label / stride
        ops.hough_voting_gpu_lowres_padded(label, z[:, :14], bias, 8, 4, ext, meta, None, 0, -1.0, 0.02, 10)
    with pytest.raises(ValueError):
        ops.hough_voting_gpu_lowres_padded(label, z, bias[:11], 8, 4, ext, meta, None, 0, -1.0, 0.02, 10)
    with pytest.raises(ValueError):                       # (kernel - stride) odd: rejected by the C-ABI (EINVAL)
        ops.hough_voting_gpu_lowres_padded(label, z, bias, 7, 4, ext, meta, None, 0, -1.0, 0.02, 10)
