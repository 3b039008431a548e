"""GPU parity tests of the ICP slice (csrc/icp.hip behind pcnn_icp_backproject_fwd / pcnn_icp_refine_fwd):
bit-identical to oracle_icp_backproject / oracle_icp_refine on analytic box scenes at 480x640 and on the reference's own
demo depth frames (tests/golden/demo_frames.npz), plus the per-frame driver posecnn_amd.icp.refine_poses."""
import os

import numpy as np
import pytest

import icp_scene as S
import oracle
from posecnn_amd import config
from test_gpu_ops import N, T, same

pytestmark = pytest.mark.gpu
F = np.float32
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _u16(gpu, a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint16)).to(gpu)


def test_backproject_matches_oracle_on_the_demo_frames(gpu):
    from posecnn_amd import icp
    fr = np.load(os.path.join(GOLD, "demo_frames.npz"))
    K = config.DEMO_INTRINSICS
    for f in (0, 3):
        depth, label = fr["depth"][f], fr["label"][f].astype(np.int32)
        for cls in (int(np.unique(label)[1]), 0):
            got = icp.backproject(_u16(gpu, depth), T(gpu, label), cls, K, config.DEMO_FACTOR_DEPTH)
            same(N(got), oracle.icp_backproject(depth, label, cls, K, config.DEMO_FACTOR_DEPTH), "frame %d class %d" % (f, cls))
        got = icp.backproject(_u16(gpu, depth), None, 0, K, config.DEMO_FACTOR_DEPTH)
        same(N(got), oracle.icp_backproject(depth, None, 0, K, config.DEMO_FACTOR_DEPTH), "frame %d unmasked" % f)


@pytest.mark.parametrize("H,W,iters", [(480, 640, 8), (120, 160, 3), (100, 131, 5)])
def test_icp_bit_identical_to_the_oracle_on_box_scenes(gpu, H, W, iters):
    """Two objects of a frame in one call (grid.y), 3- and 4-channel predicted maps, odd sizes (a partial last block)."""
    from posecnn_amd import icp
    K = config.DEMO_INTRINSICS.copy(); K[:2] *= W / 640.0
    rng = np.random.default_rng(H)
    lives, pvs, pns, Ts = [], [], [], []
    for k in range(2):
        T_true = S.pose(S.rot([0.3 + k, 1, 0.2], 0.7 + 0.3 * k), [-0.05 + 0.1 * k, 0.02, 0.7 + 0.1 * k])
        T_init = S.pose(S.rot([1, -1, 0.5], np.radians(2.5)) @ T_true[:, :3], T_true[:, 3] + np.array([0.004, -0.005, 0.006]))
        depth, label, pv, pn = S.scene(T_true, T_init, (0.09, 0.07, 0.05), K, H, W, noise=0.0005, rng=rng)
        live_g = icp.backproject(_u16(gpu, depth), T(gpu, label), 3, K, 10000.0)
        lives.append(live_g); pvs.append(pv); pns.append(pn); Ts.append((T_true, T_init))
    import torch
    live = torch.stack(lives)
    pv, pn = np.stack(pvs), np.stack(pns)
    upd, stats = icp.icp(live, T(gpu, pv), T(gpu, pn), K, iterations=iters, want_stats=True)
    want_u, want_s = oracle.icp_refine(N(live), pv, pn, K, iterations=iters)
    assert want_s[:, 0, 0].min() > 300
    assert np.array_equal(N(upd).view(np.uint64), want_u.view(np.uint64)), np.abs(N(upd) - want_u).max()
    same(N(stats), want_s, "stats")
    for k, (T_true, T_init) in enumerate(Ts):       # and it did its job: degrees -> hundredths, centimetre -> sub-millimetre
        re, te = S.pose_error(S.compose(want_u[k], T_init), T_true)
        assert re < 0.3 and te < 1e-3, (k, re, te)
    pad = lambda a: np.concatenate([a, np.full(a.shape[:-1] + (1,), 1.0, F)], axis=-1)
    upd4 = icp.icp(live, T(gpu, pad(pv)), T(gpu, pad(pn)), K, iterations=iters)
    assert np.array_equal(N(upd4).view(np.uint64), want_u.view(np.uint64))


def test_icp_on_a_real_depth_frame(gpu):
    """Live data = a demo depth frame of the reference (one segmented object); predicted maps = that same surface moved by a
    known rigid transform (vertices + normals from the depth gradients): ICP must undo the transform, GPU == oracle."""
    from posecnn_amd import icp
    fr = np.load(os.path.join(GOLD, "demo_frames.npz"))
    K = config.DEMO_INTRINSICS
    depth, label = fr["depth"][1], fr["label"][1].astype(np.int32)
    cls = max((c for c in np.unique(label) if c), key=lambda c: (label == c).sum())
    live = oracle.icp_backproject(depth, label, int(cls), K, config.DEMO_FACTOR_DEPTH)
    # normals of the live surface by central differences of the vertex map
    dx = np.zeros_like(live); dy = np.zeros_like(live)
    dx[:, 1:-1] = live[:, 2:] - live[:, :-2]
    dy[1:-1] = live[2:] - live[:-2]
    nrm = np.cross(dy, dx)          # points towards the camera (-z) for a surface facing it
    ln = np.linalg.norm(nrm, axis=-1, keepdims=True)
    valid = (live[..., 2] > 0.25) & (ln[..., 0] > 0)
    for s in ((0, 1), (0, -1), (1, 0), (-1, 0)):      # neighbours used by the differences must lie on the object too
        valid &= np.roll(live[..., 2], s, axis=(0, 1)) > 0.25
    nrm = np.where(ln > 0, nrm / np.maximum(ln, 1e-20), 0)
    nrm[nrm[..., 2] > 0] *= -1
    # "predicted" maps: the surface displaced by D^-1, so that the update ICP must find is D
    c = live[valid].mean(0)
    Rd = S.rot([0.2, 1, 0.1], np.radians(1.5))
    D = S.pose(Rd, c - Rd @ c + np.array([0.003, -0.002, 0.004]))
    Dinv = S.pose(Rd.T, -Rd.T @ D[:, 3])
    pv = np.where(valid[..., None], live @ Dinv[:, :3].T + Dinv[:, 3], 0).astype(F)
    pn = np.where(valid[..., None], nrm @ Dinv[:, :3].T, 0).astype(F)
    # the maps live on the pixel grid of the UN-displaced surface (a renderer would resample): good enough for 1.5 degrees
    want_u, want_s = oracle.icp_refine(live[None], pv[None], pn[None], K, iterations=10)
    upd, stats = icp.icp(T(gpu, live[None]), T(gpu, pv[None]), T(gpu, pn[None]), K, iterations=10, want_stats=True)
    assert np.array_equal(N(upd).view(np.uint64), want_u.view(np.uint64))
    same(N(stats), want_s, "stats")
    assert want_s[0, 0, 0] > 2000
    re0, te0 = S.pose_error(S.pose(np.eye(3), np.zeros(3)), D)
    re, te = S.pose_error(want_u[0], D)
    assert re < 0.35 * re0 and te < 0.35 * te0, ((re0, te0), (re, te))


def test_refine_poses_driver(gpu):
    """posecnn_amd.icp.refine_poses: the per-frame loop of Synthesizer::solveICP around df::icp — ROIs with class id <= 0 or
    fewer than 400 label pixels are skipped (synthesize.cpp:2093, :2152), the others get update * T_co as quaternion + translation."""
    from posecnn_amd import icp
    H, W = 240, 320
    K = config.DEMO_INTRINSICS.copy(); K[:2] *= W / 640.0
    half = {3: (0.09, 0.07, 0.05), 5: (0.05, 0.05, 0.08)}
    T_true = {3: S.pose(S.rot([0.3, 1, 0.2], 0.7), [-0.08, 0.02, 0.7]), 5: S.pose(S.rot([0.3, 1, 0.2], 0.9), [0.1, -0.03, 0.8])}     # (three faces in view each: a well-posed problem)
    depth = np.zeros((H, W), np.uint16); label = np.zeros((H, W), np.int32)
    for cls in (3, 5):
        v, _, hit = S.render_box(T_true[cls], half[cls], K, H, W)
        depth[hit] = np.round(v[..., 2][hit] * 10000).astype(np.uint16); label[hit] = cls
    label[0, :50] = 9                                   # a sliver of a third class: too few pixels
    rois = np.array([[0, 3, 0, 0, 1, 1, 1], [0, 5, 0, 0, 1, 1, 1], [0, 9, 0, 0, 1, 1, 1], [0, 0, 0, 0, 1, 1, 1]], F)
    poses = np.zeros((4, 7), F)
    init = {}
    for i, cls in enumerate((3, 5)):
        Ti = S.pose(S.rot([1, -1, 0.5], np.radians(2.0)) @ T_true[cls][:, :3], T_true[cls][:, 3] + np.array([0.004, -0.003, 0.005]))
        init[cls] = Ti
        poses[i, :4] = icp.mat2quat(Ti[:, :3]); poses[i, 4:] = Ti[:, 3]
    poses[2:, 0] = 1
    calls = []

    def render(cls, Tco):
        calls.append(cls)
        v, n, _ = S.render_box(Tco, half[cls], K, H, W)
        return v, n

    out = icp.refine_poses(label, depth, K, 10000.0, rois, poses, render, iterations=8, device=gpu)
    assert calls == [3, 5] and not out[2:].any()
    from posecnn_amd.pose_error import quat2mat
    for i, cls in enumerate((3, 5)):
        Tn = S.pose(quat2mat(out[i, :4]), out[i, 4:])
        re0, te0 = S.pose_error(init[cls], T_true[cls])
        re, te = S.pose_error(Tn, T_true[cls])
        assert re < 0.15 * re0 and te < 0.1 * te0, (cls, (re0, te0), (re, te))
        assert abs(np.linalg.norm(out[i, :4]) - 1) < 1e-6 and out[i, 0] >= 0
