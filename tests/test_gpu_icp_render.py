"""GPU parity tests of the second ICP slice (csrc/render.hip, the centre / score kernels of csrc/icp.hip, and
posecnn_amd.icp.Synthesizer.icp_python = Synthesizer::solveICP, lib/synthesize/synthesize.cpp:2052-2380, without its nlopt
stage): bit-identical to oracle_render_mesh / oracle_icp_center / oracle_icp_score, and the whole per-frame flow against
the same flow on the checker (tests/icp_scene.solve_icp_reference)."""
import ctypes

import numpy as np
import pytest

import icp_scene as S
import oracle
from posecnn_amd import config
from test_gpu_ops import N, T, same

pytestmark = pytest.mark.gpu
F = np.float32


def scaled_K(W):
    K = config.DEMO_INTRINSICS.copy()
    K[:2] *= W / 640.0
    return K


def gpu_mesh(gpu, v, n, f):
    from posecnn_amd import icp
    return icp.Mesh(v, f, n, device=gpu)


def check_render(gpu, v, n, f, poses, K, H, W, name, model_index=0, depth_range=(0.25, 6.0)):
    from posecnn_amd import icp
    got = icp.render(gpu_mesh(gpu, v, n, f), poses, K, H, W, depth_range, model_index, want=("vertices", "normals", "canonical"))
    want = oracle.render_mesh(v, n, f, poses, K, H, W, depth_range, model_index)
    for key in ("vertices", "normals", "canonical"):
        same(N(got[key]), want[key], "%s %s" % (name, key))
    return want


def test_render_bit_identical_box_and_spheres(gpu):
    """12 large triangles (the workgroup-cooperative path), 1 280 and 20 480 small ones (one thread each), a mesh that mixes
    both; several poses per call; the 480x640 frame and an odd size."""
    rng = np.random.default_rng(5)
    vb, nb, fb = S.box_mesh((0.08, 0.06, 0.05))
    vs, ns, fs = S.icosphere(0.07, 3, scale=(1.0, 0.6, 1.2))
    vl, nl, fl = S.icosphere(0.09, 5)
    poses = np.stack([S.pose(S.rot(rng.standard_normal(3), rng.uniform(0, 3)), [rng.uniform(-0.1, 0.1), rng.uniform(-0.08, 0.08), rng.uniform(0.5, 1.1)])
                      for _ in range(3)])
    for (H, W) in ((480, 640), (101, 131)):
        K = scaled_K(W)
        w = check_render(gpu, vb, nb, fb, poses, K, H, W, "box %dx%d" % (H, W), model_index=2)
        assert np.isfinite(w["vertices"][..., 2]).sum() > 1500
        check_render(gpu, vs, ns, fs, poses, K, H, W, "ellipsoid %dx%d" % (H, W))
    K = scaled_K(640)
    w = check_render(gpu, vl, nl, fl, poses[:2], K, 480, 640, "20480 faces", model_index=20)
    assert np.isfinite(w["vertices"][..., 2]).sum() > 20000
    vm = np.concatenate([vb, vs + np.array([0.0, 0.0, -0.15], F)])
    check_render(gpu, vm, np.concatenate([nb, ns]), np.concatenate([fb, fs + len(vb)]), poses, K, 480, 640, "mixed")


def test_render_bit_identical_on_random_triangulations(gpu):
    """400-point Delaunay sheets (slivers, mixed windings, shared edges at every angle) at three poses each: the atomicMin
    z-buffer and the serial checker agree on every pixel bit for bit."""
    rng = np.random.default_rng(2)
    K = scaled_K(320)
    for seed in range(3):
        v, f = S.random_sheet(10 + seed, tilt=(0.0, 0.0), z0=0.0)
        poses = np.stack([S.pose(S.rot(rng.standard_normal(3), rng.uniform(0, 1.2)), [rng.uniform(-0.05, 0.05), rng.uniform(-0.05, 0.05), rng.uniform(0.5, 0.9)])
                          for _ in range(3)])
        n = np.tile(np.array([[0, 0, -1.0]], F), (len(v), 1))
        w = check_render(gpu, v, n, f, poses, K, 240, 320, "sheet %d" % seed)
        assert np.isfinite(w["vertices"][..., 2]).sum() > 10000


def test_render_edge_cases(gpu):
    """An object half outside the image, one straddling z_near (its near triangles are dropped), one past z_far, a degenerate
    and a repeated-index triangle, an empty mesh, zero poses."""
    from posecnn_amd import icp
    H, W = 120, 160
    K = scaled_K(W)
    v, n, f = S.icosphere(0.1, 3)
    f2 = np.concatenate([f, np.array([[0, 0, 5], [3, 7, 7]], np.int32)])
    v2 = np.concatenate([v, v[:3] * 0 + np.array([0.01, 0.01, 0.0], F)])          # three coincident vertices -> zero-area triangle
    f2 = np.concatenate([f2, np.array([[len(v), len(v) + 1, len(v) + 2]], np.int32)])
    n2 = np.concatenate([n, n[:3]])
    poses = np.stack([S.pose(np.eye(3), [0.22, 0.05, 0.6]),      # right half off-screen
                      S.pose(np.eye(3), [0.0, 0.0, 0.3]),        # front cap in front of z_near = 0.25
                      S.pose(np.eye(3), [0.0, 0.0, 6.5]),        # beyond z_far
                      S.pose(S.rot([1, 0, 0], 2.0), [-0.3, -0.2, 0.7])])
    w = check_render(gpu, v2, n2, f2, poses, K, H, W, "edge cases")
    hit = np.isfinite(w["vertices"][..., 2])
    assert hit[0].sum() > 300 and hit[0][:, -1].any() and hit[1].sum() > 0 and not hit[2].any()
    assert w["vertices"][1][hit[1]][:, 2].min() >= 0.25
    empty = icp.Mesh(np.zeros((0, 3), F), np.zeros((0, 3), np.int32), np.zeros((0, 3), F), device=gpu)
    out = icp.render(empty, poses[:1], K, H, W, want=("vertices", "normals", "canonical"))
    assert all(np.isnan(N(a)).all() for a in out.values())
    assert icp.render(gpu_mesh(gpu, v, n, f), np.zeros((0, 3, 4)), K, H, W)["vertices"].shape == (0, H, W, 4)


def make_case(H, W, dz, obj=5, sub=3):
    K = scaled_K(W)
    v, n, f = S.icosphere(0.06, sub, scale=(1.0, 0.7, 1.3))
    T_true = S.pose(S.rot([0.3, 1, 0.2], 0.7), [-0.02, 0.015, 0.7])
    T_est = S.pose(T_true[:, :3], T_true[:, 3] + np.array([0.0, 0.0, dz]))
    depth, label = S.depth_scene_from_mesh(lambda P: oracle.render_mesh(v, n, f, P, K, H, W, want=("vertices",))["vertices"], T_true, K, H, W, obj_id=obj)
    return K, (v, n, f), T_true, T_est, depth, label


@pytest.mark.parametrize("H,W", [(480, 640), (100, 131)])
def test_center_and_score_bit_identical(gpu, H, W):
    import torch
    from posecnn_amd import icp
    obj = 5
    K, (v, n, f), T_true, T_est, depth, label = make_case(H, W, 0.004, obj)
    label[: H // 4] = np.where(label[: H // 4] > 0, 9, 0)                        # part of the object carries another label
    holes = (np.arange(H * W).reshape(H, W) % 11 == 0) & (label > 0)
    depth = np.where(holes, 0, depth).astype(np.uint16)                         # missing depth readings
    maps = oracle.render_mesh(v, n, f, T_est[None], K, H, W, model_index=obj - 1)
    live = oracle.icp_backproject(depth, label, obj, K, 10000.0)
    want_s, want_m = oracle.icp_center(label, live, maps["canonical"][0], maps["vertices"][0], maps["normals"][0], obj, 0.0035)
    got_s, got_m = icp.center(T(gpu, label), T(gpu, live), T(gpu, maps["canonical"][0]), T(gpu, maps["vertices"][0]), T(gpu, maps["normals"][0]), obj, 0.0035)
    assert np.array_equal(N(got_s).view(np.uint64), want_s.view(np.uint64)), (N(got_s), want_s)
    same(N(got_m), want_m, "mask")
    assert want_s[4] > want_s[3] > 200, want_s                                  # 4 mm off, 3.5 mm gate: only the slanted part of the surface agrees
    hyps = np.repeat(T_est[None], 6, 0)
    hyps[:, 2, 3] += np.array([-0.004, 0.0, -0.02, 0.012, 0.05, 1.0])
    hyps[3] = S.compose(S.pose(S.rot([0, 1, 0], 0.02), [0, 0, 0]), hyps[3])
    if H * W <= 20000:
        want_h = oracle.icp_score(live, maps["canonical"][0], want_m, hyps, 0.01)
    else:                                                                        # the exhaustive checker is O(pairs^2): thin the pairs out
        thin = want_m & (np.arange(H * W).reshape(H, W) % 3 == 0)
        want_m = thin.astype(np.uint8)
        want_h = oracle.icp_score(live, maps["canonical"][0], want_m, hyps, 0.01)
    got_h = icp.score(T(gpu, live), T(gpu, maps["canonical"][0]), T(gpu, want_m), hyps, K, 0.01)
    same(N(got_h), want_h, "hits")
    assert want_h[0] == want_h.max() and want_h[0] > 0 and want_h[5] == 0, want_h
    # a radius so large that the window is the whole image, and a hypothesis at the camera (full-scan branch)
    hyps2 = np.stack([hyps[0], S.pose(np.eye(3), [0, 0, 0.001])])
    if H * W <= 20000:
        same(N(icp.score(T(gpu, live), T(gpu, maps["canonical"][0]), T(gpu, want_m), hyps2, K, 0.3)),
             oracle.icp_score(live, maps["canonical"][0], want_m, hyps2, 0.3), "hits, 30 cm radius")


@pytest.mark.parametrize("H,W,pc", [(480, 640, 4), (100, 131, 3)])
def test_polish_bit_identical(gpu, H, W, pc):
    """pcnn_icp_polish_fwd (one launch, simplex in LDS) against oracle_icp_polish: the best vertex, its energy and the number of
    evaluations for budgets that stop in every branch (initial simplex only, mid-iteration, the reference's 50), an absent
    object, 3- and 4-channel predicted maps."""
    from posecnn_amd import icp
    obj = 5
    K, (v, n, f), T_true, _, depth, label = make_case(H, W, 0.0, obj)
    T_est = S.pose(S.rot([0, 1, 0], 0.05) @ T_true[:, :3], T_true[:, 3] + np.array([0.004, -0.003, 0.02]))
    live = oracle.icp_backproject(depth, label, obj, K, 10000.0)
    pv = oracle.render_mesh(v, n, f, T_est[None], K, H, W, want=("vertices",))["vertices"][0][..., :pc].copy()
    for budget in (8, 9, 10, 13, 20, 50, 120):
        wx, we, wn = oracle.icp_polish(label, live, pv, obj, maxeval=budget)
        U, ge, gn, gx = icp.polish(T(gpu, label), T(gpu, live), T(gpu, pv), obj, max_evaluations=budget)
        assert gn == wn == budget and ge == we, (budget, gn, wn, ge, we)
        assert np.array_equal(gx.view(np.uint64), wx.view(np.uint64)), (budget, gx, wx)
    assert we < 0.5 * oracle.icp_polish(label, live, pv, obj, maxeval=8)[1]
    U, ge, gn, gx = icp.polish(T(gpu, label), T(gpu, live), T(gpu, pv), 7)
    assert gn == 0 and ge == 0 and np.array_equal(gx, [1, 0, 0, 0, 0, 0, 0]) and np.array_equal(U, np.hstack([np.eye(3), np.zeros((3, 1))]))


def test_synthesizer_icp_python_equals_the_flow_on_the_checker(gpu):
    """lib/fcn/test.py:1925-1933 on a two-object frame + one ROI that is skipped (too few pixels) + a background ROI:
    outputs / outputs_icp of Synthesizer.icp_python against tests/icp_scene.solve_icp_reference, per hypothesis hit counts
    bit for bit; and the refined pose is closer to the truth than the network's."""
    from posecnn_amd import icp
    from posecnn_amd.pose_error import quat2mat
    H, W = 240, 320
    K = scaled_K(W)
    meshes_np = [S.icosphere(0.05, 3, scale=(1.0, 0.7, 1.3)), S.box_mesh((0.05, 0.04, 0.03)), S.icosphere(0.03, 2)]
    truths = [S.pose(S.rot([0.3, 1, 0.2], 0.7), [-0.08, 0.015, 0.7]), S.pose(S.rot([1, 0.4, 0.1], 1.0), [0.09, -0.02, 0.8]),
              S.pose(np.eye(3), [0.0, 0.11, 1.5])]
    depth = np.zeros((H, W), np.uint16)
    label = np.zeros((H, W), np.int32)
    for c, (m, Tt) in enumerate(zip(meshes_np, truths)):
        d, l = S.depth_scene_from_mesh(lambda P: oracle.render_mesh(m[0], m[1], m[2], P, K, H, W, want=("vertices",))["vertices"], Tt, K, H, W, obj_id=c + 1)
        depth = np.where(l > 0, d, depth).astype(np.uint16)
        label = np.where(l > 0, l, label).astype(np.int32)
    assert (label == 3).sum() < 400 <= min((label == 1).sum(), (label == 2).sum())
    # the network's poses: 2 cm / 3 cm too far, a degree or two off
    ests = [S.pose(S.rot([0, 1, 0], 0.02) @ truths[0][:, :3], truths[0][:, 3] * (1 + 0.02 / 0.7)),
            S.pose(S.rot([1, 0, 0], -0.03) @ truths[1][:, :3], truths[1][:, 3] * (1 + 0.03 / 0.8)),
            truths[2]]
    rois = np.zeros((4, 7), F)
    poses = np.zeros((4, 7), F)
    for r, c in enumerate((1, 2, 3, 0)):
        rois[r, 1] = c
        if c:
            poses[r, :4] = icp.mat2quat(ests[c - 1][:, :3])
            poses[r, 4:] = ests[c - 1][:, 3]
    parameters = np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2], 0.25, 6.0, 10000.0], F)
    syn = icp.Synthesizer(meshes=[icp.Mesh(m[0], m[2], m[1], device=gpu) for m in meshes_np], device=gpu)
    syn.setup(W, H)
    poses_new = np.zeros((4, 7), F)
    poses_icp = np.zeros((4, 7), F)
    syn.icp_python(label, depth, parameters, H, W, 4, 7, rois, poses, poses_new, poses_icp, 0.01)
    assert not poses_new[2:].any() and not poses_icp[2:].any() and len(syn.last) == 2      # skipped rows stay as they were
    Kf = np.array([[parameters[0], 0, parameters[2]], [0, parameters[1], parameters[3]], [0, 0, 1]], np.float64)
    for r in (0, 1):
        q_t = poses[r].astype(np.float64)
        T_in = np.zeros((3, 4)); T_in[:, :3] = quat2mat(q_t[:4]); T_in[:, 3] = q_t[4:]
        ref = S.solve_icp_reference(label, depth, Kf, 10000.0, r + 1, T_in, meshes_np[r], q_t=q_t)
        info = syn.last[r]
        same(info["hits"], ref["hits"], "hits of ROI %d" % r)
        assert info["choose"] == ref["choose"] and info["pairs"] == ref["pairs"] and info["agree"] == ref["agree"]
        for got, want in ((poses_new[r], ref["T_new"]), (poses_icp[r], ref["T_icp"])):
            assert np.abs(quat2mat(got[:4].astype(np.float64)) - want[:, :3]).max() < 1e-6 and np.abs(got[4:] - want[:, 3]).max() < 1e-6
        T_icp = np.zeros((3, 4)); T_icp[:, :3] = quat2mat(poses_icp[r, :4].astype(np.float64)); T_icp[:, 3] = poses_icp[r, 4:]
        e0, e1 = S.pose_error(ests[r], truths[r]), S.pose_error(T_icp, truths[r])
        assert e0[1] > 0.019 and e1[1] < 2e-3 and e1[0] < 0.6, (r, e0, e1)


def test_obj_loader_and_model_file(gpu, tmp_path):
    from posecnn_amd import icp
    v, n, f = S.box_mesh((0.05, 0.04, 0.03))
    lines = ["# box"] + ["v %.6f %.6f %.6f" % tuple(p) for p in v] + ["vn %.1f %.1f %.1f" % tuple(p) for p in n]
    for q in range(6):                                                           # quads with v//vn corners
        b = 4 * q + 1
        lines.append("f %d//%d %d//%d %d//%d %d//%d" % (b, b, b + 1, b + 1, b + 2, b + 2, b + 3, b + 3))
    p1 = tmp_path / "box.obj"
    p1.write_text("\n".join(lines) + "\n")
    vs, ns, fs = S.icosphere(0.05, 1)
    p2 = tmp_path / "sphere.obj"                                                # no normals: generated (area-weighted, smooth)
    p2.write_text("\n".join(["v %.7f %.7f %.7f" % tuple(p) for p in vs] + ["f %d %d %d" % tuple(t + 1) for t in fs]) + "\n")
    lst = tmp_path / "models.txt"
    lst.write_text("%s\n%s\n" % (p1, p2))
    syn = icp.Synthesizer(str(lst), None, device=gpu)
    syn.setup(64, 48)
    box, sph = syn.meshes
    assert np.array_equal(box.faces_np, f) and np.allclose(box.vertices_np, v, atol=1e-6) and np.array_equal(box.normals_np, n)
    assert np.array_equal(sph.faces_np, fs) and (np.sum(sph.normals_np * ns, axis=1) > 0.99).all()
    with pytest.raises(ValueError):
        icp.Mesh(v, np.array([[0, 1, 99]], np.int32), n, device=gpu)


def test_new_entries_reject_bad_arguments(gpu):
    import torch
    from posecnn_amd._lib import lib
    L = lib()
    nb = ctypes.c_size_t()
    assert L.pcnn_render_mesh_workspace_bytes(2, 4, 5, ctypes.byref(nb)) == 0 and nb.value == 2 * 4 * 5 * 8
    assert L.pcnn_icp_score_workspace_bytes(3, 4, 16, ctypes.byref(nb)) == 0 and nb.value == 3 * 2 * 4
    buf = torch.zeros(4096, dtype=torch.float32, device=gpu)
    p = ctypes.c_void_p(buf.data_ptr())
    z = ctypes.c_void_p(0)
    # z_near <= 0; workspace too small; NULL poses; normal map without normals
    assert L.pcnn_render_mesh_fwd(p, p, p, 3, 1, p, 1, 4, 4, 100.0, 100.0, 2.0, 2.0, 0.0, 6.0, 0.0, p, p, p, p, 4096, z) < 0
    assert L.pcnn_render_mesh_fwd(p, p, p, 3, 1, p, 1, 4, 4, 100.0, 100.0, 2.0, 2.0, 0.25, 6.0, 0.0, p, p, p, p, 8, z) < 0
    assert L.pcnn_render_mesh_fwd(p, p, p, 3, 1, z, 1, 4, 4, 100.0, 100.0, 2.0, 2.0, 0.25, 6.0, 0.0, p, p, p, p, 4096, z) < 0
    assert L.pcnn_render_mesh_fwd(p, z, p, 3, 1, p, 1, 4, 4, 100.0, 100.0, 2.0, 2.0, 0.25, 6.0, 0.0, p, p, p, p, 4096, z) < 0
    assert L.pcnn_icp_center_fwd(p, p, p, p, p, 5, 4, 4, 1, 0.01, p, p, p, 4096, z) < 0          # 5 channels
    assert L.pcnn_icp_center_fwd(p, p, p, p, p, 4, 4, 4, 1, 0.01, p, p, p, 4, z) < 0             # workspace
    assert L.pcnn_icp_score_fwd(p, p, p, 4, 4, p, 1, 100.0, 100.0, 2.0, 2.0, 0.0, p, p, 4096, z) < 0    # radius
    assert L.pcnn_icp_score_fwd(p, p, p, 4, 4, z, 1, 100.0, 100.0, 2.0, 2.0, 0.01, p, p, 4096, z) < 0   # NULL hypotheses
    assert L.pcnn_icp_polish_fwd(p, p, p, 4, 4, 4, 1, 0.25, 6.0, 7, p, p, z) < 0                        # fewer evaluations than the initial simplex
    assert L.pcnn_icp_polish_fwd(p, p, z, 4, 4, 4, 1, 0.25, 6.0, 50, p, p, z) < 0
    torch.cuda.synchronize()
