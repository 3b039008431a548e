"""CPU tests of the second ICP slice's checker: oracle_render_mesh / oracle_icp_center / oracle_icp_score (the steps of
Synthesizer::solveICP around the ICP iterations, lib/synthesize/synthesize.cpp:2104-2136, :2157-2225, :2302-2343).
PARITY UNPINNED (OpenGL, PCL and nlopt are absent and the reference has no test vectors): the renderer is held to
hand-derived known answers and to an analytic ray-caster, the other two to independent numpy restatements."""
import numpy as np

import icp_scene as S
import oracle
from posecnn_amd import config

F = np.float32
IDENT = np.hstack([np.eye(3), np.zeros((3, 1))])


def small_K(W, H):
    K = config.DEMO_INTRINSICS.copy()
    K[:2] *= W / 640.0
    return K


def test_single_triangle_known_answers():
    """fx = fy = 100, principal point (8, 6): the triangle (-0.04,-0.03,1) (0.04,-0.03,1) (-0.04,0.05,1) projects to
    (4,3) (12,3) (4,11) — pixel centres ON the edges are covered (inclusive), the hypotenuse x + y = 15 included."""
    K = np.array([[100.0, 0, 8.0], [0, 100.0, 6.0], [0, 0, 1]])
    v = np.array([[-0.04, -0.03, 1.0], [0.04, -0.03, 1.0], [-0.04, 0.05, 1.0]], F)
    n = np.array([[0, 0, -1.0]] * 3, F)
    f = np.array([[0, 1, 2]], np.int32)
    out = oracle.render_mesh(v, n, f, IDENT[None], K, 16, 20, model_index=4)
    hit = np.isfinite(out["vertices"][0, :, :, 2])
    ys, xs = np.mgrid[0:16, 0:20]
    want = (xs >= 4) & (ys >= 3) & (xs + ys <= 15)
    assert np.array_equal(hit, want)
    vm = out["vertices"][0]
    assert np.allclose(vm[hit][:, 2], 1.0, atol=1e-6) and np.array_equal(vm[hit][:, 3], np.ones(hit.sum(), F))
    assert np.allclose(vm[5, 7, :3], [(7 - 8) / 100.0, (5 - 6) / 100.0, 1.0], atol=1e-6)          # the pixel's ray at depth 1
    assert np.allclose(out["normals"][0][hit][:, :3], [0, 0, -1.0]) and not out["normals"][0][hit][:, 3].any()
    can = out["canonical"][0]
    assert np.allclose(can[5, 7], [4 - 0.01, -0.01, 1.0], atol=1e-6)                                # identity pose: object frame = camera frame, x + model index
    assert np.isnan(out["vertices"][0][~hit]).all() and np.isnan(can[~hit]).all()
    # the other winding covers the same pixels; a triangle in front of z_near or past z_far covers none
    out2 = oracle.render_mesh(v, n, f[:, ::-1], IDENT[None], K, 16, 20, want=("vertices",))
    assert np.array_equal(np.isfinite(out2["vertices"][0, :, :, 2]), want)
    assert not np.isfinite(oracle.render_mesh(v * F(0.2), n, f, IDENT[None], K, 16, 20, want=("vertices",))["vertices"]).any()
    assert not np.isfinite(oracle.render_mesh(v * F(7.0), n, f, IDENT[None], K, 16, 20, want=("vertices",))["vertices"]).any()


def test_perspective_correct_interpolation_on_a_slanted_plane():
    """A large quad in the plane z = 1 + 0.5 x: the rendered point of every covered pixel must lie on the pixel's ray AND on
    the plane (screen-space linear interpolation would not)."""
    K = np.array([[120.0, 0, 32.0], [0, 120.0, 24.0], [0, 0, 1]])
    xs = np.array([-0.2, 0.2])
    v = np.array([[x, y, 1 + 0.5 * x] for y in (-0.15, 0.15) for x in xs], F)
    f = np.array([[0, 1, 3], [0, 3, 2]], np.int32)
    out = oracle.render_mesh(v, None, f, IDENT[None], K, 48, 64, want=("vertices", "canonical"))
    vm = out["vertices"][0]
    hit = np.isfinite(vm[..., 2])
    assert hit.sum() > 1000
    yy, xx = np.nonzero(hit)
    p = vm[hit].astype(np.float64)
    assert np.abs(p[:, 2] - (1 + 0.5 * p[:, 0])).max() < 2e-6
    assert np.abs(p[:, 0] / p[:, 2] - (xx - 32.0) / 120.0).max() < 2e-6 and np.abs(p[:, 1] / p[:, 2] - (yy - 24.0) / 120.0).max() < 2e-6
    assert np.abs(out["canonical"][0][hit] - vm[hit][:, :3]).max() < 1e-6


def test_box_mesh_matches_the_ray_caster():
    H, W = 120, 160
    K = small_K(W, H)
    half = (0.08, 0.06, 0.05)
    v, n, f = S.box_mesh(half)
    T = S.pose(S.rot([1, 1, 0.3], 0.9), [0.02, -0.01, 0.8])
    out = oracle.render_mesh(v, n, f, T[None], K, H, W)
    rv, rn, rhit = S.render_box(T, half, K, H, W)
    hit = np.isfinite(out["vertices"][0, :, :, 2])
    assert rhit.sum() > 2000
    assert (hit != rhit).sum() <= 0.01 * rhit.sum()                   # only silhouette pixels (centre exactly on / next to an edge)
    both = hit & rhit
    assert np.abs(out["vertices"][0][both][:, :3] - rv[both]).max() < 5e-6
    # normals: away from the box's edges the rasterised normal is the face normal
    dn = np.abs(out["normals"][0][both][:, :3] - rn[both]).max(axis=1)
    assert (dn < 1e-5).mean() > 0.97
    # canonical map: T^-1 of the camera-frame point
    can = out["canonical"][0][both].astype(np.float64)
    back = (out["vertices"][0][both][:, :3].astype(np.float64) - T[:, 3]) @ T[:, :3]
    assert np.abs(can - back).max() < 5e-6


def test_sphere_is_watertight_and_close_to_the_analytic_surface():
    H, W = 96, 128
    K = small_K(W, H)
    r = 0.07
    v, n, f = S.icosphere(r, 3)
    c = np.array([0.01, -0.02, 0.75])
    T = S.pose(S.rot([0.2, 1, 0.4], 1.1), c)
    out = oracle.render_mesh(v, n, f, T[None], K, H, W, want=("vertices", "normals"))
    vm = out["vertices"][0]
    hit = np.isfinite(vm[..., 2])
    ys, xs = np.mgrid[0:H, 0:W]
    d = np.stack([(xs - K[0, 2]) / K[0, 0], (ys - K[1, 2]) / K[1, 1], np.ones_like(xs, dtype=np.float64)], -1)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    b = d @ c
    disc = b * b - (c @ c - r * r)
    # every ray that passes the centre within 0.97 r hits the inscribed polyhedron: no holes
    inner = b * b - (c @ c - (0.97 * r) ** 2) > 0
    assert inner.sum() > 800 and hit[inner].all()
    assert not hit[disc < 0].any()                                     # and nothing outside the sphere's silhouette
    p = vm[hit][:, :3].astype(np.float64)
    dist = np.linalg.norm(p - c, axis=1)
    assert dist.max() <= r * (1 + 1e-5) and dist.min() > 0.985 * r      # chord error of 1280 faces
    nm = out["normals"][0][hit][:, :3].astype(np.float64)
    radial = (p - c) / dist[:, None]
    assert (np.sum(nm * radial, axis=1) > 0.995).all()                 # interpolated unit-sphere normals, rotated


def test_nearer_surface_wins_and_poses_batch():
    H, W = 72, 96
    K = small_K(W, H)
    vb, nb, fb = S.box_mesh((0.1, 0.1, 0.01))
    vs, ns, fs = S.icosphere(0.04, 2)
    # one mesh: a plate with a sphere 8 cm in front of it (object frame: camera looks down +z, so "in front" = smaller z)
    v = np.concatenate([vb, vs + np.array([0, 0, -0.08], F)])
    n = np.concatenate([nb, ns])
    f = np.concatenate([fb, fs + len(vb)])
    T0 = S.pose(np.eye(3), [0, 0, 0.8])
    T1 = S.pose(S.rot([0, 1, 0], 0.3), [0.02, 0.0, 0.9])
    both = oracle.render_mesh(v, n, f, np.stack([T0, T1]), K, H, W)
    for k, T in enumerate((T0, T1)):
        one = oracle.render_mesh(v, n, f, T[None], K, H, W)
        for key in one:
            assert np.array_equal(one[key][0], both[key][k], equal_nan=True)
    z = both["vertices"][0, :, :, 2]
    cy, cx = int(round(K[1, 2])), int(round(K[0, 2]))
    assert abs(z[cy, cx] - (0.8 - 0.08 - 0.04)) < 2e-3                 # the sphere's front pole, not the plate behind it
    assert abs(z[cy, cx + 14] - (0.8 - 0.01)) < 1e-5                   # beside the sphere: the plate's near face


def np_center(label, live, canon, pv, pn, obj, max_error):
    """independent restatement of synthesize.cpp:2157-2207 (float64 sums)"""
    vx = canon[..., 0] - np.round(canon[..., 0])
    valid = (label == obj) & (live[..., 2] > 0) & np.isfinite(vx) & np.isfinite(canon[..., 1]) & np.isfinite(canon[..., 2])
    with np.errstate(invalid="ignore"):
        err = np.sum(pn[..., :3].astype(np.float64) * (live.astype(np.float64) - pv[..., :3]), axis=-1)
        agree = valid & (np.abs(err) < max_error)
    m = np.stack([vx, canon[..., 1], canon[..., 2]], -1)
    diff = (live.astype(np.float64) - m)[agree]
    return np.array([diff[:, 0].sum(), diff[:, 1].sum(), diff[:, 2].sum(), agree.sum(), valid.sum()]), valid


def make_case(H=96, W=128, dz=0.012, seed=0):
    K = small_K(W, H)
    v, n, f = S.icosphere(0.06, 2, scale=(1.0, 0.7, 1.3))
    T_true = S.pose(S.rot([0.3, 1, 0.2], 0.7), [-0.02, 0.015, 0.7])
    T_est = S.pose(T_true[:, :3], T_true[:, 3] + np.array([0.0, 0.0, dz]))
    depth, label = S.depth_scene_from_mesh(lambda P: oracle.render_mesh(v, n, f, P, K, H, W, want=("vertices",))["vertices"], T_true, K, H, W, obj_id=5)
    live = oracle.icp_backproject(depth, label, 5, K, 10000.0)
    maps = oracle.render_mesh(v, n, f, T_est[None], K, H, W, model_index=4)
    return K, (v, n, f), T_true, T_est, depth, label, live, maps


def test_center_estimate_against_numpy_and_by_value():
    K, mesh, T_true, T_est, depth, label, live, maps = make_case(dz=0.004)
    sums, mask = oracle.icp_center(label, live, maps["canonical"][0], maps["vertices"][0], maps["normals"][0], 5, 0.01)
    want, valid = np_center(label, live, maps["canonical"][0], maps["vertices"][0], maps["normals"][0], 5, 0.01)
    assert np.array_equal(mask.astype(bool), valid)
    assert sums[3] == want[3] and sums[4] == want[4] and sums[3] > 500
    assert np.allclose(sums[:3], want[:3], rtol=2e-6)
    # (d - m) averaged = the translation that carries the model points onto the depth points when R = I; with a rotation
    # it is the reference's estimate all the same: z of it lands near the true depth of the object's visible surface
    Tz = sums[2] / sums[3]
    assert abs(Tz - T_true[2, 3]) < 0.08
    # the other object's pixels, pixels without depth, and pixels the render does not cover are not pairs
    lab2 = label.copy(); lab2[:48] = 9
    s2, m2 = oracle.icp_center(lab2, live, maps["canonical"][0], maps["vertices"][0], maps["normals"][0], 5, 0.01)
    assert not m2[:48].any() and s2[4] == mask[48:].sum()
    # max_error gates the votes, not the pairs
    s3, m3 = oracle.icp_center(label, live, maps["canonical"][0], maps["vertices"][0], maps["normals"][0], 5, 0.0005)
    assert np.array_equal(m3, mask) and s3[3] < sums[3]


def np_score(live, canon, mask, hyps, radius):
    P = live.reshape(-1, 3).astype(np.float32)
    idx = np.nonzero(mask.reshape(-1))[0]
    m = canon.reshape(-1, 3)[idx].copy()
    m[:, 0] = m[:, 0] - np.round(m[:, 0])
    pts = P[idx]
    out = []
    for T in hyps:
        T = T.astype(np.float32)
        q = ((T[:, 0] * m[:, :1] + T[:, 1] * m[:, 1:2]) + T[:, 2] * m[:, 2:3]) + T[:, 3]
        marked = set()
        for k in range(len(q)):
            e = pts - q[k]
            d2 = (e[:, 0] * e[:, 0] + e[:, 1] * e[:, 1]) + e[:, 2] * e[:, 2]
            j = int(np.argmin(d2))
            if d2[j] < F(radius) * F(radius):
                marked.add(j)
        out.append(len(marked))
    return np.array(out, np.int32)


def test_score_against_numpy_and_prefers_the_true_depth():
    K, mesh, T_true, T_est, depth, label, live, maps = make_case(H=72, W=96, dz=0.0)
    _, mask = oracle.icp_center(label, live, maps["canonical"][0], maps["vertices"][0], maps["normals"][0], 5, 0.01)
    hyps = np.repeat(T_true[None], 5, 0)
    hyps[:, 2, 3] += np.array([0.0, -0.02, 0.01, 0.03, 0.2])
    hits = oracle.icp_score(live, maps["canonical"][0], mask, hyps, 0.01)
    assert np.array_equal(hits, np_score(live, maps["canonical"][0], mask, hyps, 0.01))
    assert hits[0] == hits.max() and hits[0] > 0.9 * mask.sum()       # at the true pose nearly every model point finds its own depth point
    assert hits[1] < hits[0] and hits[3] < hits[2] < hits[0] and hits[4] == 0
    # an empty mask scores nothing
    assert not oracle.icp_score(live, maps["canonical"][0], np.zeros_like(mask), hyps, 0.01).any()


def test_solve_icp_flow_recovers_a_depth_offset():
    """The whole flow on the checker (tests/icp_scene.solve_icp_reference): the network's pose is 2.5 cm too far; the
    translation estimate pulls it back to the depth data, the Nelder-Mead polish tightens it, ICP refines the 8 hypotheses, the
    score picks one near the truth."""
    K, (v, n, f), T_true, T_est, depth, label, live, maps = make_case(dz=0.025)
    res = S.solve_icp_reference(label, depth, K, 10000.0, 5, T_est, (v, n, f))
    e_in = S.pose_error(T_est, T_true)[1]
    e_new = S.pose_error(res["T_new"], T_true)[1]
    e_icp = S.pose_error(res["T_icp"], T_true)
    assert e_in > 0.024 and e_new < 5e-3 and e_icp[1] < 1e-3 and e_icp[0] < 1.0, (e_in, e_new, e_icp)   # (a smooth ellipsoid constrains the rotation weakly: the polish may move it by half a degree)
    assert res["hits"][res["choose"]] == res["hits"].max() and res["pairs"] > 700


def test_mesh_loader_and_generated_normals(tmp_path):
    """posecnn_amd.icp.Mesh on the host side (device="cpu": no kernels involved): OBJ records, polygons fanned, negative
    (relative) indices, normals generated when the file has none."""
    from posecnn_amd import icp
    v, n, f = S.box_mesh((0.05, 0.04, 0.03))
    p = tmp_path / "quad.obj"
    p.write_text("v 0 0 1\nv 1 0 1\nv 1 1 1\nv 0 1 1\nf -4 -3 -2 -1\n")
    m = icp.Mesh.load_obj(str(p), device="cpu")
    assert np.array_equal(m.faces_np, [[0, 1, 2], [0, 2, 3]]) and np.allclose(m.normals_np, [[0, 0, 1.0]] * 4)
    vs, ns, fs = S.icosphere(0.05, 2)
    gen = icp.Mesh.smooth_normals(vs, fs)
    assert (np.sum(gen * ns, axis=1) > 0.995).all() and np.allclose(np.linalg.norm(gen, axis=1), 1.0, atol=1e-6)
    import pytest
    with pytest.raises(ValueError):
        icp.Mesh(v, np.array([[0, 1, 24]], np.int32), n, device="cpu")


def test_polish_lowers_the_energy_and_respects_its_box_and_budget():
    """oracle_icp_polish (poseWithOpt / optEnergy, synthesize.cpp:2476-2570): the first 8 evaluations are the initial simplex
    (x0 first), the budget is kept exactly, the best vertex stays inside the box, the energy never goes up with the budget,
    and 50 evaluations take a 2 cm / 3 degree error to a few millimetres."""
    H, W = 96, 128
    K = small_K(W, H)
    v, n, f = S.icosphere(0.06, 2, scale=(1.0, 0.7, 1.3))
    T_true = S.pose(S.rot([0.3, 1, 0.2], 0.7), [-0.02, 0.015, 0.7])
    T_est = S.pose(S.rot([0, 1, 0], 0.05) @ T_true[:, :3], T_true[:, 3] + np.array([0.004, -0.003, 0.02]))
    depth, label = S.depth_scene_from_mesh(lambda P: oracle.render_mesh(v, n, f, P, K, H, W, want=("vertices",))["vertices"], T_true, K, H, W, obj_id=5)
    live = oracle.icp_backproject(depth, label, 5, K, 10000.0)
    pv = oracle.render_mesh(v, n, f, T_est[None], K, H, W, want=("vertices",))["vertices"][0]
    # the energy at the identity update, by hand: mean |pred - live| over the object's pixels with both depths in range
    ok = (label == 5) & np.isfinite(pv[..., 2]) & (live[..., 2] > 0.25)
    e0 = np.linalg.norm(pv[ok][:, :3].astype(np.float64) - live[ok], axis=1).mean()
    x8, e8, n8 = oracle.icp_polish(label, live, pv, 5, maxeval=8)
    assert n8 == 8 and abs(e8 - e0) < 1e-6 * e0 + 1e-7 and np.array_equal(x8, [1, 0, 0, 0, 0, 0, 0])    # every simplex step makes it worse here or not: x0 holds
    last = e8
    for budget in (9, 20, 50):
        x, e, ne = oracle.icp_polish(label, live, pv, 5, maxeval=budget)
        assert ne == budget and e <= last
        assert (np.abs(x - [1, 0, 0, 0, 0, 0, 0]) <= np.array([0.1] * 4 + [0.01, 0.01, 0.1]) + 1e-15).all()
        last = e
    assert last < 0.35 * e0
    T_pol = S.compose(S.pose(S.quat2mat(x[:4]), x[4:]), T_est)
    assert S.pose_error(T_pol, T_true)[1] < 0.4 * S.pose_error(T_est, T_true)[1]
    # no pixel of the object: the identity, zero evaluations
    x, e, ne = oracle.icp_polish(label, live, pv, 6, maxeval=50)
    assert ne == 0 and e == 0 and np.array_equal(x, [1, 0, 0, 0, 0, 0, 0])


def test_random_triangulations_are_watertight():
    """The shared-edge rule: 400-point Delaunay triangulations of a tilted square sheet (mixed windings, slivers) leave no
    hole — every pixel whose ray meets the sheet at least half a pixel inside its border is covered, and carries the plane's
    depth; nothing outside the sheet's image is covered."""
    H, W = 120, 160
    K = small_K(W, H)
    for seed in range(4):
        v, f = S.random_sheet(seed)
        out = oracle.render_mesh(v, None, f, IDENT[None], K, H, W, want=("vertices",))["vertices"][0]
        hit = np.isfinite(out[..., 2])
        ys, xs = np.mgrid[0:H, 0:W]
        rx, ry = (xs - K[0, 2]) / K[0, 0], (ys - K[1, 2]) / K[1, 1]
        # ray (rx, ry, 1) t meets z = z0 + a x + b y at t = z0 / (1 - a rx - b ry)
        t = 0.8 / (1 - 0.4 * rx + 0.25 * ry)
        px_, py_ = rx * t, ry * t
        margin = 0.6 * t / K[0, 0]
        inside = (np.abs(px_) < 0.12 - margin) & (np.abs(py_) < 0.12 - margin)
        outside = (np.abs(px_) > 0.12 + margin) | (np.abs(py_) > 0.12 + margin)
        assert inside.sum() > 4000 and hit[inside].all(), (seed, (~hit[inside]).sum())
        assert not hit[outside].any()
        assert np.abs(out[..., 2][inside] - t[inside]).max() < 5e-6


def test_refinement_checker_under_address_and_ub_sanitizers():
    """oracle/asan_driver.c: the checker in one translation unit with -fsanitize=address,undefined ("ASAN build of
    the CPU oracle"), driven over odd-sized inputs — pose refinement (objects cut by the image border, zero faces, empty
    masks, every polish budget from 8 to 40), then Hough voting (train / test, both threshold branches), ROI pooling with
    off-image / inverted / bad-batch ROIs and its backward, hard labels with ignore and out-of-range ids, softmax, both
    deconvolutions, the pose loss with a symmetric class, the vertex loss. Any out-of-bounds access, use-after-free, signed
    overflow or misaligned access aborts the driver."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["make", "-C", os.path.join(root, "oracle"), "asan"], capture_output=True, text=True, timeout=300)
    if r.returncode != 0 and "sanitize" in (r.stderr + r.stdout):
        import pytest
        pytest.skip("this compiler has no sanitizer runtime")
    assert r.returncode == 0, r.stdout + r.stderr
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([os.path.join(root, "oracle", "_asan", "asan_driver")], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "asan_driver ok" in r.stdout and "ERROR" not in r.stderr and "runtime error" not in r.stderr, r.stdout + r.stderr
