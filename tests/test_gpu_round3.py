"""GPU tests added in round 3 (VERDICT r2 "Next" #1, #4):
  * the trunk numerics study — 32 full-size RGB-D frames through a float64 trunk, a direct f32 trunk with a
    known summation order, the library's convolutions and the Winograd-MFMA default; round 4: on the
    CALIBRATED synthetic network (synth.init_calibrated) it asserts north_star literally — 0 label flips,
    |dq| < 1e-4, |dt| < 1e-4 absolute — for all three f32 trunks;
  * `datasets.run_evaluation` — the test_net_single_frame loop (lib/fcn/test.py:1154-1467) executed end to end
    on a 5-frame YCB-Video-layout tree built from the demo depth/label fixtures;
  * backproject at the shape `bench.py --config linemod` runs it on (960x1280x64, C = 14, k = 3);
  * the grouped RGB-D trunk with fused_pool=False;
  * the softmax head against an independent float64 softmax.
"""
import os

import numpy as np
import pytest

import oracle
from posecnn_amd import config, synth
from test_gpu_ops import N, T, backproject_case, same

pytestmark = pytest.mark.gpu
F = np.float32
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ---- trunk numerics: the end-to-end parity statement ---------------------------------------------------
def test_trunk_numerics_study_32_full_size_frames(gpu, capsys):
    """north_star: label maps bit-exact, quaternions / translations within 1e-4 — given IDENTICAL dense-layer
    outputs the custom kernels are bit-exact (test_gpu_hough / test_gpu_ops). The dense layers are f32 in a
    different summation order than TF/cuDNN's (unknowable), so the end-to-end statement is
    statistical: every f32 trunk (direct or Winograd) against the float64 trunk on 32 frames of 480x640."""
    import parity_study
    res = parity_study.run_study(gpu, n_frames=32, batch=4)
    with capsys.disabled():
        print("\ntrunk numerics study (vs float64 trunk):")
        for p, r in res["paths"].items():
            print("  %-9s" % p, {k: (round(v, 8) if isinstance(v, float) else v) for k, v in r.items()})
    for p, r in res["paths"].items():
        assert r["detections_compared"] >= 32 * 4 and r["detections_missing_or_extra"] == 0 and r["supported_detections"] >= 32 * 3, (p, r)
        # north_star, literally (VERDICT r3 "Next" #1): on the calibrated network (synth.init_calibrated: O(1) activations,
        # |fc8| <~ 3, depths 0.5-2 m) every f32 trunk gives the float64 trunk's label maps BIT FOR BIT (9.8 M decisions),
        # the same Hough cells and boxes, and quaternions / translations within 1e-4 ABSOLUTE for every detection
        assert r["label_flips"] == 0, (p, r)
        assert r["winning_cell_moved"] == 0 and r["max_box_diff_px"] < 1e-3, (p, r)
        assert r["max_quat_diff"] < 1e-4, (p, r)
        assert r["trans_diff_max"] < 1e-4, (p, r)
        # the scales the statement is made at: fc8 un-saturated, every detection at a depth a camera sees
        assert r["fc8_absmax_median"] < 4.0 and r["vertex_field_absmax"] < 40.0, (p, r)
        assert 0.3 < r["depth_min_m"] and r["depth_max_m"] < 3.0, (p, r)
        # mechanism (round 3): no voter crosses the hard 0.9 inlier test; a translation is mean(exp(z)) over identical voters
        assert r["voters_changed_total"] == 0, (p, r)
        assert r["max_excess_over_voter_bound"] <= 0.0 and r["trans_rel_diff_max"] < 1e-4, (p, r)
        assert r["fc8_rel_err_max"] < 1e-5 and r["max_quat_diff"] <= 1.01 * r["fc8_abs_err_max"] + 1e-7, (p, r)
    w, d, l = res["paths"]["winograd"], res["paths"]["taps_f32"], res["paths"]["library"]
    # Winograd F(4x4,3x3) against a direct f32 convolution of known summation order: same class — every error
    # figure within ~4x of the direct convolution's (measured 5.3e-6 vs 1.3e-6 on conv5_3, 2x the library's)
    assert d["conv5_3_rel_err"] < 5e-6 and l["conv5_3_rel_err"] < 1e-5 and w["conv5_3_rel_err"] < 2e-5, (w, d, l)
    assert w["vertex_field_rel_err"] < 5e-6 and w["max_prob_diff"] < 1e-4, w


def test_grouped_trunk_does_not_depend_on_fused_pool(gpu):
    """with fused_pool=False (or a cleared dual_pool) the grouped RGB-D trunk registered the
    POOLED conv4_3 under 'conv4_3'. conv4_3 must be the un-pooled tensor whatever the pooling switches say."""
    import torch
    from posecnn_amd import fcn
    from posecnn_amd.networks import vgg16_convs
    B, H, W = 2, 96, 128
    rng = np.random.default_rng(4)
    data = T(gpu, rng.standard_normal((B, H, W, 3)).astype(F) * 50)
    data_p = T(gpu, rng.standard_normal((B, H, W, 3)).astype(F) * 50)
    K = config.DEMO_INTRINSICS.copy(); K[:2] *= W / 640.0
    pts = T(gpu, synth.make_model_points(22, 64))
    outs = []
    for fused_pool in (True, False):
        net = vgg16_convs("RGBD", 22, 64, (1.0,), 1.0, -1.0, vertex_reg_2d=True, pose_reg=True, trainable=False,
                          is_train=False, seed=3, init="he", with_losses=False, device=gpu, fused_pool=fused_pool)
        synth.init_calibrated(net)
        planted_np, _ = synth.make_planted_batch(9, B, H=H, W=W, K=K, n_obj=2)
        with torch.no_grad():
            det = fcn.im_segment_batch(net, data, K, config.LOV_EXTENTS, pts, config.LOV_SYMMETRY, data_p=data_p,
                                       planted={k: T(gpu, v) for k, v in planted_np.items()})
        assert tuple(net.get_output("conv4_3").shape) == (B, H // 8, W // 8, 512)
        assert tuple(net.get_output("pool4").shape) == (B, H // 16, W // 16, 512)
        outs.append((N(net.get_output("conv4_3_p")), N(net.get_output("pool4")), N(det.rows), N(det.label_2d)))
    for a, b, name in zip(outs[0], outs[1], ("conv4_3_p", "pool4", "rows", "label_2d")):
        same(a, b, name)


def test_softmax_head_against_independent_float64_softmax(gpu):
    """kernel, C oracle and numpy reference share the canonical exp sequence, so bit-equality
    among them cannot see a defect in that sequence. Independent anchor: float64 softmax of the same scores."""
    from posecnn_amd import ops
    rng = np.random.default_rng(77)
    x = (rng.standard_normal((2, 60, 80, 22)) * 6).astype(F)
    x[0, 0, 0, :] = 0.0                      # all-equal scores: 1/22 each, argmax 0
    x[0, 0, 1, :] = -1e30; x[0, 0, 1, 7] = 3   # one live class
    prob, label = ops.softmax_argmax(T(gpu, x), want_prob=True)
    xd = x.astype(np.float64)
    e = np.exp(xd - xd.max(-1, keepdims=True))
    want = e / e.sum(-1, keepdims=True)
    got = N(prob).astype(np.float64)
    assert np.abs(got - want).max() < 4e-7, np.abs(got - want).max()          # a few f32 ulp of a value <= 1
    assert np.abs(got.sum(-1) - 1.0).max() < 2e-6
    assert np.array_equal(N(label), xd.argmax(-1).astype(np.int32))
    # relative accuracy of small probabilities (the hard_label threshold compares them): the f32 subtraction x - max
    # rounds to half an ulp of |x - max| (an absolute error of the exponent = a relative error of the result), then
    # exp (<= 1.2 ulp), the sum and the division (<= 0.5 ulp each)
    m = want > 1e-30
    tol = (np.abs(xd - xd.max(-1, keepdims=True)) * 2.0 ** -24 + 8 * 2.0 ** -23)[m]
    assert ((np.abs(got[m] - want[m]) / want[m]) <= tol).all()


def test_misaligned_bias_views(gpu):
    """the split-K / Cin-split reduction kernels read the bias as float4. A bias that is a view
    at a 4-byte offset must work through ops.* (copied) and be refused by the C-ABI itself (EINVAL, no fault)."""
    import ctypes
    import torch
    from posecnn_amd import _lib, ops
    g = torch.Generator(device="cpu").manual_seed(5)
    M, K_, N_ = 5, 25088, 128                      # 5 live rows -> the split-K path
    x = torch.randn((M, K_), generator=g).to(gpu)
    w = (torch.randn((K_, N_), generator=g) / K_ ** 0.5).to(gpu)
    bfull = torch.randn((N_ + 1,), generator=g).to(gpu)
    bias = bfull[1:]
    assert bias.data_ptr() % 16 == 4
    cnt = torch.tensor([M], dtype=torch.int32, device=gpu)
    y = ops.fc_rows(x, w.t().contiguous(), bias, True, num_rows=cnt)
    ref = torch.relu(x.double() @ w.double() + bias.double())
    assert float((y.double() - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max()))
    wt = w.t().contiguous()
    out = torch.empty((M, N_), device=gpu)
    rc = _lib.lib().pcnn_fc_rows_fwd(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(wt.data_ptr()), ctypes.c_void_p(bias.data_ptr()),
                                     M, K_, N_, 1, ctypes.c_void_p(cnt.data_ptr()), ctypes.c_void_p(0), ctypes.c_void_p(out.data_ptr()),
                                     ctypes.c_void_p(0), 0, ctypes.c_void_p(torch.cuda.current_stream(gpu).cuda_stream))
    assert rc == _lib.PCNN_EINVAL


# ---- the evaluation loop, executed -----------------------------------------------------------
def _write_demo_tree(root):
    """A YCB-Video-layout tree (lov.py:57-135) from tests/golden/demo_frames.npz (the reference's own demo depth
    images + a deterministic segmentation of them) and lov_models.npz (real model points / extents):
    colour = the depth image as grey levels, `-meta.mat` poses = identity rotation at the blob's centroid and
    median depth. Returns per frame [(cls, cx, cy, z)]."""
    import scipy.io
    from PIL import Image
    fr = np.load(os.path.join(GOLD, "demo_frames.npz"))
    md = np.load(os.path.join(GOLD, "lov_models.npz"))
    os.makedirs(os.path.join(root, "data", "0048"))
    np.savetxt(os.path.join(root, "extents.txt"), md["extents"][1:], fmt="%.6f")
    for i, c in enumerate(config.LOV_CLASSES[1:]):
        os.makedirs(os.path.join(root, "models", c))
        np.savetxt(os.path.join(root, "models", c, "points.xyz"), md["points"][i + 1], fmt="%.6f")
    K = config.DEMO_INTRINSICS
    names, objects = [], []
    for f in range(fr["depth"].shape[0]):
        depth, label = fr["depth"][f], fr["label"][f]
        name = "0048/%06d" % (f + 1)
        names.append(name)
        grey = np.clip(depth.astype(np.float64) / 20000.0 * 255, 0, 255).astype(np.uint8)
        Image.fromarray(np.stack([grey, grey // 2, 255 - grey], axis=-1)).save(os.path.join(root, "data", name + "-color.png"))
        Image.fromarray(depth).save(os.path.join(root, "data", name + "-depth.png"))
        Image.fromarray(label).save(os.path.join(root, "data", name + "-label.png"))
        objs = []
        for c in [int(c) for c in np.unique(label) if c]:
            ys, xs = np.nonzero(label == c)
            z = float(np.median(depth[ys, xs])) / config.DEMO_FACTOR_DEPTH
            objs.append((c, float(np.round(xs.mean())), float(np.round(ys.mean())), z))
        poses = np.zeros((3, 4, len(objs)))
        for j, (c, cx, cy, z) in enumerate(objs):
            poses[:, :3, j] = np.eye(3)
            poses[:, 3, j] = ((cx - K[0, 2]) / K[0, 0] * z, (cy - K[1, 2]) / K[1, 1] * z, z)
        scipy.io.savemat(os.path.join(root, "data", name + "-meta.mat"),
                         {"intrinsic_matrix": K, "factor_depth": np.array([[config.DEMO_FACTOR_DEPTH]]), "poses": poses,
                          "cls_indexes": np.array([[o[0]] for o in objs])})
        objects.append(objs)
    with open(os.path.join(root, "keyframe.txt"), "w") as fh:
        fh.write("\n".join(names) + "\n")
    return objects


def _planted_evaluation_setup(gpu, tmp_path):
    """demo tree on disk + a random-weight network whose planted 1/8-resolution scene is derived from each frame's own ground
    truth + a dataset wrapper that plants frame i's scene before handing the frame out"""
    import torch  # noqa: F401
    from posecnn_amd import datasets, fcn
    from posecnn_amd.networks import vgg16_convs
    objects = _write_demo_tree(str(tmp_path))
    ds = datasets.YCBVideo(str(tmp_path), "keyframe")
    assert len(ds) == 5
    K = config.DEMO_INTRINSICS

    class planted_net(vgg16_convs):
        scene = None

        def run(self, feed, planted=None):
            return vgg16_convs.run(self, feed, planted=self.scene)

    net = planted_net("RGBD", 22, 64, (1.0,), 1.0, -1.0, vertex_reg_2d=True, pose_reg=True, trainable=False,
                      is_train=False, seed=3, init="he", with_losses=False, device=gpu)
    synth.init_calibrated(net)

    def scene_of(i):
        label = ds.frame(i)["label"]
        low = label[4::8, 4::8]
        h, w = low.shape
        yy, xx = np.mgrid[0:h, 0:w]
        yy, xx = yy * 8 + 3.5, xx * 8 + 3.5
        s = np.zeros((1, h, w, 64), F)
        for c in range(22):
            s[0, ..., c] = 30.0 * (low == c)
        v = np.zeros((1, h, w, 128), F)
        for c, cx, cy, z in objects[i]:
            ang = np.arctan2(cy - yy, cx - xx)
            v[0, ..., 3 * c], v[0, ..., 3 * c + 1], v[0, ..., 3 * c + 2] = 30.0 * np.cos(ang), 30.0 * np.sin(ang), np.log(z)
        return {"add_score": T(gpu, s), "add_score_vertex": T(gpu, v)}

    class one_frame_at_a_time(object):       # run_evaluation pulls frames in order: plant frame i's scene before it runs
        classes, extents, num_classes, points = ds.classes, ds.extents, ds.num_classes, ds.points

        def __len__(self):
            return len(ds)

        def frame(self, i):
            net.scene = scene_of(i)
            return ds.frame(i)

    return ds, net, one_frame_at_a_time(), objects, scene_of


def test_run_evaluation_executes_the_test_net_loop(gpu, tmp_path):
    """lib/fcn/test.py:1154-1467 (`test_net_single_frame`): frame -> pad_im(16) -> im_segment_single_frame ->
    un-padded labels, NMS'd rois + poses -> imdb.evaluate_result -> evaluate_segmentations, on real depth frames.
    The network has random weights; a planted 1/8-resolution scene derived from each frame's OWN ground truth makes
    the heads emit that frame's label map and a vertex field that points at the ground-truth centres,
    so the evaluator must find high IoU and millimetre translation errors — which only happens if the reader,
    the blob construction, the single-frame driver, NMS, the pose combine and the evaluator all line up."""
    import torch
    from posecnn_amd import datasets, fcn
    ds, net, wrapped, objects, scene_of = _planted_evaluation_setup(gpu, tmp_path)
    K = config.DEMO_INTRINSICS
    _, points_all = ds.points
    mat_dir = tmp_path / "mats"
    mat_dir.mkdir()
    with torch.no_grad():
        ev = datasets.run_evaluation(net, wrapped, points_all, config.LOV_SYMMETRY, device=gpu, mat_dir=str(mat_dir))
    s = ev.summary()
    assert s["frames"] == 5 and ev.hist.sum() == 5 * 480 * 640
    # every ground-truth object was counted, class by class
    want_all = np.zeros(22)
    for objs in objects:
        for c, _, _, _ in objs:
            want_all[c] += 1
    assert s["poses_all"] == want_all[1:].tolist() and want_all.sum() >= 20
    # segmentation: the planted 1/8-resolution scene reproduces the ground truth up to 8-pixel block borders
    big = [c for c in range(1, 22) if ev.hist[c].sum() > 5 * 3000]
    assert len(big) >= 4 and all(s["per_class_iu"][config.LOV_CLASSES[c]] > 0.6 for c in big), s["per_class_iu"]
    assert s["overall_accuracy"] > 0.9
    # poses: re-run frame 0 by hand and check what evaluate_result saw
    net.scene = scene_of(0)
    fr = ds.frame(0)
    with torch.no_grad():
        labels, probs, vertex_pred, rois, poses = fcn.im_segment_single_frame(
            net, fcn.pad_im(fr["color"], 16), fcn.pad_im(fr["depth"], 16), fr["meta"], ds.extents, points_all,
            config.LOV_SYMMETRY, 22, device=gpu)
    assert labels.shape == (480, 640) and probs.shape == (480, 640, 22) and vertex_pred.shape == (480, 640, 66)
    import scipy.io
    m = scipy.io.loadmat(str(mat_dir / "000000.mat"))
    assert np.array_equal(m["labels"], labels) and np.array_equal(m["rois"], rois) and np.array_equal(m["poses"], poses)
    detected = {int(r[1]): p for r, p in zip(rois, poses)}
    first = datasets.Evaluator(ds.classes, ds.extents, ds.points[0]).evaluate_result(labels, rois, poses, fr["label"], fr["meta"])
    seen = 0
    for c, cx, cy, z in objects[0]:
        if (labels == c).sum() <= 500 or c not in detected:
            continue
        seen += 1
        t = detected[c][4:7]
        want_t = np.array([(cx - K[0, 2]) / K[0, 0] * z, (cy - K[1, 2]) / K[1, 1] * z, z])
        # the Hough centre is the planted centroid (every voter points at it); the depth is mean(exp(log z + the random
        # network's own contribution to that channel)) — right order of magnitude, not the planted value
        u, v = t[0] / t[2] * K[0, 0] + K[0, 2], t[1] / t[2] * K[1, 1] + K[1, 2]
        # the cone test cos > 0.9 is +-25 degrees and the planted directions live on 8-pixel cells: a broad maximum, within three cells
        assert abs(u - cx) < 24 and abs(v - cy) < 24, (c, (u, v), (cx, cy))
        assert t[2] > 0      # (the depth is exp(log z + the random network's own output on that channel): any positive number)
        e = ev.pose_error(c, detected[c], fr["meta"]["poses"][:, :, [o[0] for o in objects[0]].index(c)])
        assert abs(e["translation_error"] - np.linalg.norm(t.astype(np.float64) - want_t)) < 1e-6     # the evaluator's `te` of THIS detection
        mine = [p_ for p_ in first["poses"] if p_["class"] == config.LOV_CLASSES[c]]
        assert len(mine) == 1 and abs(mine[0]["translation_error"] - e["translation_error"]) < 1e-9 and mine[0]["correct"] == bool(e["error"] < ev.threshold[c])
    assert seen >= 3
    rep = ev.write_reports(str(tmp_path / "report"))
    assert rep["frames"] == 5 and os.path.exists(str(tmp_path / "report" / "confusion_matrix.txt"))


def test_run_evaluation_with_pose_refinement(gpu, tmp_path):
    """cfg.TEST.POSE_REFINE (lib/fcn/test.py:1896-1933, lov.py:381-389, :463-511): the loop hands the un-padded labels, the
    depth image and the network's rois / poses to `synthesizer.icp_python` and evaluates poses_refined / poses_icp next to the
    network's pose. Real depth frames; the models are ellipsoids with the classes' extents (no YCB mesh exists offline), so
    what is asserted is the plumbing: the loop's refined poses equal a direct call on the same inputs, they land in the .mat
    record, every evaluated detection carries the three error triples, and the translation is the depth-based one exactly
    when depth points agreed with the render (the refinement's accuracy is tested on synthetic scenes with known poses,
    tests/test_gpu_icp_render.py)."""
    import scipy.io
    import torch
    import icp_scene as S
    from posecnn_amd import datasets, fcn, icp
    ds, net, wrapped, objects, scene_of = _planted_evaluation_setup(gpu, tmp_path)
    _, points_all = ds.points
    meshes = []
    for c in range(1, 22):
        v, n, f = S.icosphere(1.0, 2)
        v = (v * (ds.extents[c] / 2.0)).astype(F)
        meshes.append(icp.Mesh(v, f, None, device=gpu))                # normals generated from the faces
    syn = icp.Synthesizer(meshes=meshes, device=gpu)
    syn.setup(640, 480)
    mat_dir = tmp_path / "mats_refine"
    mat_dir.mkdir()
    with torch.no_grad():
        ev = datasets.run_evaluation(net, wrapped, points_all, config.LOV_SYMMETRY, device=gpu, mat_dir=str(mat_dir), max_frames=2,
                                     synthesizer=syn)
    assert ev.summary()["frames"] == 2
    m = scipy.io.loadmat(str(mat_dir / "000001.mat"))
    assert m["poses_refined"].shape == m["poses"].shape == m["poses_icp"].shape and m["poses"].shape[1] == 7
    # the same frame by hand
    net.scene = scene_of(1)
    fr = ds.frame(1)
    with torch.no_grad():
        labels, _, _, rois, poses = fcn.im_segment_single_frame(net, fcn.pad_im(fr["color"], 16), fcn.pad_im(fr["depth"], 16), fr["meta"], ds.extents,
                                                                points_all, config.LOV_SYMMETRY, 22, device=gpu)
    assert np.array_equal(m["rois"], rois) and np.array_equal(m["poses"], poses)
    Km = np.asarray(fr["meta"]["intrinsic_matrix"], dtype=np.float64)
    par = np.array([Km[0, 0], Km[1, 1], Km[0, 2], Km[1, 2], 0.25, 6.0, float(np.asarray(fr["meta"]["factor_depth"]).reshape(-1)[0])], F)
    pn, pi = np.zeros_like(poses), np.zeros_like(poses)
    lab = np.ascontiguousarray(labels, np.int32)
    syn.icp_python(lab, np.ascontiguousarray(fr["depth"], np.uint16), par, 480, 640, rois.shape[0], rois.shape[1], rois, poses, pn, pi, 0.01)
    assert np.array_equal(m["poses_refined"], pn) and np.array_equal(m["poses_icp"], pi)
    assert len(syn.last) >= 2
    for info in syn.last:
        r = info["roi"]
        assert np.isfinite(pn[r]).all() and np.isfinite(pi[r]).all() and abs(np.linalg.norm(pi[r, :4]) - 1) < 1e-5
        if info["agree"] > 0:
            # depth points agreed with the render: the depth-based translation sits on the object's depth pixels
            zs = fr["depth"][labels == int(rois[r, 1])].astype(np.float64) / par[6]
            zs = zs[zs > 0]
            assert abs(pn[r, 6] - zs.mean()) < np.linalg.norm(ds.extents[int(rois[r, 1])]), (r, pn[r, 6], zs.mean())
        else:
            # (the random network's depth can put the model nowhere near the data — even in front of z_near, where nothing
            #  is rendered: the reference then keeps the network's translation, synthesize.cpp:2238-2239)
            assert np.allclose(pn[r, 4:], poses[r, 4:], atol=1e-6)
    first = datasets.Evaluator(ds.classes, ds.extents, ds.points[0]).evaluate_result(labels, rois, poses, fr["label"], fr["meta"], poses_new=pn, poses_icp=pi)
    assert first["poses"] and all(("error_new" in e and "error_icp" in e and "translation_error_icp" in e) for e in first["poses"])


@pytest.mark.parametrize("H,W", [(480, 640), (50, 70)])
def test_raw_frame_first_conv_is_bit_identical_to_the_blob_path(gpu, H, W):
    """pcnn_conv3x3_c3_winograd43_raw_fwd: uint8 BGR / uint16 depth frames in, the blobs of lib/fcn/test.py:56-74 formed
    inside the kernel — V must equal the blob path on host-built blobs (fcn._get_image_blob: float32 -= float64 means, the
    depth clip / scale / tile) bit for bit: both towers in one launch, colour only, depth only, ragged tiles."""
    import torch
    from posecnn_amd import fcn, ops
    rng = np.random.default_rng(H)
    B = 2
    im8 = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    d16 = rng.integers(0, 3000, (B, H, W)).astype(np.uint16)
    d16[0, :3, :5] = 65535
    d16[1, -2:, :] = 0
    blobs = [fcn._get_image_blob(im8[i], d16[i]) for i in range(B)]
    data = np.concatenate([b[0] for b in blobs]).astype(F)
    data_p = np.concatenate([b[1] for b in blobs]).astype(F)
    w = (rng.standard_normal((2, 3, 3, 3, 64)) * 0.1).astype(F)
    b = rng.standard_normal((2, 64)).astype(F)
    tw, tb = T(gpu, w), T(gpu, b)
    t8, t16 = T(gpu, im8), torch.from_numpy(d16).to(gpu)
    want = ops.conv3x3_c3_winograd43(T(gpu, np.concatenate([data, data_p])), tw, tb, True, groups=2)
    got = ops.conv3x3_c3_winograd43_raw(t8, t16, tw, tb, True)
    same(N(got), N(want), "both towers")
    same(N(ops.conv3x3_c3_winograd43_raw(t8, None, tw[:1].contiguous(), tb[:1].contiguous(), True)),
         N(ops.conv3x3_c3_winograd43(T(gpu, data), tw[:1].contiguous(), tb[:1].contiguous(), True)), "colour only")
    same(N(ops.conv3x3_c3_winograd43_raw(None, t16, tw[1:].contiguous(), tb[1:].contiguous(), False)),
         N(ops.conv3x3_c3_winograd43(T(gpu, data_p), tw[1:].contiguous(), tb[1:].contiguous(), False)), "depth only, no ReLU")
    # the framework-side conversion the non-fused configurations use gives the same blobs
    from posecnn_amd.networks import raw_frame_blob
    same(N(raw_frame_blob(t8)), data, "colour blob")
    same(N(raw_frame_blob(t16)), data_p, "depth blob")
    with pytest.raises(ValueError):
        ops.conv3x3_c3_winograd43_raw(None, None, tw, tb)


@pytest.mark.parametrize("fmt,strict", [("RGBD", False), ("COLOR", False), ("RGBD", True)])
def test_network_on_raw_frames_equals_network_on_blobs(gpu, fmt, strict):
    """`im_segment_batch` fed uint8 / uint16 frames (bench.py --raw-inputs) against the same frames as f32 blobs: grouped
    RGB-D towers, a single colour tower, and the strict_numerics configuration (no fused first conv: the frames are turned
    into blobs on the device first) — detections, label maps and fc7 bit for bit."""
    import torch
    from posecnn_amd import fcn
    from posecnn_amd.networks import vgg16_convs
    B, H, W = 2, 240, 320
    net = vgg16_convs(fmt, 22, 64, (1.0,), 1.0, -1.0, vertex_reg_2d=True, pose_reg=True, trainable=False, is_train=False, seed=3,
                      init="he", with_losses=False, device=gpu, strict_numerics=strict)
    synth.init_calibrated(net)
    K = config.DEMO_INTRINSICS.copy(); K[:2] *= W / 640.0
    rng = np.random.default_rng(3)
    im8 = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    d16 = rng.integers(0, 3000, (B, H, W)).astype(np.uint16)
    blobs = [fcn._get_image_blob(im8[i], d16[i]) for i in range(B)]
    data = T(gpu, np.concatenate([b[0] for b in blobs]).astype(F))
    data_p = T(gpu, np.concatenate([b[1] for b in blobs]).astype(F)) if fmt == "RGBD" else None
    planted_np, _ = synth.make_planted_batch(7, B, H=H, W=W, K=K, n_obj=3)
    planted = {k: T(gpu, v) for k, v in planted_np.items()}
    pts = T(gpu, synth.make_model_points(22, 256))
    outs = []
    with torch.no_grad():
        for (d_, dp_) in ((data, data_p), (T(gpu, im8), torch.from_numpy(d16).to(gpu) if fmt == "RGBD" else None)):
            det = fcn.im_segment_batch(net, d_, K, config.LOV_EXTENTS, pts, config.LOV_SYMMETRY, data_p=dp_, planted=planted)
            outs.append([N(det.rows), N(det.count), N(det.label_2d), N(net.get_output("fc7"))])
    assert int(outs[0][1].reshape(-1)[0]) >= 3
    for a_, b_, name in zip(outs[0], outs[1], ("rows", "count", "label_2d", "fc7")):
        if not strict:
            same(b_.reshape(-1), a_.reshape(-1), "%s (%s)" % (name, fmt))
        elif name == "label_2d":
            assert (a_ == b_).mean() > 0.9999
        else:
            # the library convolutions of this configuration are not run-to-run reproducible to the bit (the SAME blobs give
            # conv5_3 2e-6 of its range apart in a second pass); the blobs themselves are compared bit for bit in the test above
            assert np.abs(b_.astype(np.float64) - a_).max() <= 1e-5 * max(np.abs(a_).max(), 1.0), name


def test_listener_callback_builds_the_posecnn_message(gpu, tmp_path):
    """ros/listener.py:41-90 without ROS (posecnn_amd.listener.ImageListener): a (colour, depth) pair of the demo tree ->
    the PoseCNNMsg fields, equal to what im_segment_single_frame returns for that frame; both publishers are called; a 32FC1
    depth message gives the same result as its 16UC1 twin; an unsupported encoding is dropped."""
    import torch
    from posecnn_amd import fcn, listener
    ds, net, wrapped, objects, scene_of = _planted_evaluation_setup(gpu, tmp_path)
    fr = ds.frame(0)
    net.scene = scene_of(0)
    sent, sent_label, logged = [], [], []
    li = listener.ImageListener(net, ds, fr["meta"], config.LOV_SYMMETRY, publish=sent.append, publish_label=sent_label.append, device=gpu,
                                log=logged.append)
    with torch.no_grad():
        msg = li.callback(fr["color"], fr["depth"], "16UC1")
        _, points_all = ds.points
        labels, _, _, rois, poses = fcn.im_segment_single_frame(net, fcn.pad_im(fr["color"], 16), fcn.pad_im(fr["depth"], 16), fr["meta"], ds.extents,
                                                                points_all, config.LOV_SYMMETRY, 22, device=gpu)
        metres = fr["depth"].astype(np.float32) / 1000.0
        msg32 = li.callback(fr["color"], metres, "32FC1")
        assert li.callback(fr["color"], fr["depth"], "8UC1") is None and len(logged) == 1 and "Unsupported depth type" in logged[0]
    K = np.asarray(fr["meta"]["intrinsic_matrix"])
    assert (msg["height"], msg["width"], msg["roi_num"], msg["roi_channel"]) == (480, 640, rois.shape[0], 7) and rois.shape[0] >= 3
    assert msg["fx"] == float(K[0, 0]) and msg["py"] == float(K[1, 2]) and msg["znear"] == 0.25 and msg["zfar"] == 6.0
    assert np.array_equal(msg["label"], labels.astype(np.uint8)) and np.array_equal(msg["depth"], fr["depth"])
    assert msg["rois"] == rois.astype(F).flatten().tolist() and msg["poses"] == poses.astype(F).flatten().tolist()
    assert len(sent) == 2 and sent[0] is msg and len(sent_label) == 2 and sent_label[0].shape == (480, 640, 3)
    assert np.array_equal(sent_label[0], listener.labels_to_image(labels))
    # uint16(depth / 1000 * 1000) can differ from depth by one count: the RGB-D network sees a slightly different depth blob
    assert msg32["roi_num"] == msg["roi_num"] and np.abs(np.asarray(msg32["depth"], np.int32) - fr["depth"]).max() <= 1


def test_two_graphs_replaying_concurrently_equal_the_eager_steps(gpu):
    """bench.py --graph --streams 2: one hipGraph per device slot, the two replayed concurrently on two
    streams. Every GraphedStep warms up and captures on a stream of its own, so the library scratch it bakes in (keyed by
    stream: Hough / ADL workspaces, fc6 split-K partials, Cin-split partials at batch 1) is private to it. 8 rounds of
    overlapping replays on changing frame contents must reproduce the eager single-stream results bit for bit."""
    import torch
    from posecnn_amd import fcn, pipeline
    from posecnn_amd.networks import vgg16_convs
    from test_gpu_round2 import _rgbd_inputs
    B, H, W = 1, 240, 320
    net = vgg16_convs("RGBD", 22, 64, (1.0,), 1.0, -1.0, vertex_reg_2d=True, pose_reg=True, trainable=False,
                      is_train=True, seed=3, init="he", with_losses=False, device=gpu)
    synth.init_calibrated(net)
    K = config.DEMO_INTRINSICS.copy(); K[:2] *= W / 640.0
    rng = np.random.default_rng(18)
    pts = T(gpu, synth.make_model_points(22, 256))
    frames = []
    for i in range(4):
        data, data_p = _rgbd_inputs(rng, B, H, W)
        planted_np, scenes = synth.make_planted_batch(80 + i, B, H=H, W=W, K=K, n_obj=3)
        frames.append((T(gpu, data), T(gpu, data_p), {k: T(gpu, v) for k, v in planted_np.items()}, T(gpu, synth.make_gt_poses(scenes, K, seed=i))))
    slots = []
    for k in range(2):
        f = frames[k]
        slots.append({"data": f[0].clone(), "data_p": f[1].clone(), "plant": {n_: v.clone() for n_, v in f[2].items()}, "gt": f[3].clone()})
    feed = fcn._feed(net, slots[0]["data"], slots[0]["data_p"], K, config.LOV_EXTENTS, pts, config.LOV_SYMMETRY, 22, gpu)

    def make_step(sl):
        def step():
            det = fcn.im_segment_batch(net, sl["data"], K, config.LOV_EXTENTS, pts, config.LOV_SYMMETRY, data_p=sl["data_p"],
                                       planted=sl["plant"], feed_cache=feed, with_losses=True, gt_poses=sl["gt"])
            return det.rows, det.count, det.label_2d, net.get_output("loss_pose"), net.get_output("fc7")
        return step

    def fill(sl, f):
        sl["data"].copy_(f[0]); sl["data_p"].copy_(f[1]); sl["gt"].copy_(f[3])
        for n_ in sl["plant"]:
            sl["plant"][n_].copy_(f[2][n_])

    with torch.no_grad():
        eager = []
        for f in frames:
            fill(slots[0], f)
            eager.append([t.clone() for t in make_step(slots[0])()])
        torch.cuda.synchronize()
        graphs = [pipeline.GraphedStep(make_step(sl), warmup=1, device=gpu) for sl in slots]
        streams = [torch.cuda.Stream(device=gpu), torch.cuda.Stream(device=gpu)]
        bad = 0
        for rnd in range(8):
            pick = [(rnd + k) % 4 for k in range(2)] if rnd % 2 else [(rnd + 2 * k + 1) % 4 for k in range(2)]
            outs = []
            for k in range(2):
                with torch.cuda.stream(streams[k]):
                    fill(slots[k], frames[pick[k]])
                    outs.append([t.clone() for t in graphs[k].replay()])
            torch.cuda.synchronize()
            for k in range(2):
                for got, want, name in zip(outs[k], eager[pick[k]], ("rows", "count", "label_2d", "loss_pose", "fc7")):
                    g_, w_ = N(got).reshape(-1), N(want).reshape(-1)
                    if name == "fc7":
                        live = int(outs[k][1]) * 9
                        g_, w_ = N(got)[:live].reshape(-1), N(want)[:live].reshape(-1)
                    if not np.array_equal(g_.view(np.uint32) if g_.dtype.kind == "f" else g_, w_.view(np.uint32) if w_.dtype.kind == "f" else w_):
                        bad += 1
        assert bad == 0, "%d tensors differed between concurrent graph replays and the eager steps" % bad
        assert int(eager[0][1]) > 0


# ---- configs[4]: backproject at the shape the LINEMOD preset runs ------------------------------------------
def test_backproject_at_the_linemod_bench_shape(gpu):
    """backprojecting_op_gpu.cu.cc:17-126 at 960x1280 inputs, 64 data channels, C = 14 class channels, k = 3
    (7x7 window), G = 64 — `bench.py --config linemod` runs this shape (G = 128 there) through
    backproject_fused_kernel; bit-exact against the oracle."""
    from posecnn_amd import ops
    rng = np.random.default_rng(31)
    B, H, W, Cd, Cl, G, k = 1, 960, 1280, 64, 14, 64, 3
    data, label, depth, meta, label3d = backproject_case(rng, B, H, W, Cd, Cl, G)
    m4 = meta.reshape(B, 1, 1, 48)
    td, tl, tf = ops.backproject(T(gpu, data), T(gpu, label), T(gpu, depth), T(gpu, m4), T(gpu, label3d), G, k, 0.05)
    wd, wl, wf = oracle.backproject(data, label, depth, meta, label3d, G, k, 0.05)
    assert 0.02 < wf.mean() < 0.98          # both branches (surface hit / miss) are exercised
    same(N(td), wd, "top_data")
    same(N(tf), wf, "top_flag")
    same(N(tl), wl, "top_label")


# ---- BASELINE configs[1]: the few-row pose branch and the small head kernels -------------------------------------
@pytest.mark.parametrize("M,K,N,count,act", [(5, 25088, 4096, 5, "relu"), (21, 4096, 4096, 17, "relu"), (21, 4096, 88, 4, "tanh"),
                                             (21, 4096, 88, 21, "tanh"), (1, 256, 40, None, "none"), (16, 1024, 130, 0, "relu"),
                                             (32, 2064, 200, 32, "none"), (7, 25088, 256, 3, "relu")])
def test_fc_skinny_matches_float64_and_is_deterministic(gpu, M, K, N, count, act):
    """`Network.fc` at <= 32 rows (csrc/fc_skinny.hip): rows below the device-side count equal act(x @ W + b) up to f32
    roundoff against float64, rows at or past it are exact zeros whatever the buffer held (NaN poison), two runs
    give the same bits (fixed-order split-K sum in the last workgroup), and the ticket counters are back at zero."""
    import torch
    from posecnn_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N)
    x = torch.randn((M, K), generator=g).to(gpu)
    w = (torch.randn((K, N), generator=g) / K ** 0.5).to(gpu)
    b = torch.randn((N,), generator=g).to(gpu)
    n = M if count is None else count
    if n < M:
        x[n:] = float("nan")
    cnt = None if count is None else torch.tensor([count], dtype=torch.int32, device=gpu)
    wt = w.t().contiguous()
    out = ops.fc_skinny(x, wt, b, act, num_rows=cnt)
    out2 = ops.fc_skinny(x, wt, b, act, num_rows=cnt)
    y, t = out if act == "tanh" else (out, None)
    y2, t2 = out2 if act == "tanh" else (out2, None)
    same(y.cpu().numpy(), y2.cpu().numpy(), "run-to-run")
    ref = x[:n].double() @ w.double() + b.double()
    if act == "relu":
        ref = torch.relu(ref)
    if n:
        scale = max(1.0, float(ref.abs().max()))
        assert float((y[:n].double() - ref).abs().max()) < 2e-5 * scale
    assert not y[n:].cpu().numpy().view(np.uint32).any()
    if act == "tanh":
        same(t.cpu().numpy(), t2.cpu().numpy(), "tanh run-to-run")
        if n:
            assert float((t[:n].double() - torch.tanh(y[:n].double())).abs().max()) < 3e-7
        assert not t[n:].cpu().numpy().view(np.uint32).any()
    assert all(int(v.abs().max()) == 0 for v in ops._tickets.values())


def test_head_lowres_equals_the_op_sequence_it_replaces(gpu):
    """csrc/heads_small.hip: add_score = score_conv4 + deconv(4,2)(score_conv5) [+ planted] must carry the bits of
    pcnn_deconv_bilinear_fwd + two framework adds; the 1x1 product is checked against float64."""
    import torch
    from posecnn_amd import ops
    rng = np.random.default_rng(12)
    for (B, h, w, U, Cout, plant) in ((2, 60, 80, 64, 22, True), (1, 60, 80, 128, 66, False), (3, 6, 10, 64, 5, True), (1, 2, 2, 8, 3, False)):
        a = T(gpu, rng.standard_normal((B, h, w, U)).astype(F))
        b5 = T(gpu, rng.standard_normal((B, h // 2, w // 2, U)).astype(F))
        pl = T(gpu, rng.standard_normal((B, h, w, U)).astype(F)) if plant else None
        wt = T(gpu, (rng.standard_normal((U, Cout)) / U ** 0.5).astype(F))
        add, z = ops.head_lowres(a, b5, wt, planted=pl)
        want = a + ops.deconv_bilinear(b5, 4, 2)
        if plant:
            want = want + pl
        same(N(add), N(want), "add_score %s" % ((B, h, w, U),))
        zr = want.double().reshape(-1, U) @ wt.double()
        assert float((z.double().reshape(-1, Cout) - zr).abs().max()) < 1e-5 * max(1.0, float(zr.abs().max()))


@pytest.mark.parametrize("stride", [1, 9])
def test_det_assemble_equals_the_host_pose_combine(gpu, stride):
    """lib/fcn/test.py:206-211: poses[i, :4] = poses_tanh[i, 4c : 4c+4]; rows past the count are zero; training mode
    keeps the first row of every group of 9."""
    import torch
    from posecnn_amd import ops
    rng = np.random.default_rng(3)
    C, R, n = 22, 27, 18
    rois = rng.standard_normal((R, 7)).astype(F)
    rois[:, 1] = rng.integers(0, C, R)
    rois[5, 1] = -1                       # clamped to class 0
    pt = rng.standard_normal((R, 4 * C)).astype(F)
    tp = rng.standard_normal((R, 7)).astype(F)
    rows, count = ops.det_assemble(T(gpu, rois), T(gpu, pt), T(gpu, tp), torch.tensor([n], dtype=torch.int32, device=gpu), row_stride=stride)
    want = np.zeros(((R + stride - 1) // stride, 14), F)
    for i in range(want.shape[0]):
        ri = i * stride
        if ri < n:
            c = max(int(rois[ri, 1]), 0)
            want[i] = np.concatenate([rois[ri], pt[ri, 4 * c:4 * c + 4], tp[ri, 4:]])
    same(N(rows), want, "rows")
    assert int(count) == n // stride


@pytest.mark.parametrize("C", [22, 14, 16])
def test_label_head_fixed_class_count_kernel_on_extreme_scores(gpu, C):
    """The unrolled packed-f32 label head (upscore_softmax_argmax_fixed_kernel<C>) must carry the bits of the generic kernel
    and the oracle everywhere the exponential does something special: arguments below -87.3 (sub-normal results, the two-step
    scaling), the clamp at -104, ties, and NaN / inf scores."""
    from posecnn_amd import ops
    rng = np.random.default_rng(C)
    B, h, w = 2, 9, 11
    z = (rng.standard_normal((B, h, w, C)) * 4).astype(F)
    z[0, 2, 3, :] = 0.0                                   # ties everywhere: argmax 0
    z[0, 4, :, 1] = 95.0                                  # one huge class: the others land at exp(-95 ...): sub-normal
    z[0, 5, :, 2] = 120.0; z[0, 5, :, 3] = -120.0         # past the clamp
    z[1, 1, 1, 4] = np.nan
    z[1, 6, 2, 5] = np.inf
    z[1, 7, 7, :] = -1e30
    bias = rng.standard_normal(C).astype(F)
    for relu in (True, False):
        score, prob, label = ops.upscore_softmax_argmax(T(gpu, z), T(gpu, bias), 16, 8, relu=relu, want_score=True)
        ws, wp, wl = oracle.upscore_softmax_argmax(z, bias, 16, 8, relu)
        same(N(score), ws, "score relu=%s" % relu)
        same(N(prob), wp, "prob relu=%s" % relu)
        same(N(label), wl, "label relu=%s" % relu)
    assert (wp[0, 4 * 8 + 4] < 1e-38).any() and (wp[0, 4 * 8 + 4] > 0).any()      # sub-normal probabilities were exercised


def test_new_entries_reject_bad_arguments(gpu):
    """Error conventions of the round-3 entries (validate, return negative codes, never fault): the Python layer
    raises ValueError for InvalidArgument / NULL, like the reference's OP_REQUIRES."""
    import torch
    from posecnn_amd import icp, ops
    x = torch.zeros((33, 64), device=gpu); w = torch.zeros((8, 64), device=gpu); b = torch.zeros(8, device=gpu)
    with pytest.raises(ValueError, match="1..32"):
        ops.fc_skinny(x, w, b)                                   # more than 32 rows belong to fc_rows
    with pytest.raises(ValueError, match="multiple of 16"):
        ops.fc_skinny(torch.zeros((4, 40), device=gpu), torch.zeros((8, 40), device=gpu), b)
    with pytest.raises(ValueError):
        ops.fc_skinny(x[:4], w, torch.zeros(9, device=gpu))      # bias / weight mismatch
    with pytest.raises(KeyError):
        ops.fc_skinny(x[:4], w, b, activation="gelu")
    a = torch.zeros((1, 6, 8, 16), device=gpu)
    with pytest.raises(ValueError):
        ops.head_lowres(a, torch.zeros((1, 3, 4, 8), device=gpu), torch.zeros((16, 5), device=gpu))     # channel mismatch
    with pytest.raises(ValueError, match="multiple of 4"):
        ops.head_lowres(torch.zeros((1, 6, 8, 6), device=gpu), torch.zeros((1, 3, 4, 6), device=gpu), torch.zeros((6, 5), device=gpu))
    with pytest.raises(ValueError):
        ops.det_assemble(torch.zeros((4, 6), device=gpu), torch.zeros((4, 88), device=gpu), torch.zeros((4, 7), device=gpu),
                         torch.zeros(1, dtype=torch.int32, device=gpu))
    live = torch.zeros((1, 16, 16, 3), device=gpu)
    with pytest.raises(ValueError):
        icp.icp(live, torch.zeros((1, 16, 16, 5), device=gpu), torch.zeros((1, 16, 16, 5), device=gpu), config.DEMO_INTRINSICS)
    with pytest.raises(ValueError):
        icp.icp(live, torch.zeros((1, 16, 16, 3), device=gpu), torch.zeros((1, 16, 16, 3), device=gpu), config.DEMO_INTRINSICS, max_error=-1.0)
    # zero iterations: the identity; zero objects: an empty result, nothing launched
    u = icp.icp(live, torch.zeros((1, 16, 16, 3), device=gpu), torch.zeros((1, 16, 16, 3), device=gpu), config.DEMO_INTRINSICS, iterations=0)
    assert np.array_equal(u.cpu().numpy()[0], np.hstack([np.eye(3), np.zeros((3, 1))]))
    u0 = icp.icp(live[:0], torch.zeros((0, 16, 16, 3), device=gpu), torch.zeros((0, 16, 16, 3), device=gpu), config.DEMO_INTRINSICS)
    assert tuple(u0.shape) == (0, 3, 4)
    with pytest.raises(ValueError):
        icp.backproject(torch.zeros((4, 4), dtype=torch.uint16, device=gpu), None, 0, config.DEMO_INTRINSICS, 0.0)   # factor_depth must be positive
    with pytest.raises(RuntimeError, match="GPU"):
        ops.fc_skinny(torch.zeros((4, 64)), w, b)                 # no CPU path
