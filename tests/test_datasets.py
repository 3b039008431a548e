"""CPU tests of posecnn_amd.datasets: the YCB-Video layout reader (lov.py:57-170) on a tree written
here with PIL / scipy, the evaluation of lov.py:397-680 on hand-computable cases, and — when the
reference tree is present (this container, not the GPU box) — the real fixtures: data/LOV/extents.txt
and data/LINEMOD/extents.txt against the constants in posecnn_amd/config.py, the 21 model point clouds,
the five demo frames."""
import os

import numpy as np
import pytest

from posecnn_amd import config, datasets, pose_error

REF = "/root/reference"
F = np.float32


def write_tree(root, rng, n_frames=3, H=24, W=32):
    import scipy.io
    from PIL import Image
    classes = config.LOV_CLASSES
    os.makedirs(os.path.join(root, "data", "0001"))
    np.savetxt(os.path.join(root, "extents.txt"), config.LOV_EXTENTS[1:], fmt="%.6f")
    for i, c in enumerate(classes[1:]):
        os.makedirs(os.path.join(root, "models", c))
        np.savetxt(os.path.join(root, "models", c, "points.xyz"), rng.uniform(-0.05, 0.05, (40 + i, 3)), fmt="%.6f")
    idx, frames = [], []
    for f in range(n_frames):
        name = "0001/%06d" % (f + 1)
        idx.append(name)
        color = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)           # RGB on disk
        depth = rng.integers(0, 30000, (H, W)).astype(np.uint16)
        label = np.zeros((H, W), np.uint8); label[4:12, 5:20] = 3; label[14:22, 10:30] = 13
        Image.fromarray(color).save(os.path.join(root, "data", name + "-color.png"))
        Image.fromarray(depth).save(os.path.join(root, "data", name + "-depth.png"))
        Image.fromarray(label).save(os.path.join(root, "data", name + "-label.png"))
        poses = np.zeros((3, 4, 2)); poses[:, :3, 0] = np.eye(3); poses[:, :3, 1] = np.eye(3)
        poses[:, 3, 0] = (0.1, 0.0, 0.8); poses[:, 3, 1] = (-0.1, 0.05, 1.0)
        scipy.io.savemat(os.path.join(root, "data", name + "-meta.mat"),
                         {"intrinsic_matrix": config.DEMO_INTRINSICS, "factor_depth": np.array([[10000]]),
                          "poses": poses, "cls_indexes": np.array([[3], [13]])})
        frames.append((color, depth, label, poses))
    with open(os.path.join(root, "keyframe.txt"), "w") as fh:
        fh.write("\n".join(idx) + "\n")
    return frames


def test_reader_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    frames = write_tree(str(tmp_path), rng)
    ds = datasets.YCBVideo(str(tmp_path), "keyframe")
    assert len(ds) == 3 and ds.num_classes == 22
    assert np.allclose(ds.extents, config.LOV_EXTENTS, atol=1e-6) and ds.extents[0].sum() == 0
    pts, pts_all = ds.points
    assert pts_all.shape == (22, 40, 3) and pts_all.dtype == np.float32       # cut to the shortest model (lov.py:152-156)
    assert pts[5].shape == (44, 3) and np.allclose(pts_all[5], pts[5][:40], atol=1e-6)
    fr = ds.frame(1)
    color, depth, label, poses = frames[1]
    assert fr["index"] == "0001/000002"
    assert np.array_equal(fr["color"], color[:, :, ::-1])                     # BGR, like cv2.imread
    assert fr["depth"].dtype == np.uint16 and np.array_equal(fr["depth"], depth)
    assert np.array_equal(fr["label"], label)
    assert np.allclose(fr["meta"]["intrinsic_matrix"], config.DEMO_INTRINSICS)
    assert fr["meta"]["poses"].shape == (3, 4, 2) and list(fr["meta"]["cls_indexes"]) == [3, 13]
    with pytest.raises(FileNotFoundError):
        datasets.YCBVideo(str(tmp_path), "no_such_set")


def test_fast_hist_and_segmentation_summary():
    gt = np.array([0, 0, 1, 1, 2, 2, 2, 5])      # 5 is outside n = 3: ignored
    pr = np.array([0, 1, 1, 1, 2, 0, 2, 1])
    h = datasets.fast_hist(gt, pr, 3)
    assert h.tolist() == [[1, 1, 0], [0, 2, 0], [1, 0, 2]]
    ev = datasets.Evaluator(("bg", "a", "b"), np.ones((3, 3)), [np.zeros((1, 3))] * 3)
    ev.hist += h
    s = ev.summary()
    assert np.isclose(s["overall_accuracy"], 5 / 7)
    assert np.isclose(s["per_class_iu"]["bg"], 1 / 3) and np.isclose(s["per_class_iu"]["a"], 2 / 3) and np.isclose(s["per_class_iu"]["b"], 2 / 3)
    assert np.isclose(s["mean_iu"], (1 / 3 + 2 / 3 + 2 / 3) / 3)
    assert np.isclose(s["fwavacc"], (2 / 7) * (1 / 3) + (2 / 7) * (2 / 3) + (3 / 7) * (2 / 3))


def test_evaluate_result_add_and_adds(tmp_path):
    rng = np.random.default_rng(1)
    write_tree(str(tmp_path), rng, n_frames=1)
    ds = datasets.YCBVideo(str(tmp_path))
    fr = ds.frame(0)
    ev = datasets.Evaluator(ds.classes, ds.extents, ds.points[0])
    labels = fr["label"].copy(); labels[4:6, 5:20] = 0                         # 30 of class 3's 120 pixels missed
    # detections: class 3 with the exact gt pose, class 13 (024_bowl: ADD-S) 30 cm off, class 7 without a gt object
    rois = np.array([[0, 3, 0, 0, 1, 1, 9], [0, 13, 0, 0, 1, 1, 9], [0, 7, 0, 0, 1, 1, 9]], F)
    poses = np.array([[1, 0, 0, 0, 0.1, 0.0, 0.8], [1, 0, 0, 0, -0.1, 0.05, 1.3], [1, 0, 0, 0, 0, 0, 1]], F)
    out = ev.evaluate_result(labels, rois, poses, fr["label"], fr["meta"], mat_path=str(tmp_path / "r.mat"))
    assert np.isclose(out["iou"]["004_sugar_box"], 90 / 120) and out["iou"]["024_bowl"] == 1.0
    assert [p["class"] for p in out["poses"]] == ["004_sugar_box", "024_bowl"]
    a, b = out["poses"]
    assert a["correct"] and a["error"] < 1e-6 and a["rotation_error_deg"] < 1e-3 and a["translation_error"] < 1e-6
    assert np.isclose(a["threshold"], 0.1 * np.linalg.norm(config.LOV_EXTENTS[3]))
    assert not b["correct"] and 0.2 < b["error"] <= 0.3 and np.isclose(b["translation_error"], 0.3, atol=1e-6)
    # ADD-S really is nearest-neighbour: it can only be <= ADD for the same poses
    pts = ds.points[0][13]
    assert b["error"] <= pose_error.add(np.eye(3), poses[1, 4:], np.eye(3), fr["meta"]["poses"][:, 3, 1], pts) + 1e-12
    s = ev.summary()
    assert s["poses_all"][2] == 1 and s["poses_correct"][2] == 1 and s["pose_accuracy"]["004_sugar_box"] == 1.0
    assert s["pose_accuracy"]["024_bowl"] == 0.0 and s["pose_accuracy"]["002_master_chef_can"] is None
    import scipy.io
    m = scipy.io.loadmat(str(tmp_path / "r.mat"))
    assert np.array_equal(m["labels"], labels) and m["rois"].shape == (3, 7) and m["poses"].shape == (3, 7)
    rep = ev.write_reports(str(tmp_path / "out"))
    assert rep["frames"] == 1 and len(open(str(tmp_path / "out" / "segmentation.txt")).read().split()) == 22


# ---- the reference's own fixtures (data, not code): only where the tree is mounted ---------------------
needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "data", "LOV")), reason="reference tree absent (GPU box)")


@needs_ref
def test_config_constants_equal_the_reference_files():
    ext = datasets.load_object_extents(os.path.join(REF, "data", "LOV", "extents.txt"), 22)
    assert np.array_equal(ext, config.LOV_EXTENTS)                            # lov.py:161-170
    lm = datasets.load_object_extents(os.path.join(REF, "data", "LINEMOD", "extents.txt"), 16)
    assert np.array_equal(lm, config.LINEMOD_EXTENTS_ALL) and np.array_equal(lm[:14], config.LINEMOD_EXTENTS)
    classes = [l.strip() for l in open(os.path.join(REF, "data", "LOV", "classes.txt")) if l.strip()]
    assert tuple(["__background__"] + classes) == config.LOV_CLASSES
    import json
    cam = json.load(open(os.path.join(REF, "data", "LOV", "camera.json")))["rig"]["camera"][0]["camera_model"]["params"]
    K = config.DEMO_INTRINSICS
    assert (K[0, 0], K[1, 1], K[0, 2], K[1, 2]) == tuple(cam[:4])             # tools/demo.py:100


@needs_ref
def test_real_model_points_and_demo_frames():
    pts, pts_all = datasets.load_object_points(os.path.join(REF, "data", "LOV", "models"), config.LOV_CLASSES)
    assert pts_all.shape == (22, config.NUM_MODEL_POINTS, 3) and min(p.shape[0] for p in pts[1:]) == 2620
    # every model fits its extent box (that is what extents.txt holds)
    for c in range(1, 22):
        span = pts[c].max(0) - pts[c].min(0)
        assert np.all(span <= config.LOV_EXTENTS[c] * 1.001 + 1e-4), config.LOV_CLASSES[c]
    for i in range(1, 6):
        color = datasets.read_color_bgr(os.path.join(REF, "data", "demo_images", "%06d-color.png" % i))
        depth = datasets.read_depth(os.path.join(REF, "data", "demo_images", "%06d-depth.png" % i))
        assert color.shape == (480, 640, 3) and color.dtype == np.uint8
        assert depth.shape == (480, 640) and depth.dtype == np.uint16 and 0 < np.median(depth[depth > 0]) / config.DEMO_FACTOR_DEPTH < 3.0
