"""Pose error measures (lib/utils/pose_error.py contract) against brute-force definitions."""
import numpy as np

from posecnn_amd import pose_error as pe
from posecnn_amd import synth


def rot(axis, deg):
    a = np.radians(deg); c, s = np.cos(a), np.sin(a)
    return {"x": np.array([[1, 0, 0], [0, c, -s], [0, s, c]]), "y": np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]]),
            "z": np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])}[axis]


def test_quat2mat_and_errors():
    rng = np.random.default_rng(0)
    q = synth.random_unit_quats(rng, 5)
    for qi in q:
        R = pe.quat2mat(qi)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-12) and abs(np.linalg.det(R) - 1) < 1e-12
    assert np.allclose(pe.quat2mat([np.cos(np.pi / 8), 0, 0, np.sin(np.pi / 8)]), rot("z", 45))
    assert abs(pe.re(rot("x", 30), np.eye(3)) - 30) < 1e-9
    assert abs(pe.re(rot("y", 200), np.eye(3)) - 160) < 1e-9
    assert abs(pe.te([0, 0, 1], [0, 3, 5]) - 5) < 1e-12


def test_add_adi_reproj():
    rng = np.random.default_rng(1)
    pts = rng.standard_normal((300, 3)) * 0.05
    R1, R2 = rot("z", 10) @ rot("x", 20), rot("y", 35)
    t1, t2 = np.array([0.1, 0.0, 0.8]), np.array([0.12, -0.02, 0.83])
    a = (pts @ R1.T + t1); b = (pts @ R2.T + t2)
    assert abs(pe.add(R1, t1, R2, t2, pts) - np.linalg.norm(a - b, axis=1).mean()) < 1e-12
    brute = np.sqrt(((b[:, None] - a[None]) ** 2).sum(-1)).min(1).mean()
    assert abs(pe.adi(R1, t1, R2, t2, pts) - brute) < 1e-9
    assert pe.adi(R1, t1, R2, t2, pts) <= pe.add(R1, t1, R2, t2, pts) + 1e-12
    assert pe.add(R1, t1, R1, t1, pts) == 0 and pe.adi(R1, t1, R1, t1, pts) == 0
    K = np.array([[500.0, 0, 320], [0, 500, 240], [0, 0, 1]])
    pa, pb = a @ K.T, b @ K.T
    want = np.linalg.norm(pa[:, :2] / pa[:, 2:] - pb[:, :2] / pb[:, 2:], axis=1).mean()
    assert abs(pe.reproj(K, R1, t1, R2, t2, pts) - want) < 1e-9


def test_evaluate_detections_matches_rows_by_image_and_class():
    rng = np.random.default_rng(2)
    points = rng.standard_normal((4, 50, 3)) * 0.05
    q = synth.random_unit_quats(rng, 2)
    gt = np.zeros((2, 13)); gt[0, :2] = (0, 2); gt[0, 6:10] = q[0]; gt[0, 10:] = (0, 0, 1)
    gt[1, :2] = (1, 3); gt[1, 6:10] = q[1]; gt[1, 10:] = (0.1, 0, 1)
    rois = np.array([[0, 2, 0, 0, 1, 1, 9], [1, 3, 0, 0, 1, 1, 9], [1, 1, 0, 0, 1, 1, 9]], float)
    poses = np.zeros((3, 7)); poses[0, :4] = q[0]; poses[0, 4:] = (0, 0, 1.01); poses[1, :4] = q[1]; poses[1, 4:] = (0.1, 0, 1)
    poses[2, 0] = 1
    res = pe.evaluate_detections(rois, poses, gt, points, [0, 0, 0, 1])
    assert [(r["image"], r["cls"]) for r in res] == [(0, 2), (1, 3)]
    assert abs(res[0]["add"] - 0.01) < 1e-9 and abs(res[0]["te"] - 0.01) < 1e-9 and res[0]["re"] < 1e-5
    assert res[1]["add"] < 1e-12 and res[1]["symmetric"]
