"""6D pose error measures used by the reference's evaluation (lib/utils/pose_error.py:12-117, called
from lib/datasets/lov.py `evaluation`): ADD, ADD-S (ADI), rotation / translation / reprojection
error. Host-side numpy, written from the definitions in Hinterstoisser et al. (ACCV 2012) and
Hodan et al. (ECCVW 2016); same function names and argument order as the reference module."""
import numpy as np


def quat2mat(q):
    """Unit quaternion (w, x, y, z) -> 3x3 rotation matrix (the convention of the pose rows:
    poses[:, :4], lib/fcn/test.py:206-211)."""
    w, x, y, z = [float(v) for v in q]
    n = w * w + x * x + y * y + z * z
    if n < 1e-16:
        return np.eye(3)
    s = 2.0 / n
    return np.array([[1 - s * (y * y + z * z), s * (x * y - z * w), s * (x * z + y * w)],
                     [s * (x * y + z * w), 1 - s * (x * x + z * z), s * (y * z - x * w)],
                     [s * (x * z - y * w), s * (y * z + x * w), 1 - s * (x * x + y * y)]])


def transform_pts_Rt(pts, R, t):
    """nx3 points -> R @ p + t, nx3."""
    pts = np.asarray(pts, dtype=np.float64)
    assert pts.ndim == 2 and pts.shape[1] == 3
    return pts @ np.asarray(R, dtype=np.float64).T + np.asarray(t, dtype=np.float64).reshape(1, 3)


def add(R_est, t_est, R_gt, t_gt, pts):
    """Average distance between corresponding model points under the two poses (ADD)."""
    d = transform_pts_Rt(pts, R_est, t_est) - transform_pts_Rt(pts, R_gt, t_gt)
    return float(np.linalg.norm(d, axis=1).mean())


def adi(R_est, t_est, R_gt, t_gt, pts, chunk=2048):
    """ADD-S: average distance from every ground-truth point to its nearest estimated point
    (for objects with indistinguishable views)."""
    pe = transform_pts_Rt(pts, R_est, t_est)
    pg = transform_pts_Rt(pts, R_gt, t_gt)
    try:
        from scipy.spatial import cKDTree
        dist, _ = cKDTree(pe).query(pg, k=1)
        return float(dist.mean())
    except ImportError:  # brute force, chunked
        best = np.empty(len(pg))
        for i in range(0, len(pg), chunk):
            d2 = ((pg[i:i + chunk, None, :] - pe[None, :, :]) ** 2).sum(-1)
            best[i:i + chunk] = np.sqrt(d2.min(axis=1))
        return float(best.mean())


def re(R_est, R_gt):
    """Rotation error in degrees: the angle of R_est @ R_gt^-1."""
    R_est, R_gt = np.asarray(R_est, dtype=np.float64), np.asarray(R_gt, dtype=np.float64)
    assert R_est.shape == (3, 3) and R_gt.shape == (3, 3)
    c = 0.5 * (np.trace(R_est @ np.linalg.inv(R_gt)) - 1.0)
    return float(np.degrees(np.arccos(min(1.0, max(-1.0, c)))))


def te(t_est, t_gt):
    """Translation error: Euclidean distance of the two translations."""
    t_est, t_gt = np.asarray(t_est, dtype=np.float64).ravel(), np.asarray(t_gt, dtype=np.float64).ravel()
    assert t_est.size == 3 and t_gt.size == 3
    return float(np.linalg.norm(t_gt - t_est))


def reproj(K, R_est, t_est, R_gt, t_gt, pts):
    """Mean pixel distance between the model points projected with the two poses."""
    K = np.asarray(K, dtype=np.float64)

    def project(R, t):
        p = transform_pts_Rt(pts, R, t) @ K.T
        return p[:, :2] / p[:, 2:3]

    return float(np.linalg.norm(project(R_est, t_est) - project(R_gt, t_gt), axis=1).mean())


def evaluate_detections(rois, poses, gt_poses, points, symmetry, K=None):
    """Per-detection ADD / ADD-S / re / te against the ground-truth rows of the same image and class.
    rois [R,7] (image, class, ...), poses [R,7] (quat wxyz, t) as returned by `im_segment_*`;
    gt_poses [N,13] in the Hough layer's layout (image, class, box4, quat4, trans3);
    points [C,P,3]; symmetry [C]. Returns a list of dicts (one per matched detection)."""
    out = []
    gt_poses = np.asarray(gt_poses)
    for r, p in zip(np.asarray(rois), np.asarray(poses)):
        m = (gt_poses[:, 0] == r[0]) & (gt_poses[:, 1] == r[1])
        if not m.any():
            continue
        g = gt_poses[m][0]
        cls = int(r[1])
        Re, Rg = quat2mat(p[:4]), quat2mat(g[6:10])
        pts = np.asarray(points)[cls]
        d = {"image": int(r[0]), "cls": cls, "add": add(Re, p[4:7], Rg, g[10:13], pts),
             "adds": adi(Re, p[4:7], Rg, g[10:13], pts), "re": re(Re, Rg), "te": te(p[4:7], g[10:13]),
             "symmetric": bool(np.asarray(symmetry)[cls] > 0)}
        if K is not None:
            d["reproj"] = reproj(K, Re, p[4:7], Rg, g[10:13], pts)
        out.append(d)
    return out
