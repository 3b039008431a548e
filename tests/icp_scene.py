"""Test scenes for the ICP slice (TEST INFRASTRUCTURE): an analytic box renderer standing in for the reference's OpenGL
vertex / normal render (lib/synthesize/synthesize.cpp:1972-1991) and small SE(3) helpers."""
import numpy as np


def rot(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


def pose(R, t):
    T = np.zeros((3, 4))
    T[:, :3], T[:, 3] = R, t
    return T


def compose(A, B):
    """A * B for 3x4 rigid transforms"""
    return pose(A[:, :3] @ B[:, :3], A[:, :3] @ B[:, 3] + A[:, 3])


def pose_error(A, B):
    """(rotation angle in degrees, translation distance) between two 3x4 poses"""
    Rd = A[:, :3] @ B[:, :3].T
    ang = np.degrees(np.arccos(np.clip((np.trace(Rd) - 1) / 2, -1, 1)))
    return ang, float(np.linalg.norm(A[:, 3] - B[:, 3]))


def render_box(T, half, K, H, W):
    """Ray-cast an axis-aligned box (half extents `half`, object frame) placed at pose T (camera <- object).
    Returns (vertex_map f32 [H,W,3] in camera coordinates, 0 on the background; normal_map f32 [H,W,3]; mask)."""
    R, t = T[:, :3], T[:, 3]
    ys, xs = np.mgrid[0:H, 0:W]
    d = np.stack([(xs - K[0, 2]) / K[0, 0], (ys - K[1, 2]) / K[1, 1], np.ones_like(xs, dtype=np.float64)], axis=-1)   # camera rays (z = 1)
    o = -R.T @ t                      # camera centre in the object frame
    dd = d @ R                        # ray directions in the object frame (R^T d)
    half = np.asarray(half, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        t1 = (-half - o) / dd
        t2 = (half - o) / dd
    tn, tf = np.minimum(t1, t2), np.maximum(t1, t2)
    tmin, tmax = tn.max(-1), tf.min(-1)
    hit = (tmax >= tmin) & (tmin > 0)
    axis = tn.argmax(-1)
    n_obj = np.zeros((H, W, 3))
    sgn = -np.sign(np.take_along_axis(dd, axis[..., None], -1))[..., 0]
    np.put_along_axis(n_obj, axis[..., None], sgn[..., None], -1)
    vert = d * tmin[..., None]        # camera frame: z = tmin because the rays have z = 1
    norm = n_obj @ R.T
    vert[~hit] = 0
    norm[~hit] = 0
    return vert.astype(np.float32), norm.astype(np.float32), hit


def scene(T_true, T_init, half, K, H, W, factor=10000.0, obj_id=3, noise=0.0, rng=None):
    """live depth (uint16, quantised by `factor`) + label from the box at T_true; predicted maps from the box at T_init"""
    v_true, _, hit = render_box(T_true, half, K, H, W)
    z = v_true[..., 2].astype(np.float64)
    if noise and rng is not None:
        z = z + rng.standard_normal(z.shape) * noise * hit
    depth = np.clip(np.round(z * factor), 0, 65535).astype(np.uint16)
    label = np.where(hit, obj_id, 0).astype(np.int32)
    pv, pn, _ = render_box(T_init, half, K, H, W)
    return depth, label, pv, pn
