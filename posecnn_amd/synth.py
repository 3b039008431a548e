"""Deterministic synthetic PoseCNN frames.

A frame is what the label/vertex heads would hand to the Hough layer: a label map with a few
elliptical object masks and a vertex field whose own-class channels point at the object centre
(with angular noise) and carry log-depth, mirroring the target construction of
lib/gt_synthesize_layer/minibatch.py:583-594. Everything is seeded from the frame index.
"""
import numpy as np

from .config import DEMO_INTRINSICS, LOV_EXTENTS

SEED0 = 20180626


def make_frame(frame_idx, H=480, W=640, C=22, n_obj=5, extents=None, K=None, min_pixels=501,
               dir_noise=0.05, depth_noise=0.02, bg_noise=0.1, dtype=np.float32):
    """Returns dict(label int32 [H,W], vertex f32 [H,W,3C], centers [(cls,cx,cy,z)], K)."""
    rng = np.random.default_rng(SEED0 + int(frame_idx))
    extents = LOV_EXTENTS if extents is None else np.asarray(extents, dtype=np.float32)
    if K is None:
        K = DEMO_INTRINSICS.copy()
        K[:2, :] *= W / 640.0  # same rule as lib/fcn/test.py:130-131 (K * im_scale)
    fx, fy = K[0, 0], K[1, 1]
    n_obj = min(n_obj, C - 1)
    classes = rng.choice(np.arange(1, C), size=n_obj, replace=False)
    objs = []
    for cls in classes:
        cx = rng.uniform(0.15 * W, 0.85 * W)
        cy = rng.uniform(0.15 * H, 0.85 * H)
        z = rng.uniform(0.6, 1.2)
        objs.append((int(cls), cx, cy, z))
    objs.sort(key=lambda o: -o[3])  # paint far -> near
    label = np.zeros((H, W), dtype=np.int32)
    yy, xx = np.mgrid[0:H, 0:W]
    for cls, cx, cy, z in objs:
        ax = max(0.5 * fx * float(extents[cls % len(extents), 0]) / z, 14.0)
        ay = max(0.5 * fy * float(extents[cls % len(extents), 1]) / z, 14.0)
        m = ((xx - cx) / ax) ** 2 + ((yy - cy) / ay) ** 2 <= 1.0
        label[m] = cls
    vertex = (rng.standard_normal((H, W, 3 * C)) * bg_noise).astype(dtype)
    for cls, cx, cy, z in objs:
        m = label == cls
        n = int(m.sum())
        if n == 0:
            continue
        dx = cx - xx[m]
        dy = cy - yy[m]
        ang = np.arctan2(dy, dx) + rng.standard_normal(n) * dir_noise
        vertex[m, 3 * cls + 0] = np.cos(ang)
        vertex[m, 3 * cls + 1] = np.sin(ang)
        vertex[m, 3 * cls + 2] = np.log(z) + rng.standard_normal(n) * depth_noise
    return {"label": label, "vertex": vertex, "objects": objs, "K": K, "min_pixels": min_pixels}


def make_batch(first_idx, B, **kw):
    frames = [make_frame(first_idx + i, **kw) for i in range(B)]
    label = np.stack([f["label"] for f in frames])
    vertex = np.stack([f["vertex"] for f in frames])
    return label, vertex, frames


def make_model_points(C, P, extents=None, seed=7):
    """Stand-in for data/LOV/models/*/points.xyz: P points inside each class' extent box."""
    rng = np.random.default_rng(seed)
    extents = LOV_EXTENTS if extents is None else np.asarray(extents, dtype=np.float32)
    pts = np.zeros((C, P, 3), dtype=np.float32)
    for c in range(1, C):
        e = extents[c % len(extents)]
        pts[c] = (rng.uniform(-0.5, 0.5, size=(P, 3)) * e).astype(np.float32)
    return pts


def random_unit_quats(rng, n):
    q = rng.standard_normal((n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return q.astype(np.float32)
