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


def make_planted_scene(frame_idx, H=480, W=640, C=22, num_units=64, n_obj=5, extents=None, K=None,
                       amplitude=30.0, stride=8):
    """Benchmark aid (DESIGN.md §synthetic workload). With random weights the label / vertex heads
    output noise, so the Hough layer would see no objects at all. This builds, at the 1/8
    resolution of `add_score` / `add_score_vertex` (vgg16_convs.py:135-136,159-160), additive
    feature maps that — through the real bilinear deconvs and the planted identity 1x1 heads of
    `init_planted_heads` — make the network emit the label map and vertex field of a synthetic
    scene with `n_obj` objects. Every layer still runs at full cost on dense data.
    Returns dict(add_score [h,w,num_units], add_score_vertex [h,w,128], objects)."""
    rng = np.random.default_rng(SEED0 + int(frame_idx))
    extents = LOV_EXTENTS if extents is None else np.asarray(extents, dtype=np.float32)
    if K is None:
        K = DEMO_INTRINSICS.copy()
        K[:2, :] *= W / 640.0
    fx, fy = K[0, 0], K[1, 1]
    h, w = H // stride, W // stride
    n_obj = min(n_obj, C - 1)
    classes = rng.choice(np.arange(1, C), size=n_obj, replace=False)
    objs = []
    for cls in classes:
        objs.append((int(cls), rng.uniform(0.15 * W, 0.85 * W), rng.uniform(0.15 * H, 0.85 * H), rng.uniform(0.6, 1.2)))
    objs.sort(key=lambda o: -o[3])
    # centres of the low-res cells in full-res pixel coordinates
    yy, xx = np.mgrid[0:h, 0:w]
    yy = yy * stride + (stride - 1) / 2.0
    xx = xx * stride + (stride - 1) / 2.0
    label = np.zeros((h, w), dtype=np.int32)
    for cls, cx, cy, z in objs:
        ax = max(0.5 * fx * float(extents[cls % len(extents), 0]) / z, 14.0)
        ay = max(0.5 * fy * float(extents[cls % len(extents), 1]) / z, 14.0)
        label[((xx - cx) / ax) ** 2 + ((yy - cy) / ay) ** 2 <= 1.0] = cls
    add_score = np.zeros((h, w, num_units), dtype=np.float32)
    for c in range(C):
        add_score[..., c] = amplitude * (label == c)
    add_vertex = np.zeros((h, w, 128), dtype=np.float32)
    for cls, cx, cy, z in objs:
        ang = np.arctan2(cy - yy, cx - xx) + rng.standard_normal((h, w)) * 0.05
        add_vertex[..., 3 * cls + 0] = amplitude * np.cos(ang)
        add_vertex[..., 3 * cls + 1] = amplitude * np.sin(ang)
        add_vertex[..., 3 * cls + 2] = np.log(z)
    return {"add_score": add_score, "add_score_vertex": add_vertex, "objects": objs, "label_lowres": label}


def make_planted_batch(first_idx, B, **kw):
    scenes = [make_planted_scene(first_idx + i, **kw) for i in range(B)]
    return {"add_score": np.stack([s["add_score"] for s in scenes]),
            "add_score_vertex": np.stack([s["add_score_vertex"] for s in scenes])}, scenes


def init_planted_heads(net, seed=11):
    """Give the head 1x1 convolutions the reference's own initialiser scale
    (tf.truncated_normal(stddev=0.001), network.py:170) and make the two final 1x1 convs
    ('score' 64->C, 'vertex_pred' 128->3C) identity maps on their first channels, so that the
    planted 1/8-resolution scene of make_planted_scene reaches the Hough layer. The backbone
    keeps its random (He) weights. Shapes follow Network.conv's [c_out, c_in, k, k] storage."""
    import torch
    g = torch.Generator().manual_seed(seed)
    C, U = net.num_classes, net.num_units

    def tn(shape):
        w = torch.empty(shape)
        torch.nn.init.trunc_normal_(w, 0.0, 0.001, -0.002, 0.002, generator=g)
        return w.contiguous(memory_format=torch.channels_last).to(net.device)

    cin = 1024 if net.input_format == "RGBD" else 512
    net.vars["score_conv5/weights"] = tn((U, cin, 1, 1))
    net.vars["score_conv4/weights"] = tn((U, cin, 1, 1))
    net.vars["score_conv5_vertex/weights"] = tn((128, 512, 1, 1))
    net.vars["score_conv4_vertex/weights"] = tn((128, 512, 1, 1))
    ws = torch.zeros((C, U, 1, 1))
    for c in range(C):
        ws[c, c, 0, 0] = 1.0
    net.vars["score/weights"] = ws.contiguous(memory_format=torch.channels_last).to(net.device)
    wv = torch.zeros((3 * C, 128, 1, 1))
    for c in range(3 * C):
        wv[c, c, 0, 0] = 1.0
    net.vars["vertex_pred/weights"] = wv.contiguous(memory_format=torch.channels_last).to(net.device)
    for n, k in (("score_conv5", U), ("score_conv4", U), ("score_conv5_vertex", 128), ("score_conv4_vertex", 128),
                 ("score", C), ("vertex_pred", 3 * C)):
        net.vars[n + "/biases"] = torch.zeros((k,), device=net.device)


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


def make_gt_poses(scenes, K, seed=5, max_tilt=0.15):
    """`poses` blob [N,13] = (batch, cls, box4, quat wxyz, trans3) (the pose_blob of
    lib/gt_synthesize_layer/minibatch.py, fed to the Hough layer as bottom_gt) for planted scenes:
    every object sits at its planted centre and depth with a small random rotation, so the
    projected 3-D box overlaps the detected box (IoU > 0.2, hough_voting_gpu_op.cu.cc:440-466) and
    the layer emits pose targets — which is what gives average_distance_loss real rows to work on."""
    rng = np.random.default_rng(seed)
    fx, fy, px, py = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    rows = []
    for b, s in enumerate(scenes):
        for cls, cx, cy, z in s["objects"]:
            q = np.concatenate([[1.0], rng.standard_normal(3) * max_tilt])
            q /= np.linalg.norm(q)
            rows.append([b, cls, 0, 0, 0, 0, q[0], q[1], q[2], q[3], (cx - px) / fx * z, (cy - py) / fy * z, z])
    return np.asarray(rows, dtype=np.float32).reshape(-1, 13)
