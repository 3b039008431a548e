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


# Per-layer gains on top of He initialisation that give every layer an O(1) output on the synthetic
# frames (uniform uint8 colour, uniform uint16 depth < 3000): LSUV-style calibration, measured ONCE on the
# host by tools/calibrate_synth.py (two 480x640 RGB-D frames, seed 11) and frozen here so that every
# process — CPU checker, GPU tests, bench.py — builds bit-identical weights without running a calibration
# pass. Targets: post-ReLU std 1 for every trunk layer, score_conv4/5, fc6, fc7; 0.05 for the two vertex
# score convs (the planted log-depth channel then reads 0.5-2 m for every class); 0.8 for fc8 (|fc8| <~ 3:
# tanh is not saturated, as in a trained network — the reference trains from sigma = 0.001,
# lib/networks/network.py:170, and a converged PoseCNN has O(1-10) activations behind conv1).
CALIBRATED_GAINS = {
    "conv1_1": 0.01671, "conv1_2": 0.9871, "conv2_1": 0.6956, "conv2_2": 0.9289, "conv3_1": 0.7484,
    "conv3_2": 0.9021, "conv3_3": 1.031, "conv4_1": 0.7821, "conv4_2": 1.058, "conv4_3": 1.018, "conv5_1": 0.8867,
    "conv5_2": 1.027, "conv5_3": 1, "conv1_1_p": 0.01152, "conv1_2_p": 1.024, "conv2_1_p": 0.6101,
    "conv2_2_p": 1.04, "conv3_1_p": 0.8679, "conv3_2_p": 1.037, "conv3_3_p": 0.9219, "conv4_1_p": 0.8914,
    "conv4_2_p": 0.952, "conv4_3_p": 1.048, "conv5_1_p": 0.9093, "conv5_2_p": 0.9724, "conv5_3_p": 1.02,
    "score_conv5": 1.173, "score_conv4": 1.041, "score_conv5_vertex": 0.03071, "score_conv4_vertex": 0.02808,
    "fc6": 0.5393, "fc7": 0.99, "fc8": 0.4564,
}
CALIBRATED_TARGET_STD = {"score_conv5_vertex": 0.05, "score_conv4_vertex": 0.05, "fc8": 0.8}


def calibrated_layers(input_format, num_classes, num_units):
    """(name, weight shape, fan_in, kind) of every variable-carrying layer in creation order of
    vgg16_convs.setup() (vgg16_convs.py:36-197)."""
    from .networks import vgg16_convs
    out = []
    towers = ("", "_p") if input_format == "RGBD" else ("",)
    for sfx in towers:
        for name, ci, co, _ in vgg16_convs.TRUNK:
            out.append((name + sfx, (co, ci, 3, 3), 9 * ci, "conv"))
    cin = 512 * len(towers)
    out += [("score_conv5", (num_units, cin, 1, 1), cin, "conv"), ("score_conv4", (num_units, cin, 1, 1), cin, "conv"),
            ("score_conv5_vertex", (128, 512, 1, 1), 512, "conv"), ("score_conv4_vertex", (128, 512, 1, 1), 512, "conv"),
            ("fc6", (7 * 7 * 512, 4096), 7 * 7 * 512, "fc"), ("fc7", (4096, 4096), 4096, "fc"),
            ("fc8", (4096, 4 * num_classes), 4096, "fc")]
    return out


def init_calibrated(net, seed=11, gains=None):
    """Seeded weights with realistic activation scales for the whole network (VERDICT r3 "Next" #1): He
    initialisation times the frozen per-layer gain of CALIBRATED_GAINS, zero biases, and the planted identity
    heads of `init_planted_heads` ('score', 'vertex_pred'). Generated on the host from one torch.Generator,
    so the CPU checker and the GPU network hold the same bits. Call before the first `run`."""
    import math
    import torch
    g = torch.Generator(device="cpu").manual_seed(seed)
    gains = CALIBRATED_GAINS if gains is None else gains
    C, U = net.num_classes, net.num_units
    for name, shape, fan_in, kind in calibrated_layers(net.input_format, C, U):
        w = torch.randn(shape, generator=g) * (math.sqrt(2.0 / fan_in) * float(gains.get(name, 1.0)))
        if kind == "conv":
            w = w.contiguous(memory_format=torch.channels_last)
        net.vars[name + "/weights"] = w.to(net.device)
        net.vars[name + "/biases"] = torch.zeros((shape[0] if kind == "conv" else shape[1],), device=net.device)
    ws = torch.zeros((C, U, 1, 1))
    for c in range(C):
        ws[c, c, 0, 0] = 1.0
    net.vars["score/weights"] = ws.contiguous(memory_format=torch.channels_last).to(net.device)
    wv = torch.zeros((3 * C, 128, 1, 1))
    for c in range(3 * C):
        wv[c, c, 0, 0] = 1.0
    net.vars["vertex_pred/weights"] = wv.contiguous(memory_format=torch.channels_last).to(net.device)
    net.vars["score/biases"] = torch.zeros((C,), device=net.device)
    net.vars["vertex_pred/biases"] = torch.zeros((3 * C,), device=net.device)
    return net


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
