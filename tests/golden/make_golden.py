#!/usr/bin/env python
"""Generates tests/golden/*.npz from the REFERENCE's own kernel bodies (oracle/_ref/libposecnn_ref.so,
built by `make -C oracle ref` where /root/reference exists): seeded inputs + the outputs the reference
kernels produce for them. The fixtures are small (tens of KB) and committed, so the parity tests keep a
reference-derived anchor even where the reference tree (and therefore _ref) is unavailable.

    python tests/golden/make_golden.py          # rewrites the fixtures
"""
import ctypes
import os
import sys
from ctypes import c_float, c_void_p

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402  (loads liboracle.so: the shim resolves oracle_expf from it)
from posecnn_amd import config, synth  # noqa: E402

F = np.float32


def p(a):
    return a.ctypes.data_as(c_void_p) if a is not None else c_void_p(0)


def hough_case(ref, first, B, H, W, C, n_obj, ext_scale, gt, is_train, vote_thr, per_thr, skip, label_thr):
    K = config.DEMO_INTRINSICS.copy(); K[:2] *= W / 640.0
    label, vertex, fr = synth.make_batch(first, B, H=H, W=W, C=C, n_obj=n_obj, K=K)
    vertex = vertex.astype(np.float16).astype(F)  # fp16-representable inputs: the fixture stores them as f16
    meta = np.stack([config.make_meta_data(K)] * B)
    ext = (config.LOV_EXTENTS[:C] * ext_scale).astype(F)
    if gt == "objects":
        rng = np.random.default_rng(first)
        rows = []
        for n in range(B):
            for (cls, cx, cy, z) in fr[n]["objects"]:
                q = synth.random_unit_quats(rng, 1)[0]
                rows.append([n, cls, 0, 0, 0, 0, q[0], q[1], q[2], q[3], (cx - K[0, 2]) / K[0, 0] * z, (cy - K[1, 2]) / K[1, 1] * z, z])
        gt = np.array(rows, F)
    cap = 128 * 9
    out = [np.empty((cap, 7), F), np.empty((cap, 7), F), np.empty((cap, 4 * C), F), np.empty((cap, 4 * C), F),
           np.empty(cap, np.int32), np.zeros(2, np.int32)]
    ref.ref_hough_voting(p(label), p(vertex), p(ext), p(meta), p(gt), B, H, W, C, 48, 0 if gt is None else len(gt),
                         int(is_train), c_float(vote_thr), c_float(per_thr), int(skip), c_float(0.9), int(label_thr),
                         *[p(o) for o in out], None)
    r = int(out[5][0])
    return dict(label=label.astype(np.int8), vertex=vertex.astype(np.float16), extents=ext, meta=meta,
                gt=np.zeros((0, 13), F) if gt is None else gt,
                params=np.array([is_train, vote_thr, per_thr, skip, label_thr], np.float64),
                top_box=out[0][:r], top_pose=out[1][:r], top_target=out[2][:r], top_weight=out[3][:r],
                top_domain=out[4][:r], num_rois=out[5])


def main():
    so = os.path.join(ROOT, "oracle", "_ref", "libposecnn_ref.so")
    if not os.path.exists(so):
        sys.exit("build oracle/_ref first: make -C oracle ref")
    oracle.lib()
    ref = ctypes.CDLL(so)
    rng = np.random.default_rng(2018)

    np.savez_compressed(os.path.join(HERE, "hough_default.npz"), **hough_case(ref, 300, 2, 72, 96, 5, 2, 2.0, None, 0, -1.0, 0.02, 4, 60))
    np.savez_compressed(os.path.join(HERE, "hough_threshold.npz"), **hough_case(ref, 310, 1, 72, 96, 5, 2, 2.0, None, 0, 3.0, 0.0005, 3, 60))
    np.savez_compressed(os.path.join(HERE, "hough_train.npz"), **hough_case(ref, 320, 2, 72, 96, 5, 2, 0.6, "objects", 1, -1.0, 0.02, 4, 60))

    # roi pooling (forward + backward)
    B, H, W, C, R = 2, 15, 20, 8, 12
    data = rng.standard_normal((B, H, W, C)).astype(F)
    rois = np.zeros((R, 7), F)
    rois[:, 0] = rng.integers(0, B, R); rois[:, 1] = rng.integers(0, C, R)
    x1 = rng.uniform(-40, 300, R); y1 = rng.uniform(-40, 220, R)
    rois[:, 2], rois[:, 3] = x1, y1
    rois[:, 4] = x1 + rng.uniform(-20, 200, R); rois[:, 5] = y1 + rng.uniform(-20, 200, R)
    rois[:3, 2:6] = np.round(rois[:3, 2:6] / 16) * 16 + 8
    top = np.empty((R, 7, 7, C), F); arg = np.empty((R, 7, 7, C), np.int32)
    ref.ref_roi_pool(p(data), p(rois), H, W, C, R, 7, 7, 7, c_float(1 / 16.0), 0, p(top), p(arg))
    g = rng.standard_normal(top.shape).astype(F); bd = np.empty((B, H, W, C), F)
    ref.ref_roi_pool_bwd(p(g), p(rois), p(arg), B, H, W, C, R, 7, 7, 7, c_float(1 / 16.0), 0, p(bd))
    np.savez_compressed(os.path.join(HERE, "roi_pool.npz"), data=data, rois=rois, top=top, argmax=arg, top_diff=g, bottom_diff=bd)

    # hard label
    prob = rng.random((1, 9, 11, 22)).astype(F); gt = rng.integers(-1, 22, (1, 9, 11)).astype(np.int32)
    out = np.empty_like(prob)
    ref.ref_hard_label(p(prob), p(gt), gt.size, 22, c_float(0.4), p(out))
    np.savez_compressed(os.path.join(HERE, "hard_label.npz"), prob=prob, gt=gt, out=out.astype(np.int8), threshold=F(0.4))

    # average distance loss (a symmetric and a non-symmetric class)
    C, P, R = 4, 80, 4
    pts = synth.make_model_points(C, P, extents=config.LOV_EXTENTS[:C] + 0.05)
    sym = np.array([0, 0, 1, 0], F)
    pred = np.zeros((R, 4 * C), F); tgt = np.zeros((R, 4 * C), F); wgt = np.zeros((R, 4 * C), F)
    for n in range(R):
        if n == 2:
            continue
        c = 1 + n % 3
        pred[n, 4 * c:4 * c + 4] = np.tanh(rng.standard_normal(4)); tgt[n, 4 * c:4 * c + 4] = synth.random_unit_quats(rng, 1)[0]
        wgt[n, 4 * c:4 * c + 4] = 1
    loss = np.zeros(1, F); diff = np.zeros((R, 4 * C), F)
    ref.ref_average_distance(p(pred), p(tgt), p(wgt), p(pts), p(sym), R, C, P, c_float(0.01), p(loss), p(diff))
    np.savez_compressed(os.path.join(HERE, "average_distance.npz"), pred=pred, target=tgt, weight=wgt, points=pts, symmetry=sym,
                        margin=F(0.01), loss=loss, bottom_diff=diff)

    # backproject (forward + backward)
    B, H, W, Cd, Cl, G = 1, 12, 16, 4, 3, 6
    data = rng.standard_normal((B, H, W, Cd)).astype(F); label = rng.random((B, H, W, Cl)).astype(F)
    depth = (1.5 + 0.3 * rng.random((B, H, W, 1))).astype(F); l3 = rng.random((B, G, G, G, Cl)).astype(F)
    Kb = np.array([[10.0, 0, 8.0], [0, 10.0, 6.0], [0, 0, 1]])
    a = 0.1
    w2l = np.array([[np.cos(a), -np.sin(a), 0, 0.02], [np.sin(a), np.cos(a), 0, -0.01], [0, 0, 1, 0.05]], F)
    l2w = np.array([[np.cos(a), np.sin(a), 0, -0.02], [-np.sin(a), np.cos(a), 0, 0.01], [0, 0, 1, -0.05]], F)
    meta = config.make_meta_data(Kb, voxel_step=(0.4, 0.3, 0.12), voxel_min=(-1.0, -0.8, 1.2), pose_world2live=w2l, pose_live2world=l2w)[None]
    td = np.empty((B, G, G, G, Cd), F); tf = np.empty((B, G, G, G, Cd), F); tl = np.empty((B, G, G, G, Cl), F)
    ref.ref_backproject(p(data), p(label), p(depth), p(meta), p(l3), B, H, W, Cd, Cl, 48, G, 1, c_float(0.08), p(td), p(tl), p(tf))
    gg = rng.standard_normal(td.shape).astype(F); bd = np.empty((B, H, W, Cd), F)
    ref.ref_backproject_bwd(p(gg), p(depth), p(meta), B, H, W, Cd, 48, G, p(bd))
    np.savez_compressed(os.path.join(HERE, "backproject.npz"), data=data, label=label, depth=depth, meta=meta, label_3d=l3,
                        top_data=td, top_label=tl, top_flag=tf, top_diff=gg, bottom_diff=bd, params=np.array([G, 1, 0.08]))
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
