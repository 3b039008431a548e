#!/usr/bin/env python
"""Per-op timings of the custom gfx950 kernels at the bench configuration (B=16, 640x480, C=22),
measured with HIP events on the launch stream. Prints one JSON object. Usage:
    python tools/bench_ops.py [--batch 16] [--iters 20] [--ops hough,hard_label,...]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from posecnn_amd import _lib, config, ops, synth  # noqa: E402


def timeit(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return {"ms_median": ts[len(ts) // 2], "ms_min": ts[0], "ms_max": ts[-1]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--thr", type=float, default=50.0)
    ap.add_argument("--ops", default="hough,hough_lowres,hough_thr,hard_label,softmax,roi_pool,adl,backproject,trunk,upscore,smooth_l1")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B, H, W, C = a.batch, a.height, a.width, 22
    res = {"batch": B, "H": H, "W": W, "C": C}
    which = set(a.ops.split(","))
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    K = config.DEMO_INTRINSICS.copy(); K[:2] *= W / 640.0
    meta = T(np.stack([config.make_meta_data(K)] * B))
    ext = T(config.LOV_EXTENTS)

    if which & {"hough", "hough_thr", "roi_pool", "hough_lowres"}:
        label_np, vertex_np, _ = synth.make_batch(0, B, H=H, W=W, C=C, K=K)
        label, vertex = T(label_np), T(vertex_np)
        nfg = int((label_np > 0).sum())
        out = ops.hough_voting_gpu_padded(label, vertex, ext, meta, None, 0, -1.0, 0.02, 10)
        torch.cuda.synchronize()
        nroi = int(out[5][1])
        rois = out[0][:max(nroi, 1)].clone()
        if "hough" in which:
            ws = ops.Workspace()
            r = timeit(lambda: ops.hough_voting_gpu_padded(label, vertex, ext, meta, None, 0, -1.0, 0.02, 10, workspace=ws, out=out), a.iters)
            alg = 4 * H * W * B + 12 * nfg + 56 * nroi
            r.update({"rois": nroi, "fg_pixels": nfg, "algorithmic_bytes": alg, "GBps_algorithmic": alg / r["ms_median"] / 1e6,
                      "streamed_field_bytes": 4 * H * W * (1 + 3 * C) * B})
            res["hough"] = r
        if "hough_empty" in which:
            # all-background labels: every hv_vote block exits at once -> launch/dispatch floor of the sequence
            zl = torch.zeros_like(label)
            _lib.profile_enable(True)
            r = timeit(lambda: ops.hough_voting_gpu_padded(zl, vertex, ext, meta, None, 0, -1.0, 0.02, 10), a.iters)
            r["kernels_us"] = {k: round(v["avg_us"], 2) for k, v in _lib.profile_report().items()}
            _lib.profile_enable(False)
            res["hough_empty"] = r
        if "hough" in which:
            _lib.profile_enable(True)
            for _ in range(5):
                ops.hough_voting_gpu_padded(label, vertex, ext, meta, None, 0, -1.0, 0.02, 10)
            torch.cuda.synchronize()
            res["hough"]["kernels_us"] = {k: round(v["avg_us"], 2) for k, v in _lib.profile_report().items()}
            _lib.profile_enable(False)
        if "hough_lowres" in which:
            # the fused vertex head path: 1/8-resolution field, interpolated for the sampled pixels only
            z = vertex[:, 4::8, 4::8, :].contiguous()
            zb = torch.zeros(3 * C, device=dev)
            r = timeit(lambda: ops.hough_voting_gpu_lowres_padded(label, z, zb, 16, 8, ext, meta, None, 0, -1.0, 0.02, 10), a.iters)
            r["note"] = "vs materialising vertex_pred first: + deconv_bilinear below"
            r["deconv_vertex_pred"] = timeit(lambda: ops.deconv_bilinear(z, 16, 8, bias=zb), a.iters)
            res["hough_lowres"] = r
        if "hough_thr" in which:
            r = timeit(lambda: ops.hough_voting_gpu_padded(label, vertex, ext, meta, None, 0, a.thr, 0.002, 10), a.iters)
            _lib.profile_enable(True)
            for _ in range(3):
                ops.hough_voting_gpu_padded(label, vertex, ext, meta, None, 0, a.thr, 0.002, 10)
            torch.cuda.synchronize()
            r["kernels_us"] = {k: round(v["avg_us"], 2) for k, v in _lib.profile_report().items()}
            _lib.profile_enable(False)
            res["hough_vote_threshold_50"] = r
        if "roi_pool" in which:
            g = torch.Generator(device=dev).manual_seed(1)
            c5 = torch.randn((B, H // 16, W // 16, 512), device=dev, generator=g)
            c4 = torch.randn((B, H // 8, W // 8, 512), device=dev, generator=g)
            r5 = timeit(lambda: ops.roi_pool(c5, rois, 7, 7, 1 / 16.0, 0), a.iters)
            r4 = timeit(lambda: ops.roi_pool(c4, rois, 7, 7, 1 / 8.0, 0), a.iters)
            rf = timeit(lambda: ops.roi_pool_add2(c5, 1 / 16.0, c4, 1 / 8.0, rois), a.iters)
            res["roi_pool"] = {"rois": int(rois.shape[0]), "pool5": r5, "pool4": r4, "fused_add2": rf,
                               "out_bytes_each": int(rois.shape[0]) * 49 * 512 * 8}
    if which & {"hard_label", "softmax"}:
        g = torch.Generator(device=dev).manual_seed(2)
        score = torch.relu(torch.randn((B, H, W, C), device=dev, generator=g) * 3)
        gt = torch.randint(0, C, (B, H, W), device=dev, generator=g, dtype=torch.int32)
        if "softmax" in which:
            r = timeit(lambda: ops.softmax_argmax(score), a.iters)
            byt = B * H * W * (8 * C + 4)
            r.update({"algorithmic_bytes": byt, "GBps": byt / r["ms_median"] / 1e6})
            res["softmax_argmax"] = r
        if "hard_label" in which:
            prob, _ = ops.softmax_argmax(score)
            r = timeit(lambda: ops.hard_label(prob, gt, 1.0), a.iters)
            byt = B * H * W * (4 + 4 + 4 * C)
            r.update({"algorithmic_bytes": byt, "GBps": byt / r["ms_median"] / 1e6})
            res["hard_label"] = r
    if "trunk" in which:
        g = torch.Generator(device=dev).manual_seed(5)
        x = torch.randn((B, H, W, 3), device=dev, generator=g) * 50
        w = torch.randn((3, 3, 3, 64), device=dev, generator=g) * 0.1
        bias = torch.randn(64, device=dev, generator=g)
        r = timeit(lambda: ops.conv3x3_c3(x, w, bias, True), a.iters)
        byt = B * H * W * (3 + 64) * 4
        r.update({"algorithmic_bytes": byt, "GBps": byt / r["ms_median"] / 1e6})
        res["conv3x3_c3_bias_relu"] = r
        y = torch.randn((B, H, W, 64), device=dev, generator=g)
        r = timeit(lambda: ops.bias_relu_pool2(y, bias, True), a.iters)
        byt = int(B * H * W * 64 * 4 * 1.25)
        r.update({"algorithmic_bytes": byt, "GBps": byt / r["ms_median"] / 1e6})
        res["bias_relu_pool2_conv1_2"] = r
        r = timeit(lambda: ops.bias_act_(y, bias, True), a.iters)
        byt = B * H * W * 64 * 4 * 2
        r.update({"algorithmic_bytes": byt, "GBps": byt / r["ms_median"] / 1e6})
        res["bias_act_conv1"] = r
    if "upscore" in which:
        g = torch.Generator(device=dev).manual_seed(6)
        z = torch.randn((B, H // 8, W // 8, C), device=dev, generator=g)
        zb = torch.randn(C, device=dev, generator=g)
        r = timeit(lambda: ops.upscore_softmax_argmax(z, zb, 16, 8), a.iters)
        byt = B * H * W * (C + 1) * 4
        r.update({"algorithmic_bytes": byt, "GBps": byt / r["ms_median"] / 1e6})
        res["upscore_softmax_argmax"] = r
        z64 = torch.randn((B, H // 8, W // 8, 64), device=dev, generator=g)
        r = timeit(lambda: ops.deconv_bilinear(z64, 16, 8), max(3, a.iters // 2))
        byt = B * H * W * 64 * 4
        r.update({"algorithmic_bytes": byt, "GBps": byt / r["ms_median"] / 1e6})
        res["deconv_bilinear_64ch_x8"] = r
        gout = torch.randn((B, H, W, 64), device=dev, generator=g)
        r = timeit(lambda: ops.deconv_bilinear_grad(gout, 16, 8), max(3, a.iters // 2))
        r.update({"algorithmic_bytes": byt, "GBps": byt / r["ms_median"] / 1e6})
        res["deconv_bilinear_bwd_64ch_x8"] = r
    if "smooth_l1" in which:
        g = torch.Generator(device=dev).manual_seed(7)
        n = (B, H, W, 3 * C)
        p = torch.randn(n, device=dev, generator=g).requires_grad_(True)
        t = torch.randn(n, device=dev, generator=g)
        wt = (torch.rand(n, device=dev, generator=g) < 0.1).float()
        r = timeit(lambda: ops.smooth_l1_loss_vertex(p.detach(), t, wt), max(3, a.iters // 2))
        byt = p.numel() * 4 * 3
        r.update({"algorithmic_bytes": byt, "GBps": byt / r["ms_median"] / 1e6})
        res["smooth_l1_vertex_fwd"] = r

        def fb():
            p.grad = None
            ops.smooth_l1_loss_vertex(p, t, wt).backward()
        r = timeit(fb, max(3, a.iters // 2))
        byt = p.numel() * 4 * 7
        r.update({"algorithmic_bytes": byt, "GBps": byt / r["ms_median"] / 1e6})
        res["smooth_l1_vertex_fwd_bwd"] = r
    if "adl" in which:
        rng = np.random.default_rng(3)
        P = config.NUM_MODEL_POINTS
        R = 5 * B
        pts = T(synth.make_model_points(C, P)); sym = T(config.LOV_SYMMETRY)
        pred = np.zeros((R, 4 * C), np.float32); tgt = pred.copy(); wgt = pred.copy()
        for n in range(R):
            c = [1, 16, 5, 21, 9][n % 5]
            pred[n, 4 * c:4 * c + 4] = np.tanh(rng.standard_normal(4)); tgt[n, 4 * c:4 * c + 4] = synth.random_unit_quats(rng, 1)[0]; wgt[n, 4 * c:4 * c + 4] = 1
        pred, tgt, wgt = T(pred), T(tgt), T(wgt)
        r = timeit(lambda: ops.average_distance_loss(pred, tgt, wgt, pts, sym, 0.01), a.iters)
        r.update({"rois": R, "symmetric_rois": 2 * B, "pair_evals": 2 * B * P * P})
        res["average_distance_loss"] = r
    if "backproject" in which:
        g = torch.Generator(device=dev).manual_seed(4)
        G, Cd = 128, 64
        data = torch.randn((1, H, W, Cd), device=dev, generator=g)
        lab = torch.rand((1, H, W, C), device=dev, generator=g)
        depth = (1.5 + 0.5 * torch.rand((1, H, W, 1), device=dev, generator=g))
        l3 = torch.rand((1, G, G, G, C), device=dev, generator=g)
        ident = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]], np.float32)
        m = T(config.make_meta_data(K, voxel_step=(6.0 / G, 6.0 / G, 7.0 / G), voxel_min=(-3, -3, -3), pose_world2live=ident, pose_live2world=ident).reshape(1, 1, 1, 48))
        r = timeit(lambda: ops.backproject(data, lab, depth, m, l3, G, 3, 0.02), max(3, a.iters // 4))
        byt = G ** 3 * (2 * Cd + C) * 4 + G ** 3 * C * 4
        r.update({"G": G, "bytes_written_plus_label3d": byt, "GBps": byt / r["ms_median"] / 1e6})
        res["backproject_G128"] = r
    print(json.dumps(res))


if __name__ == "__main__":
    main()
