"""Backprojecting layer by itself (lib/backprojecting_layer/backprojecting_op_gpu.cu.cc:17-126): the two grids that matter —
G = 256 (the reference default, lib/fcn/config.py:106,222) on the parity tests' smooth scene and G = 128 on the LINEMOD bench's
random-depth frames — timed with events, HBM fraction printed. `--once` runs each shape a few times without timing (what the
rocprofv3 --pmc passes of tools/pmc_kernel.sh wrap).   python tools/bench_backproject.py [--once] [--grids 256,128]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from posecnn_amd import config, ops  # noqa: E402


def scene(G, H=480, W=640, Cd=64, Cl=22, B=1, kind="smooth"):
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(7)
    data = torch.randn((B, H, W, Cd), generator=g, device=dev)
    label = torch.rand((B, H, W, Cl), generator=g, device=dev)
    l3 = torch.rand((B, G, G, G, Cl), generator=g, device=dev)
    if kind == "smooth":   # tests/test_gpu_ops.py::backproject_case + tests/test_gpu_round4.py::_meta_for_grid
        yy, xx = np.mgrid[0:H, 0:W]
        rng = np.random.default_rng(43)
        depth = (1.6 + 0.3 * np.sin(xx / 9.0) + 0.2 * np.cos(yy / 7.0) + 0.01 * rng.standard_normal((B, H, W))).astype(np.float32)
        K = np.array([[W * 0.9, 0, W / 2.0], [0, W * 0.9, H / 2.0], [0, 0, 1]])
        a = 0.05
        w2l = np.array([[np.cos(a), 0, np.sin(a), 0.01], [0, 1, 0, -0.02], [-np.sin(a), 0, np.cos(a), 0.03]], np.float32)
        l2w = np.array([[np.cos(a), 0, -np.sin(a), -0.01], [0, 1, 0, 0.02], [np.sin(a), 0, np.cos(a), -0.03]], np.float32)
        meta = np.stack([config.make_meta_data(K, voxel_step=(2.4 / G, 2.0 / G, 1.2 / G), voxel_min=(-1.2, -1.0, 1.1),
                                               pose_world2live=w2l, pose_live2world=l2w)] * B)
        thr = 0.05
    else:                  # bench.py --config linemod: per-pixel random depth, identity pose
        K = config.DEMO_INTRINSICS.copy()
        K[:2] *= W / 640.0
        ident = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]], np.float32)
        meta = np.stack([config.make_meta_data(K, voxel_step=(6.0 / G, 6.0 / G, 7.0 / G), voxel_min=(-3, -3, -3),
                                               pose_world2live=ident, pose_live2world=ident)] * B)
        depth = (1.5 + 0.5 * torch.rand((B, H, W), generator=torch.Generator().manual_seed(99))).numpy()
        thr = 0.02
    return (data, label, torch.from_numpy(depth.reshape(B, H, W, 1)).to(dev), torch.from_numpy(meta.reshape(B, 1, 1, 48).astype(np.float32)).to(dev),
            l3, G, 3, thr)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--once", action="store_true")
    ap.add_argument("--grids", default="256,128")
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--kinds", default="smooth,random")
    a = ap.parse_args()
    out = {}
    for G in [int(x) for x in a.grids.split(",")]:
        for kind, (H, W, Cl, B) in (("smooth", (480, 640, 22, 1)), ("random", (960, 1280, 14, 4 if G <= 128 else 1))):
            if kind not in a.kinds.split(","):
                continue
            args = scene(G, H, W, 64, Cl, B, kind)
            nv = B * G ** 3
            byts = 4.0 * (nv * (2 * 64 + Cl) + nv * Cl + B * H * W * (64 + Cl + 1))
            res = ops.backproject(*args)
            torch.cuda.synchronize()
            hit = float((res[2].view(nv, 64)[:, 0] != 0).float().mean())
            times = []
            for _ in range(2 if a.once else a.iters):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                del res
                e0.record()
                res = ops.backproject(*args)
                e1.record()
                torch.cuda.synchronize()
                times.append(e0.elapsed_time(e1))
            del res
            ms = min(times)
            out["G%d_%s_%dx%dx%d_C%d" % (G, kind, B, H, W, Cl)] = {"ms": round(ms, 3), "GB": round(byts / 1e9, 2), "TBps": round(byts / ms / 1e9, 3),
                                                                  "frac_of_8TBps": round(byts / ms / 1e9 / 8.0, 3), "hit_fraction": round(hit, 4)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
