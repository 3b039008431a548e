#!/usr/bin/env python
"""Pose refinement of one 640x480 frame (posecnn_amd.icp.Synthesizer.icp_python = Synthesizer::solveICP, its nlopt polish as the library's Nelder-Mead):
5 objects, meshes of 81 920 triangles each (the size of a YCB `textured_simple.obj`), poses 2 cm off in depth.
Prints one JSON object: wall time per frame and per object, per-kernel times from the library's own HIP-event profiler,
and the same flow on the CPU checker for ONE object (tests/icp_scene.solve_icp_reference, bounded sample)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from posecnn_amd import _lib, config, icp  # noqa: E402


def main():
    import icp_scene as S
    dev = torch.device("cuda:0")
    H, W = 480, 640
    K = config.DEMO_INTRINSICS.copy()
    sub = int(os.environ.get("ICP_BENCH_SUBDIV", "6"))
    rng = np.random.default_rng(7)
    shapes = [(0.05, (1.0, 0.7, 1.3)), (0.06, (0.8, 0.8, 1.2)), (0.045, (1.3, 1.0, 0.7)), (0.055, (1.0, 1.0, 1.0)), (0.05, (0.6, 1.2, 1.0))]
    centres = [(-0.18, -0.1, 0.75), (0.0, -0.1, 0.85), (0.18, -0.08, 0.8), (-0.1, 0.1, 0.7), (0.12, 0.1, 0.9)]
    meshes_np, truths = [], []
    depth = np.zeros((H, W), np.uint16)
    label = np.zeros((H, W), np.int32)
    for c, ((r, sc), ctr) in enumerate(zip(shapes, centres)):
        m = S.icosphere(r, sub, scale=sc)
        Tt = S.pose(S.rot(rng.standard_normal(3), rng.uniform(0, 3)), ctr)
        meshes_np.append(m)
        truths.append(Tt)
    gm = [icp.Mesh(m[0], m[2], m[1], device=dev) for m in meshes_np]
    for c, (g, Tt) in enumerate(zip(gm, truths)):
        v = icp.render(g, Tt[None], K, H, W, want=("vertices",))["vertices"][0].cpu().numpy()
        hit = np.isfinite(v[..., 2])
        depth = np.where(hit, np.clip(np.round(np.where(hit, v[..., 2], 0) * 10000.0), 0, 65535), depth).astype(np.uint16)
        label = np.where(hit, c + 1, label).astype(np.int32)
    R = len(gm)
    rois = np.zeros((R, 7), np.float32)
    poses = np.zeros((R, 7), np.float32)
    for r in range(R):
        rois[r, 1] = r + 1
        est = S.pose(S.rot([0, 1, 0], 0.02) @ truths[r][:, :3], truths[r][:, 3] * (1 + 0.02 / truths[r][2, 3]))
        poses[r, :4] = icp.mat2quat(est[:, :3])
        poses[r, 4:] = est[:, 3]
    params = np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2], 0.25, 6.0, 10000.0], np.float32)
    syn = icp.Synthesizer(meshes=gm, device=dev)
    syn.setup(W, H)
    out, out_icp = np.zeros((R, 7), np.float32), np.zeros((R, 7), np.float32)

    def frame():
        syn.icp_python(label, depth, params, H, W, R, 7, rois, poses, out, out_icp, 0.01)

    for _ in range(3):
        frame()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        frame()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    _lib.profile_enable(True)
    _lib.profile_report()
    frame()
    torch.cuda.synchronize()
    rep = _lib.profile_report()
    _lib.profile_enable(False)
    errs = []
    for r in range(R):
        Ti = np.zeros((3, 4))
        Ti[:, :3] = icp.quat2mat(out_icp[r, :4].astype(np.float64))
        Ti[:, 3] = out_icp[r, 4:]
        errs.append(S.pose_error(Ti, truths[r]))
    res = {"frame": "%dx%d, %d objects, %d triangles per mesh, %d label pixels" % (H, W, R, len(meshes_np[0][2]), int((label > 0).sum())),
           "ms_per_frame": round(ms, 3), "ms_per_object": round(ms / R, 3),
           "kernels": {k: {"calls": v["calls"], "total_us": round(v["total_ms"] * 1e3, 1)} for k, v in sorted(rep.items())},
           "kernel_ms_per_frame": round(sum(v["total_ms"] for v in rep.values()), 3),
           "translation_error_mm": [round(e[1] * 1e3, 3) for e in errs], "rotation_error_deg": [round(e[0], 3) for e in errs],
           "hits": [[int(h) for h in info["hits"]] for info in syn.last]}
    # CPU checker, one object (the flow is serial per object)
    q_t = poses[0].astype(np.float64)
    T_in = np.zeros((3, 4))
    T_in[:, :3] = icp.quat2mat(q_t[:4])
    T_in[:, 3] = q_t[4:]
    t0 = time.perf_counter()
    ref = S.solve_icp_reference(label, depth, K, 10000.0, 1, T_in, meshes_np[0], q_t=q_t)
    res["cpu_checker_ms_per_object"] = round((time.perf_counter() - t0) * 1e3, 1)
    res["cpu_checker_note"] = "tests/icp_scene.solve_icp_reference on 1 core; its SegICP search is exhaustive (O(pairs^2)), the product's is windowed"
    res["matches_checker"] = bool(np.array_equal(ref["hits"], syn.last[0]["hits"]))
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
