#!/usr/bin/env python
"""Times the fused pool_score op (roi_pool_add2) and roi_pool fwd/bwd at the bench shapes:
16 frames, 480x640 -> conv5_3 30x40x512 + conv4_3 60x80x512, 468 ROI rows of YCB-like boxes."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posecnn_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    B, R = 16, 468
    c5 = torch.randn(B, 30, 40, 512, device=dev)
    c4 = torch.randn(B, 60, 80, 512, device=dev)
    rois = np.zeros((R, 7), np.float32)
    rois[:, 0] = np.arange(R) % B
    rois[:, 1] = rng.integers(1, 22, R)
    cx, cy = rng.uniform(120, 520, R), rng.uniform(100, 380, R)
    w, h = rng.uniform(60, 260, R), rng.uniform(60, 260, R)
    rois[:, 2], rois[:, 3], rois[:, 4], rois[:, 5] = cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2
    rois_t = torch.from_numpy(rois).to(dev)

    def timeit(fn, n=50):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6

    print("roi_pool_add2 (pool_score): %.1f us" % timeit(lambda: ops.roi_pool_add2(c5, 1 / 16.0, c4, 1 / 8.0, rois_t)))
    d = c5.clone().requires_grad_(True)
    top, arg = ops.roi_pool(d, rois_t, 7, 7, 1 / 16.0, 0)
    print("roi_pool fwd conv5_3: %.1f us" % timeit(lambda: ops.roi_pool(c5, rois_t, 7, 7, 1 / 16.0, 0)))
    print("roi_pool fwd conv4_3: %.1f us" % timeit(lambda: ops.roi_pool(c4, rois_t, 7, 7, 1 / 8.0, 0)))
    g = torch.randn_like(top)
    print("roi_pool bwd conv5_3: %.1f us" % timeit(lambda: torch.autograd.grad(top, d, g, retain_graph=True)))


if __name__ == "__main__":
    main()
