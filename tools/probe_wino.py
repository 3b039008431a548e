#!/usr/bin/env python
"""Runs the Winograd F(4x4,3x3) transform kernels (and the fused first conv) a few times at bench
shapes so that `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes can attribute HBM traffic
to them:   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -o p -- python tools/probe_wino.py
Prints the algorithmic bytes per dispatch of each case as JSON."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from posecnn_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    B = 16
    out = {}
    g = torch.Generator(device=dev).manual_seed(0)
    for name, H, W, C in (("conv3_2", 120, 160, 256), ("conv4_2", 60, 80, 512)):
        x = torch.randn((B, H, W, C), device=dev, generator=g)
        bias = torch.randn(C, device=dev, generator=g)
        xb = x.numel() * 4
        for _ in range(3):
            v = ops.winograd_input(x, 4)
            y = ops.winograd_output(v, bias, B, H, W, True, False, 4)
            yp = ops.winograd_output(v, bias, B, H, W, True, True, 4)
        torch.cuda.synchronize()
        out[name] = {"X_bytes": xb, "input_transform_algorithmic": int(3.25 * xb), "output_transform_algorithmic": int(3.25 * xb),
                     "output_pool_algorithmic": int(2.5 * xb)}
        del v, y, yp
    x = torch.randn((B, 480, 640, 3), device=dev, generator=g) * 50
    w = torch.randn((3, 3, 3, 64), device=dev, generator=g) * 0.1
    b = torch.randn(64, device=dev, generator=g)
    for _ in range(3):
        v = ops.conv3x3_c3_winograd43(x, w, b, True)
    torch.cuda.synchronize()
    out["conv1_1_fused"] = {"algorithmic": int(x.numel() * 4 + v.numel() * 4)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
