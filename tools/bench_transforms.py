#!/usr/bin/env python
"""The two HBM-bound transform kernels of the trunk at the bench shapes (2 towers x 16 frames): conv3x3_c3_wino43_kernel
(conv1_1 inside conv1_2's input transform) and wino43_input_kernel (conv2_2's input). Prints us and GB/s."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from posecnn_amd import ops
from tools.bench_fc_skinny import timeit
dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
x = torch.randn((32, 480, 640, 3), device=dev) * 50
w = torch.randn((2, 3, 3, 3, 64), device=dev) * 0.1
b = torch.randn((2, 64), device=dev)
us = timeit(lambda: ops.conv3x3_c3_winograd43(x, w, b, True, groups=2), iters, 2)
res = {"conv3x3_c3_wino43": {"us": round(us, 1), "GBps_written": round(2.25 * 32 * 480 * 640 * 64 * 4 / us / 1e3, 1)}}
for name, (H, W, C) in (("conv2_1 input", (240, 320, 64)), ("conv2_2 input", (240, 320, 128)), ("conv4_2 input", (60, 80, 512))):
    y = torch.randn((32, H, W, C), device=dev)
    us = timeit(lambda: ops.winograd_input(y, 4), iters, 2)
    res["wino43_input " + name] = {"us": round(us, 1), "GBps_read_plus_written": round(3.25 * y.numel() * 4 / us / 1e3, 1)}
print(json.dumps(res, indent=1))
