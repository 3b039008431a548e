#!/usr/bin/env python
"""Debug aid: where does the fused F(4x4,3x3) GEMM+output kernel differ from the unfused pair?"""
import sys, os
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from posecnn_amd import ops

torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda:0")
B, H, W, C, cout, pool = 1, 30, 44, 128, 128, False
g = torch.Generator(device=dev).manual_seed(1)
x = torch.relu(torch.randn((B, H, W, C), device=dev, generator=g))
w = torch.randn((cout, C, 3, 3), device=dev, generator=g) * (2.0 / (9 * C)) ** 0.5
b = torch.randn(cout, device=dev, generator=g)
u = ops.winograd_filter(w, 4)
v = ops.winograd_input(x, 4)
want = ops.winograd_output(torch.bmm(v, u), b, B, H, W, True, pool, 4)
for rep in range(3):
    got = ops.winograd43_conv(v, u.transpose(1, 2).contiguous(), b, B, H, W, True, 1 if pool else 0)
    d = (got - want).abs()
    bad = (d > 1e-3).nonzero()
    print("rep", rep, "bad", bad.shape[0], "of", d.numel(), "max", float(d.max()))
    if bad.shape[0]:
        bb = bad.cpu().numpy()
        print(" rows", np.unique(bb[:, 1])[:40], "\n cols", np.unique(bb[:, 2])[:60], "\n chans", np.unique(bb[:, 3])[:140])
        print(" tiles(ty,tx):", sorted(set((int(r) // 4, int(c) // 4) for _, r, c, _ in bb))[:60])
