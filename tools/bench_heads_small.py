#!/usr/bin/env python
"""head_lowres (csrc/heads_small.hip) against the op sequence it replaces (deconv kernel + 2 adds + library 1x1 conv)."""
import os, sys, json, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from posecnn_amd import ops
from tools.bench_fc_skinny import timeit
dev = torch.device("cuda:0")
res = {}
for B in (1, 16):
    for U, Cout in ((64, 22), (128, 66)):
        a = torch.randn((B, 60, 80, U), device=dev); b5 = torch.randn((B, 30, 40, U), device=dev); pl = torch.randn_like(a)
        w = torch.randn((Cout, U, 1, 1), device=dev).contiguous(memory_format=torch.channels_last); wt = w.reshape(Cout, U).t().contiguous()
        us = timeit(lambda: ops.head_lowres(a, b5, wt, planted=pl))
        def old():
            t = a + ops.deconv_bilinear(b5, 4, 2); t = t + pl
            return F.conv2d(t.permute(0, 3, 1, 2), w).permute(0, 2, 3, 1).contiguous()
        res["B=%d %d->%d" % (B, U, Cout)] = {"head_lowres_us": round(us, 1), "op_sequence_us": round(timeit(old), 1)}
print(json.dumps(res, indent=1))
