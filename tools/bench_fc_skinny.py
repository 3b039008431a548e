#!/usr/bin/env python
"""fc6 / fc7 / fc8 at the single-frame row counts: the weight-streaming kernel (csrc/fc_skinny.hip) next to the
64x64-block kernel with split-K (csrc/fc_mfma.hip). Prints one JSON object (times in us, weight GB/s)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from posecnn_amd import ops  # noqa: E402


def timeit(fn, iters=30, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return 1000.0 * ts[len(ts) // 2]


def main():
    dev = torch.device("cuda:0")
    res = {}
    for name, K, N, act in (("fc6", 25088, 4096, "relu"), ("fc7", 4096, 4096, "relu"), ("fc8", 4096, 88, "tanh")):
        wt = torch.randn((N, K), device=dev) / K ** 0.5
        b = torch.randn((N,), device=dev)
        for cap, live in ((21, 5), (21, 21), (8, 5), (32, 32)):
            x = torch.randn((cap, K), device=dev)
            cnt = torch.tensor([live], dtype=torch.int32, device=dev)
            us = timeit(lambda: ops.fc_skinny(x, wt, b, act, num_rows=cnt))
            e = {"skinny_us": round(us, 1), "weight_GBps": round(4.0 * N * K / us / 1e3, 1)}
            if N % 64 == 0:
                us2 = timeit(lambda: ops.fc_rows(x, wt, b, act == "relu", num_rows=cnt))
                e.update({"fc_rows_us": round(us2, 1), "fc_rows_weight_GBps": round(4.0 * N * K / us2 / 1e3, 1)})
            res["%s cap=%d live=%d" % (name, cap, live)] = e
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
