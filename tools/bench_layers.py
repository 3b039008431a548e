#!/usr/bin/env python
"""Per-layer timings of the dense trunk at the bench configuration (B=16, 640x480): every VGG16
3x3 convolution as torch/MIOpen runs it (fp32, channels-last, MIOpen find), plus the bias/ReLU and
pooling passes, with TF/s per layer. Prints one JSON object.
    python tools/bench_layers.py [--batch 16]
"""
import argparse
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from posecnn_amd import ops  # noqa: E402

LAYERS = [("conv1_1", 3, 64, 1), ("conv1_2", 64, 64, 1), ("conv2_1", 64, 128, 2), ("conv2_2", 128, 128, 2),
          ("conv3_1", 128, 256, 4), ("conv3_2", 256, 256, 4), ("conv3_3", 256, 256, 4), ("conv4_1", 256, 512, 8),
          ("conv4_2", 512, 512, 8), ("conv4_3", 512, 512, 8), ("conv5_1", 512, 512, 16), ("conv5_2", 512, 512, 16),
          ("conv5_3", 512, 512, 16)]


def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--layers", default="", help="comma-separated subset, e.g. conv1_2,conv2_1")
    a = ap.parse_args()
    torch.backends.cudnn.benchmark = True
    torch.backends.cudnn.allow_tf32 = False
    dev = torch.device("cuda:0")
    B = a.batch
    res = {"batch": B, "layers": {}}
    tot = 0.0
    with torch.no_grad():
        for name, ci, co, div in LAYERS:
            if a.layers and name not in a.layers.split(","):
                continue
            H, W = a.height // div, a.width // div
            x = torch.randn((B, ci, H, W), device=dev).contiguous(memory_format=torch.channels_last)
            w = torch.randn((co, ci, 3, 3), device=dev).contiguous(memory_format=torch.channels_last) * 0.05
            b = torch.randn(co, device=dev)
            ms = timeit(lambda: F.conv2d(x, w, None, padding=1))
            y = F.conv2d(x, w, None, padding=1).permute(0, 2, 3, 1).contiguous()
            ms_b = timeit(lambda: ops.bias_act_(y, b, True))
            fl = 2.0 * B * H * W * ci * co * 9
            res["layers"][name] = {"conv_ms": round(ms, 4), "TFLOPs": round(fl / ms / 1e9, 1), "bias_act_ms": round(ms_b, 4),
                                   "act_GBps": round(2 * y.numel() * 4 / ms_b / 1e6, 0)}
            if ci >= 64 and ci % 4 == 0:
                xh = x.permute(0, 2, 3, 1).contiguous()
                u = ops.winograd_filter(w)
                ms_w = timeit(lambda: ops.conv3x3_winograd(xh, u, b, True))
                ms_in = timeit(lambda: ops.winograd_input(xh))
                v = ops.winograd_input(xh)
                ms_mm = timeit(lambda: torch.bmm(v, u))
                res["layers"][name].update({"winograd_total_ms": round(ms_w, 4), "winograd_input_ms": round(ms_in, 4),
                                            "winograd_gemm_ms": round(ms_mm, 4), "winograd_gemm_TFLOPs": round(fl / 2.25 / ms_mm / 1e9, 1),
                                            "direct_plus_bias_ms": round(ms + ms_b, 4)})
                del v
                u4 = ops.winograd_filter(w, 4)
                ms_w4 = timeit(lambda: ops.conv3x3_winograd(xh, u4, b, True, tile=4))
                ms_in4 = timeit(lambda: ops.winograd_input(xh, 4))
                v4 = ops.winograd_input(xh, 4)
                ms_mm4 = timeit(lambda: torch.bmm(v4, u4))
                res["layers"][name].update({"winograd43_total_ms": round(ms_w4, 4), "winograd43_input_ms": round(ms_in4, 4),
                                            "winograd43_gemm_ms": round(ms_mm4, 4),
                                            "winograd43_gemm_TFLOPs": round(2.0 * v4.numel() * co / ms_mm4 / 1e9, 1)})
                del v4
            if ci == 3:
                xh = x.permute(0, 2, 3, 1).contiguous()
                wh = w.permute(2, 3, 1, 0).contiguous()
                ms_f = timeit(lambda: ops.conv3x3_c3(xh, wh, b, True))
                res["layers"][name].update({"fused_conv_bias_relu_ms": round(ms_f, 4),
                                            "fused_write_GBps": round(y.numel() * 4 / ms_f / 1e6, 0)})
            if name in ("conv1_2", "conv2_2", "conv3_3", "conv4_3"):
                ms_p = timeit(lambda: F.max_pool2d(y.permute(0, 3, 1, 2), 2, 2))
                ms_f = timeit(lambda: ops.bias_relu_pool2(y, b, True))
                res["layers"][name].update({"max_pool_ms": round(ms_p, 4), "fused_bias_relu_pool2_ms": round(ms_f, 4),
                                            "fused_GBps": round(1.25 * y.numel() * 4 / ms_f / 1e6, 0)})
            tot += ms
    res["conv_total_ms"] = round(tot, 3)
    if not a.layers:
        res["conv_total_TFLOPs"] = round(sum(2.0 * B * (a.height // d) * (a.width // d) * ci * co * 9 for _, ci, co, d in LAYERS) / tot / 1e9, 1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
