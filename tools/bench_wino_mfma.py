#!/usr/bin/env python
"""Per-layer timing of the fused Winograd F(4x4,3x3) fp32-MFMA kernel (csrc/wino_mfma.hip) at the bench
configuration, next to the input transform and the round-1 path (library batched GEMM + output transform
kernel). Prints one JSON object.
    python tools/bench_wino_mfma.py [--batch 16] [--groups 2] [--layers conv1_2,conv4_2]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from posecnn_amd import ops  # noqa: E402

LAYERS = [("conv1_2", 64, 64, 1, 1), ("conv2_1", 64, 128, 2, 0), ("conv2_2", 128, 128, 2, 1), ("conv3_1", 128, 256, 4, 0),
          ("conv3_2", 256, 256, 4, 0), ("conv3_3", 256, 256, 4, 1), ("conv4_1", 256, 512, 8, 0), ("conv4_2", 512, 512, 8, 0),
          ("conv4_3", 512, 512, 8, 2), ("conv5_1", 512, 512, 16, 0), ("conv5_2", 512, 512, 16, 0), ("conv5_3", 512, 512, 16, 0)]


def timeit(fn, iters=8, warmup=2):
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
    ap.add_argument("--batch", type=int, default=16, help="frames per tower")
    ap.add_argument("--groups", type=int, default=2, help="towers in one launch")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--layers", default="")
    ap.add_argument("--no-library", action="store_true", help="skip the library GEMM comparison")
    a = ap.parse_args()
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cuda.preferred_blas_library("hipblas")
    dev = torch.device("cuda:0")
    G, B = a.groups, a.batch * a.groups
    res = {"frames_per_tower": a.batch, "groups": G, "layers": {}}
    tot_mfma = tot_in = tot_lib = tot_fl = 0.0
    with torch.no_grad():
        for name, ci, co, div, pool in LAYERS:
            if a.layers and name not in a.layers.split(","):
                continue
            H, W = a.height // div, a.width // div
            x = torch.relu(torch.randn((B, H, W, ci), device=dev))
            w = torch.randn((G, co, ci, 3, 3), device=dev) * (2.0 / (9 * ci)) ** 0.5
            b = torch.randn((G, co), device=dev)
            ut = torch.stack([ops.winograd_filter(w[g], 4).transpose(1, 2) for g in range(G)]).contiguous()
            v = ops.winograd_input(x, 4)
            fl = 2.0 * 36 * v.shape[1] * ci * co
            ms_in = timeit(lambda: ops.winograd_input(x, 4))
            ms = timeit(lambda: ops.winograd43_conv(v, ut, b, B, H, W, True, pool, G))
            ms0 = timeit(lambda: ops.winograd43_conv(v, ut, b, B, H, W, True, 0, G)) if pool else ms
            e = {"tiles": v.shape[1], "cin": ci, "cout": co, "pool": pool, "input_transform_ms": round(ms_in, 4),
                 "mfma_ms": round(ms, 4), "mfma_TFLOPs": round(fl / ms / 1e9, 1), "mfma_nopool_ms": round(ms0, 4)}
            if not a.no_library:
                u0 = ops.winograd_filter(w[0], 4)
                ms_mm = timeit(lambda: torch.bmm(v, u0))
                m = torch.bmm(v, u0)
                ms_out = timeit(lambda: ops.winograd_output(m, b[0], B, H, W, True, pool == 1, 4))
                del m
                e.update({"library_gemm_ms": round(ms_mm, 4), "library_gemm_TFLOPs": round(fl / ms_mm / 1e9, 1),
                          "output_transform_ms": round(ms_out, 4), "round1_path_ms": round(ms_mm + ms_out, 4)})
                tot_lib += ms_mm + ms_out
            res["layers"][name] = e
            tot_mfma += ms; tot_in += ms_in; tot_fl += fl
            del x, v, ut
    res["total"] = {"mfma_ms": round(tot_mfma, 3), "input_transform_ms": round(tot_in, 3), "mfma_TFLOPs": round(tot_fl / tot_mfma / 1e9, 1),
                    "round1_gemm_plus_output_ms": round(tot_lib, 3)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
