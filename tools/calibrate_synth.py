#!/usr/bin/env python
"""Measures the per-layer gains frozen in posecnn_amd/synth.py::CALIBRATED_GAINS (VERDICT r3 "Next" #1).

LSUV-style: with He-initialised weights and zero biases the network is positively homogeneous layer by
layer, so one forward pass gives each layer's output std and the gain that brings it to its target
(1 for the trunk / score convs / fc6-7, see CALIBRATED_TARGET_STD for the rest). Three passes on the host
(PyTorch-CPU + the C checker for the custom layers, the same graph tests/cpu_reference.py runs):
  1. gains = 1: trunk stds -> trunk gains (exact by homogeneity, tower by tower);
  2. trunk calibrated: head / fc stds -> their gains (the heads mix two towers, fc6 sees pooled ROIs);
  3. verification: prints every layer's std with the final table.
The table is rounded to 4 significant digits and printed as a Python dict to paste into synth.py.
TEST/BENCH INFRASTRUCTURE — not on the product path. Usage: python tools/calibrate_synth.py [--frames 2]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from posecnn_amd import config, synth  # noqa: E402


def frames(n, H, W, seed=1234):
    import torch
    g = torch.Generator(device="cpu").manual_seed(seed)
    im = torch.randint(0, 256, (n, H, W, 3), generator=g, dtype=torch.uint8).float()
    depth = torch.randint(0, 3000, (n, H, W, 1), generator=g).float()
    data = (im - torch.from_numpy(config.PIXEL_MEANS)).float().numpy()
    data_p = ((torch.clamp(depth / 2000.0, 0, 1) * 255).expand(n, H, W, 3) - torch.from_numpy(config.PIXEL_MEANS)).float().numpy()
    return data, data_p


def measure(gains, n, H, W, C=22):
    from cpu_reference import run_cpu_pipeline, vgg16_convs_cpu
    net = vgg16_convs_cpu("RGBD", C, 64, (1.0,), 1.0, -1.0, vertex_reg_2d=True, pose_reg=True, trainable=False,
                          is_train=False, init="he", with_losses=False)
    synth.init_calibrated(net, gains=gains)
    K = config.DEMO_INTRINSICS.copy()
    data, data_p = frames(n, H, W)
    planted, _ = synth.make_planted_batch(1000, n, H=H, W=W, C=C, K=K)
    pts = synth.make_model_points(C, 64)
    out = run_cpu_pipeline(net, data, K, config.LOV_EXTENTS, pts, config.LOV_SYMMETRY, planted=planted, data_p=data_p)
    rows = out["rois"].shape[0]         # the checker's Hough layer returns exactly the detected rows
    std = {}
    for name, _, _, _ in synth.calibrated_layers("RGBD", C, 64):
        t = net.get_output(name)
        t = t[0] if isinstance(t, tuple) else t
        if name.startswith("fc"):
            t = t[:max(rows, 1)]
        std[name] = float(t.double().std())
    std["_abs_fc8_max"] = float(net.get_output("fc8")[:max(rows, 1)].abs().max())
    std["_rows"] = rows
    return std


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    a = ap.parse_args()
    tgt = lambda n: synth.CALIBRATED_TARGET_STD.get(n, 1.0)
    names = [n for n, _, _, _ in synth.calibrated_layers("RGBD", 22, 64)]
    gains = {n: 1.0 for n in names}
    s1 = measure(gains, a.frames, a.height, a.width)
    for sfx in ("", "_p"):
        prev = 1.0
        for n in names:
            if n.startswith("conv") and n.endswith("_p") == (sfx == "_p"):
                gains[n] = prev / s1[n]        # input now has std 1 instead of prev
                prev = s1[n]
    s2 = measure(gains, a.frames, a.height, a.width)
    for n in ("score_conv5", "score_conv4", "score_conv5_vertex", "score_conv4_vertex"):
        gains[n] = tgt(n) / s2[n]
    gains["fc6"] = tgt("fc6") / s2["fc6"]
    gains["fc7"] = tgt("fc7") * s2["fc6"] / s2["fc7"]
    gains["fc8"] = tgt("fc8") * s2["fc7"] / s2["fc8"]
    gains = {n: float("%.4g" % g) for n, g in gains.items()}
    s3 = measure(gains, a.frames, a.height, a.width)
    print("# verification pass (std per layer with the rounded table):", file=sys.stderr)
    for n in names:
        print("#   %-22s std %.4f (target %.2f)" % (n, s3[n], tgt(n)), file=sys.stderr)
    print("#   max |fc8| %.3f over %d rows" % (s3["_abs_fc8_max"], s3["_rows"]), file=sys.stderr)
    print("CALIBRATED_GAINS = {")
    line = "   "
    for n in names:
        item = ' "%s": %.4g,' % (n, gains[n])
        if len(line) + len(item) > 116:
            print(line)
            line = "   "
        line += item
    print(line)
    print("}")


if __name__ == "__main__":
    main()
