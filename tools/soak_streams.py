#!/usr/bin/env python
"""Soak of the mode the headline is measured in: batches of 16 RGB-D frames of 480 x 640, train-mode Hough,
fused conv1_1 -> conv1_2 -> pool1, issued round-robin on THREE HIP streams, against the same batches run serially on one
stream — rows, count, label_2d, loss_pose and the live rows of fc7 / poses_tanh / pool_score, bit for bit. Prints one line
per mismatching tensor and a summary; profiles/r05_streams_soak.txt is this script's output on the final build.
    python tools/soak_streams.py [--rounds 8] [--streams 3]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from posecnn_amd import config, fcn, synth  # noqa: E402
from posecnn_amd.networks import vgg16_convs  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=8)
ap.add_argument("--streams", type=int, default=3)
ap.add_argument("--batches", type=int, default=4)
a = ap.parse_args()
gpu = torch.device("cuda:0")
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(gpu)
B, H, W, C = 16, 480, 640, 22
net = vgg16_convs("RGBD", C, 64, (1.0,), 1.0, -1.0, vertex_reg_2d=True, pose_reg=True, trainable=False, is_train=True, seed=3, init="he",
                  with_losses=False, device=gpu)
synth.init_calibrated(net)
K = config.DEMO_INTRINSICS.copy()
pts = T(synth.make_model_points(C, config.NUM_MODEL_POINTS, extents=config.LOV_EXTENTS))
g = torch.Generator(device="cpu").manual_seed(123)
batches = []
for i in range(a.batches):
    im = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8).float()
    data = (im - torch.from_numpy(config.PIXEL_MEANS)).float().contiguous()
    depth = torch.randint(0, 3000, (B, H, W, 1), generator=g).float()
    d = (torch.clamp(depth / 2000.0, 0, 1) * 255).expand(B, H, W, 3)
    data_p = (d - torch.from_numpy(config.PIXEL_MEANS)).float().contiguous()
    planted_np, scenes = synth.make_planted_batch(900 + i * B, B, H=H, W=W, K=K, C=C, extents=config.LOV_EXTENTS)
    batches.append((data.to(gpu), data_p.to(gpu), {k: T(v) for k, v in planted_np.items()}, T(synth.make_gt_poses(scenes, K, seed=i))))
names = ("fc7", "poses_tanh", "pool_score", "poses_pred")


def one(b):
    det = fcn.im_segment_batch(net, b[0], K, config.LOV_EXTENTS, pts, config.LOV_SYMMETRY, data_p=b[1], planted=b[2], with_losses=True, gt_poses=b[3])
    return [det.rows.clone(), det.count.clone(), det.label_2d.clone(), net.get_output("loss_pose").clone()] + [net.get_output(n).clone() for n in names]


with torch.no_grad():
    serial = []
    for b in batches:
        serial.append(one(b))
        torch.cuda.synchronize()
    streams = [torch.cuda.current_stream(gpu)] + [torch.cuda.Stream(device=gpu) for _ in range(a.streams - 1)]
    bad, total, t0 = 0, 0, time.perf_counter()
    for rnd in range(a.rounds):
        got = []
        for j in range(6 * len(batches)):
            with torch.cuda.stream(streams[j % len(streams)]):
                got.append(one(batches[j % len(batches)]))
        torch.cuda.synchronize()
        for j, gt in enumerate(got):
            want = serial[j % len(batches)]
            live = 9 * int(want[1])
            for nm, x, y in zip(("rows", "count", "label_2d", "loss_pose") + names, gt, want):
                if nm in names:          # rows at or past the device-side count are not defined (never written / masked)
                    x, y = x[:live], y[:live]
                if not torch.equal(x, y):
                    dd = (x.float() - y.float()).abs()
                    print("round %d batch %d: %s differs (max %.3g, %d values)" % (rnd, j, nm, float(dd.max()), int((dd > 0).sum())))
                    bad += 1
            total += 1
    dt = time.perf_counter() - t0
print("soak: %d batches of %d RGB-D frames %dx%d on %d streams (fused conv12: %s), %d detections per batch (serial), %.1f s: %d mismatching tensors"
      % (total, B, W, H, len(streams), bool(net.fused_conv12), int(serial[0][1]), dt, bad))
sys.exit(1 if bad else 0)
