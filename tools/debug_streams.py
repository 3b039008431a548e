#!/usr/bin/env python
"""Diagnostic (round 3): batches issued alternately on two HIP streams versus the same batches on one stream.
Prints every tensor of the pose branch that differs, with the rows affected.

History: on the round-2 kernels a few batches per hundred differed (fc7 output rows = the operand rows of single waves'
DMA instructions, once a Hough row) — with one network object per stream (`two_nets`) too, without split-K rarely, with a
3-deep ring in fc_rows too; never on one stream in this script, once in ~10 full test-suite runs on one stream. Cause: the
MFMA kernels (fc_rows_mfma_kernel, wino43_mfma_kernel) recycle their LDS operand ring as the epilogue's staging buffer
behind `__syncthreads()`, which is a bare s_barrier there — the operand DMAs are inline asm, invisible to the compiler's
s_waitcnt insertion — so another wave's parked prefetch could land on top of freshly staged output rows. With an explicit
`s_waitcnt vmcnt(0)` in front of that barrier: 0 mismatching tensors in 4 x 24 batches.
    python tools/debug_streams.py [two_nets]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from posecnn_amd import config, synth, fcn
from posecnn_amd.networks import vgg16_convs
from test_gpu_round2 import _rgbd_inputs
gpu = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu)
B, H, W = 2, 240, 320
net = vgg16_convs("RGBD", 22, 64, (1.0,), 1.0, -1.0, vertex_reg_2d=True, pose_reg=True, trainable=False, is_train=True, seed=3, init="he", with_losses=False, device=gpu)
synth.init_calibrated(net)
TWO = "two_nets" in sys.argv
if TWO:
    net2 = vgg16_convs("RGBD", 22, 64, (1.0,), 1.0, -1.0, vertex_reg_2d=True, pose_reg=True, trainable=False, is_train=True, seed=3, init="he", with_losses=False, device=gpu)
    synth.init_calibrated(net2)
nets = [net, net2] if TWO else [net, net]
SYNC_ALLOC = "fence" in sys.argv
K = config.DEMO_INTRINSICS.copy(); K[:2] *= W / 640.0
rng = np.random.default_rng(5)
pts = T(synth.make_model_points(22, 256))
batches = []
for i in range(4):
    data, data_p = _rgbd_inputs(rng, B, H, W)
    planted_np, scenes = synth.make_planted_batch(40 + i, B, H=H, W=W, K=K, n_obj=3)
    batches.append((T(data), T(data_p), {k: T(v) for k, v in planted_np.items()}, T(synth.make_gt_poses(scenes, K, seed=i))))
names = ("poses_tanh", "poses_target", "poses_weight", "poses_pred", "fc7", "fc6", "pool_score", "rois", "loss_pose")
def one(b, net=net):
    det = fcn.im_segment_batch(net, b[0], K, config.LOV_EXTENTS, pts, config.LOV_SYMMETRY, data_p=b[1], planted=b[2], with_losses=True, gt_poses=b[3])
    return [det.rows.clone(), det.count.clone()] + [net.get_output(n).clone() for n in names]
with torch.no_grad():
    serial = [one(b) for b in batches]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=gpu), torch.cuda.Stream(device=gpu)]
    got = []
    for rep in range(6):
        for i, b in enumerate(batches):
            with torch.cuda.stream(streams[i % 2]):
                got.append(one(b, nets[i % 2]))
    torch.cuda.synchronize()
bad = 0
for j, g in enumerate(got):
    w = serial[j % 4]
    live = 9 * int(w[1])   # rows at or past the device-side count of the capacity-sized buffers are not defined
    for nm, a, b in zip(("rows", "count") + names, g, w):
        if nm in ("poses_tanh", "poses_pred", "fc7", "fc6", "pool_score"):
            a, b = a[:live], b[:live]
        if not torch.equal(a, b):
            d = (a.float() - b.float()).abs()
            print("batch", j, nm, "differs: max", float(d.max()), "n", int((d > 0).sum()), "rows touched", sorted(set(torch.nonzero(d.reshape(d.shape[0], -1).sum(1) > 0).flatten().tolist()))[:12] if d.dim() > 1 else "")
            bad += 1
print("mismatching tensors:", bad)
