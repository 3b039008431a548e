"""GPU parity tests added in round 2 (VERDICT r1 "next" #1, #5c):
  * the registered zero gradients of Hough voting (H6) and hard_label (L3), through the C-ABI and
    through autograd;
  * per-image capacity decoupled from the batch size (`rois_per_image`): a batched call returns the
    rows B single-frame calls of the reference return;
  * capacity-sized buffers + device-side row count in roi_pool_add2 / average_distance_loss;
  * the RGB-D two-tower graph in training mode (BASELINE configs[2]) end to end against the CPU
    restatement of the same graph (PyTorch-CPU fp32 + C oracle).
Custom-kernel outputs are compared bit for bit; the dense layers within the tolerances of
tests/test_gpu_pipeline.py."""
import numpy as np
import pytest

import oracle
from posecnn_amd import config, synth
from test_gpu_hough import NAMES, compare, frames, run_gpu
from test_gpu_ops import N, T, adl_case, random_rois, same

pytestmark = pytest.mark.gpu
F = np.float32


# ---- H6 / L3 -----------------------------------------------------------------------------------
def test_hough_and_hard_label_gradient_entries_write_zeros(gpu):
    import torch
    from posecnn_amd import ops
    B, H, W, C = 2, 24, 40, 5
    label = torch.zeros((B, H, W), dtype=torch.int32, device=gpu)
    vertex = torch.randn((B, H, W, 3 * C), device=gpu)
    # poison the allocator's pool so that stale bytes would show
    junk = torch.full((B * H * W * 3 * C * 2,), float("nan"), device=gpu); del junk
    gl, gv = ops.hough_voting_grad(label, vertex)
    assert gl.shape == (B, H, W) and gv.shape == vertex.shape
    assert not gl.cpu().numpy().view(np.uint32).any() and not gv.cpu().numpy().view(np.uint32).any()   # +0.0 bits
    prob = torch.rand((B, H, W, C), device=gpu)
    junk = torch.full((B * H * W * C * 2,), float("nan"), device=gpu); del junk
    gp, gg = ops.hard_label_grad(prob, label)
    assert gp.shape == prob.shape and gg.shape == label.shape
    assert not gp.cpu().numpy().view(np.uint32).any() and not gg.cpu().numpy().view(np.uint32).any()


def test_hough_and_hard_label_inside_autograd(gpu):
    """HoughvotinggpuGrad / HardlabelGrad (hough_voting_gpu_op.cc:440-484, hard_label_op_gpu.cu.cc:55-85):
    gradients exist and are exactly zero."""
    import torch
    from posecnn_amd import ops
    label, vertex, meta, _ = frames(300, 1, H=120, W=160, C=8, n_obj=3)
    ext = config.LOV_EXTENTS[:8]
    v = T(gpu, vertex).requires_grad_(True)
    out = ops.hough_voting_gpu(T(gpu, label), v, T(gpu, ext), T(gpu, meta), None, 0, -1.0, 0.02, 10, label_threshold=100)
    assert out[0].shape[0] >= 2
    (out[0].sum() + out[1].sum()).backward()
    assert v.grad is not None and v.grad.shape == v.shape and float(v.grad.abs().max()) == 0.0
    p = torch.rand((1, 16, 16, 8), device=gpu).requires_grad_(True)
    g = torch.randint(-1, 8, (1, 16, 16), dtype=torch.int32, device=gpu)
    hl = ops.hard_label(p, g, 0.3)
    same(N(hl), oracle.hard_label(N(p), N(g), 0.3), "hard_label under autograd")
    (hl * 2).sum().backward()
    assert float(p.grad.abs().max()) == 0.0


# ---- capacity ----------------------------------------------------------------------------------
def test_rois_per_image_equals_single_frame_calls(gpu):
    """B = 16 under the reference's rule keeps 8 maxima per image — the 8 lowest
    class ids. With rois_per_image = C-1 the batched call must return exactly what 16 single-frame
    calls (capacity 128 each, lib/fcn/test.py:1867) return, and match the oracle bit for bit."""
    B = 16
    label, vertex, meta, fr = frames(40, B, H=240, W=320, n_obj=12)
    ext = config.LOV_EXTENTS
    strict = run_gpu(gpu, label, vertex, ext, meta, None, 0, -1.0, 0.02, 10, label_threshold=150)
    per_image_strict = np.bincount(strict[0][:int(strict[5][1]), 0].astype(int), minlength=B)
    assert per_image_strict.max() == 8          # the truncation exists ...
    got = run_gpu(gpu, label, vertex, ext, meta, None, 0, -1.0, 0.02, 10, label_threshold=150, rois_per_image=21)
    want = oracle.hough_voting(label, vertex, ext, meta, None, 0, -1.0, 0.02, 10, label_thr=150, padded=True, rois_per_image=21)
    compare(got, want)
    n = int(got[5][1])
    per_image = np.bincount(got[0][:n, 0].astype(int), minlength=B)
    assert per_image.max() > 8                  # ... and is lifted
    rows = []
    for b in range(B):
        one = run_gpu(gpu, label[b:b + 1], vertex[b:b + 1], ext, meta[b:b + 1], None, 0, -1.0, 0.02, 10, label_threshold=150)
        k = int(one[5][1])
        r = np.concatenate([one[0][:k], one[1][:k]], axis=1)
        r[:, 0] = b
        rows.append(r)
    rows = np.concatenate(rows)
    assert rows.shape[0] == n
    same(np.concatenate([got[0][:n], got[1][:n]], axis=1), rows, "batched == single-frame calls")


def test_rois_per_image_train_mode_and_capacity_check(gpu):
    import torch
    from posecnn_amd import ops
    B = 4
    label, vertex, meta, fr = frames(70, B, H=240, W=320, n_obj=6)
    ext = config.LOV_EXTENTS
    K = config.DEMO_INTRINSICS.copy(); K[:2] *= 0.5
    gt = synth.make_gt_poses([{"objects": f["objects"]} for f in fr], K, seed=1)
    got = run_gpu(gpu, label, vertex, ext, meta, gt, 1, -1.0, 0.02, 10, label_threshold=150, rois_per_image=21)
    want = oracle.hough_voting(label, vertex, ext, meta, gt, 1, -1.0, 0.02, 10, label_thr=150, padded=True, rois_per_image=21)
    assert got[0].shape[0] == B * 21 * 9
    compare(got, want)
    n = int(got[5][1])
    assert n % 9 == 0 and n >= 9 * 2 * B
    assert (got[3][:n].sum(axis=1) > 0).sum() >= n // 2     # most rows carry a pose target (IoU > 0.2 with the gt box)
    # too few output rows for the requested capacity -> InvalidArgument, nothing launched
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu)
    small = tuple(torch.empty((10,) + s, dtype=d, device=gpu) for s, d in (((7,), torch.float32), ((7,), torch.float32),
                  ((88,), torch.float32), ((88,), torch.float32), ((), torch.int32))) + (torch.empty(2, dtype=torch.int32, device=gpu),)
    with pytest.raises(ValueError, match="rows"):
        ops.hough_voting_gpu_padded(t(label), t(vertex), t(ext), t(meta), None, 0, -1.0, 0.02, 10, out=small, rois_per_image=21)


# ---- device-side row counts ---------------------------------------------------------------------
def test_roi_pool_add2_with_device_row_count(gpu):
    import torch
    from posecnn_amd import ops
    rng = np.random.default_rng(3)
    B, C = 2, 512
    a = rng.standard_normal((B, 15, 20, C)).astype(F)
    b = rng.standard_normal((B, 30, 40, C)).astype(F)
    R, cap = 9, 24
    rois = np.zeros((cap, 7), F)
    rois[:R] = random_rois(rng, R, B, 22, 320, 240)
    rois[R:] = random_rois(rng, cap - R, B, 22, 320, 240)    # garbage past the count must not matter
    cnt = torch.tensor([R], dtype=torch.int32, device=gpu)
    got = N(ops.roi_pool_add2(T(gpu, a), 1 / 16.0, T(gpu, b), 1 / 8.0, T(gpu, rois), num_rows=cnt))
    wa, _ = oracle.roi_pool(a, rois[:R], 7, 7, 1 / 16.0, 0)
    wb, _ = oracle.roi_pool(b, rois[:R], 7, 7, 1 / 8.0, 0)
    same(got[:R], wa + wb, "rows below the count")
    assert not got[R:].any()


@pytest.mark.parametrize("R,cap", [(7, 16), (1, 9), (0, 4)])
def test_average_distance_with_device_row_count(gpu, R, cap):
    """on capacity-sized buffers the loss must be normalised by the TRUE row count
    (average_distance_loss_op_gpu.cu.cc:190,203 divide by batch_size * num_points)."""
    import torch
    from posecnn_amd import ops
    rng = np.random.default_rng(23)
    C, P = 22, 700
    pred, tgt, wgt, pts, sym = adl_case(rng, cap, C, P)
    cnt = torch.tensor([R], dtype=torch.int32, device=gpu)
    loss, diff = ops.average_distance_loss(T(gpu, pred), T(gpu, tgt), T(gpu, wgt), T(gpu, pts), T(gpu, sym), 0.01, num_rows=cnt)
    wl, wd = oracle.average_distance(pred[:R], tgt[:R], wgt[:R], pts, sym, 0.01)
    same(N(loss), wl, "loss")
    if R:
        same(N(diff)[:R], wd, "bottom_diff")
    assert not N(diff)[R:].any()


# ---- BASELINE configs[2]: RGB-D, training-mode Hough, losses --------------------------------------
def _rgbd_inputs(rng, B, H, W):
    im = rng.integers(0, 256, (B, H, W, 3)).astype(F)
    depth = rng.integers(0, 3000, (B, H, W, 1)).astype(F)
    data = (im - config.PIXEL_MEANS).astype(F)
    data_p = (np.tile(np.clip(depth / 2000.0, 0, 1) * 255, (1, 1, 1, 3)) - config.PIXEL_MEANS).astype(F)   # test.py:70-74
    return data, data_p


def test_winograd_mfma_trunk_at_full_size_against_direct_convolutions(gpu, capsys):
    """The default trunk (F(4x4,3x3) on the fp32 matrix cores, both towers grouped) against the same
    network with every 3x3 layer as a direct library convolution, at the REAL configuration: 4 RGB-D
    frames of 480x640, 22 classes — 1.2 M label decisions, the Hough layer and the pose branch behind
    them. Both are f32 with different summation orders; labels may differ only at near-ties."""
    import torch
    from posecnn_amd import fcn
    from posecnn_amd.networks import vgg16_convs
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    B, H, W = 4, 480, 640
    net = vgg16_convs("RGBD", 22, 64, (1.0,), 1.0, -1.0, vertex_reg_2d=True, pose_reg=True, trainable=False,
                      is_train=False, seed=3, init="he", with_losses=False, device=gpu)
    synth.init_calibrated(net)
    K = config.DEMO_INTRINSICS.copy()
    rng = np.random.default_rng(21)
    data, data_p = _rgbd_inputs(rng, B, H, W)
    planted_np, _ = synth.make_planted_batch(77, B, H=H, W=W, K=K, n_obj=5)
    planted = {k: T(gpu, v) for k, v in planted_np.items()}
    pts = T(gpu, synth.make_model_points(22, 256))
    outs = {}
    for mode, minch in (("winograd_mfma", 64), ("direct", 0)):
        net.winograd_min_channels = minch
        with torch.no_grad():
            det = fcn.im_segment_batch(net, T(gpu, data), K, config.LOV_EXTENTS, pts, config.LOV_SYMMETRY,
                                       data_p=T(gpu, data_p), planted=planted)
            n = int(det.count.item())
            outs[mode] = (N(det.label_2d), N(net.get_output("prob_normalized")), N(det.rows[:n]),
                          N(net.get_output("conv5_3")), N(net.get_output("conv4_3_p")), N(net.get_output("fc8")[:n]))
    net.winograd_min_channels = 64
    a, b = outs["winograd_mfma"], outs["direct"]
    flips = int((a[0] != b[0]).sum())
    report = {"label_flips": flips, "of": a[0].size, "max_prob_diff": float(np.abs(a[1] - b[1]).max()),
              "conv5_3_rel_err": float(np.abs(a[3] - b[3]).max() / np.abs(b[3]).max()),
              "conv4_3_p_rel_err": float(np.abs(a[4] - b[4]).max() / np.abs(b[4]).max()), "detections": int(a[2].shape[0])}
    assert flips == 0, report            # calibrated network: bit-identical label maps (1.2 M decisions)
    assert report["max_prob_diff"] < 1e-3 and report["conv5_3_rel_err"] < 1e-4, report
    assert a[2].shape == b[2].shape and np.array_equal(a[2][:, :2], b[2][:, :2]) and a[2].shape[0] >= 3 * B, report
    report["max_box_diff_px"] = float(np.abs(a[2][:, 2:6] - b[2][:, 2:6]).max())
    report["max_quat_diff"] = float(np.abs(a[2][:, 7:11] - b[2][:, 7:11]).max())
    report["max_trans_diff"] = float(np.abs(a[2][:, 11:] - b[2][:, 11:]).max())
    report["max_trans_rel_diff"] = float((np.abs(a[2][:, 11:] - b[2][:, 11:]).max(1) / np.maximum(np.abs(b[2][:, 11:]).max(1), 1e-6)).max())
    report["fc8_absmax"] = float(np.abs(b[5]).max())
    report["fc8_abs_diff"] = float(np.abs(a[5] - b[5]).max())
    report["depth_max_m"] = float(b[2][:, 13].max())
    # calibrated network (synth.init_calibrated): north_star's tolerance holds literally between the two f32 trunks
    assert np.array_equal(a[2][:, 2:7], b[2][:, 2:7]), report                       # boxes and vote counts: identical
    assert report["max_trans_diff"] < 1e-4 and report["max_quat_diff"] < 1e-4 and report["depth_max_m"] < 3.0, report
    assert report["fc8_absmax"] < 6.0 and report["fc8_abs_diff"] < 1e-5 * max(report["fc8_absmax"], 1.0), report
    with capsys.disabled():
        print("\nfull-size winograd-MFMA vs direct:", report)


@pytest.mark.parametrize("B,fmt,train", [(2, "RGBD", True), (1, "COLOR", False)])
def test_hipgraph_replay_equals_the_eager_step(gpu, B, fmt, train):
    """pipeline.GraphedStep (bench.py --graph / --latency): one whole step — trunk, heads, Hough voting,
    RoI pooling, fc6-8 on device-counted rows, the loss layers — captured into a hipGraph and replayed
    on new frame contents must give the eager step's detections bit for bit. Batch 1 exercises the
    Cin-split trunk launches and split-K fc6 (a memset node and two extra kernels inside the graph)."""
    import torch
    from posecnn_amd import fcn, pipeline
    from posecnn_amd.networks import vgg16_convs
    H, W = 240, 320
    net = vgg16_convs(fmt, 22, 64, (1.0,), 1.0, -1.0, vertex_reg_2d=True, pose_reg=True, trainable=False,
                      is_train=train, seed=3, init="he", with_losses=False, device=gpu)
    synth.init_calibrated(net)
    K = config.DEMO_INTRINSICS.copy(); K[:2] *= W / 640.0
    rng = np.random.default_rng(8)
    pts = T(gpu, synth.make_model_points(22, 256))
    frames = []
    for i in range(3):
        data, data_p = _rgbd_inputs(rng, B, H, W)
        planted_np, scenes = synth.make_planted_batch(60 + i, B, H=H, W=W, K=K, n_obj=3)
        frames.append((T(gpu, data), T(gpu, data_p) if fmt == "RGBD" else None, {k: T(gpu, v) for k, v in planted_np.items()},
                       T(gpu, synth.make_gt_poses(scenes, K, seed=i)) if train else None))
    # static inputs of the graph: refilled before every replay
    sdata, sdata_p = frames[0][0].clone(), None if frames[0][1] is None else frames[0][1].clone()
    splant = {k: v.clone() for k, v in frames[0][2].items()}
    sgt = None if frames[0][3] is None else frames[0][3].clone()

    # the constant feeds (extents, meta data, model points ...) are uploaded once, outside the capture
    feed = fcn._feed(net, sdata, sdata_p, K, config.LOV_EXTENTS, pts, config.LOV_SYMMETRY, 22, gpu)

    def step():
        det = fcn.im_segment_batch(net, sdata, K, config.LOV_EXTENTS, pts, config.LOV_SYMMETRY, data_p=sdata_p,
                                   planted=splant, feed_cache=feed, with_losses=True, gt_poses=sgt)
        return det.rows, det.count, det.label_2d, net.get_output("loss_pose")

    def fill(f):
        sdata.copy_(f[0])
        if sdata_p is not None:
            sdata_p.copy_(f[1])
        for k in splant:
            splant[k].copy_(f[2][k])
        if sgt is not None:
            sgt.copy_(f[3])

    with torch.no_grad():
        eager = []
        for f in frames:
            fill(f)
            eager.append([t.clone() for t in step()])
        torch.cuda.synchronize()
        g = pipeline.GraphedStep(step, warmup=1, device=gpu)
        for j in (1, 2, 0, 1):
            fill(frames[j])
            out = [t.clone() for t in g.replay()]
            torch.cuda.synchronize()
            assert int(out[1]) == int(eager[j][1]) and int(out[1]) > 0
            for got, want, name in zip(out, eager[j], ("rows", "count", "label_2d", "loss_pose")):
                same(N(got).reshape(-1), N(want).reshape(-1), "%s of frame set %d" % (name, j))


def test_batches_on_alternating_streams_equal_the_serial_run(gpu):
    """bench.py --streams 2: consecutive batches go to different HIP streams so that one batch's trunk
    overlaps the other's heads / Hough / RoI tail. Everything a batch touches is stream-local (allocator
    pools, library workspaces keyed by stream), so the detections must equal the one-stream run bit for bit.

    Round 3: with 3 repetitions this test passed while the kernels under it had a race — the MFMA kernels recycled their
    LDS ring as the epilogue's staging buffer behind a bare s_barrier, without waiting for the other waves' in-flight
    operand DMAs (inline asm, invisible to the compiler's s_waitcnt insertion): a few batches per hundred came out with
    operand rows in place of fc7 output rows once a second stream competed for memory (tools/debug_streams.py). Fixed in
    csrc/fc_mfma.hip / csrc/wino_mfma.hip; the test now runs 8 rounds and compares the pose branch's tensors too."""
    import torch
    from posecnn_amd import fcn
    from posecnn_amd.networks import vgg16_convs
    B, H, W = 2, 240, 320
    net = vgg16_convs("RGBD", 22, 64, (1.0,), 1.0, -1.0, vertex_reg_2d=True, pose_reg=True, trainable=False,
                      is_train=True, seed=3, init="he", with_losses=False, device=gpu)
    synth.init_calibrated(net)
    K = config.DEMO_INTRINSICS.copy(); K[:2] *= W / 640.0
    rng = np.random.default_rng(5)
    pts = T(gpu, synth.make_model_points(22, 256))
    batches = []
    for i in range(4):
        data, data_p = _rgbd_inputs(rng, B, H, W)
        planted_np, scenes = synth.make_planted_batch(40 + i, B, H=H, W=W, K=K, n_obj=3)
        batches.append((T(gpu, data), T(gpu, data_p), {k: T(gpu, v) for k, v in planted_np.items()},
                        T(gpu, synth.make_gt_poses(scenes, K, seed=i))))

    def one(b):
        det = fcn.im_segment_batch(net, b[0], K, config.LOV_EXTENTS, pts, config.LOV_SYMMETRY, data_p=b[1], planted=b[2],
                                   with_losses=True, gt_poses=b[3])
        return det.rows.clone(), det.count.clone(), net.get_output("loss_pose").clone(), net.get_output("fc7").clone(), net.get_output("poses_tanh").clone()

    with torch.no_grad():
        serial = [one(b) for b in batches]
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(device=gpu), torch.cuda.Stream(device=gpu)]
        got = []
        for rep in range(8):                       # many rounds: the two streams really run concurrently
            for i, b in enumerate(batches):
                with torch.cuda.stream(streams[i % 2]):
                    got.append(one(b))
        torch.cuda.synchronize()
    for j, (rows, count, loss, fc7, ptanh) in enumerate(got):
        want = serial[j % len(batches)]
        assert int(count) == int(want[1]) and int(count) > 0
        same(N(rows), N(want[0]), "rows of batch %d" % j)
        same(N(loss).reshape(-1), N(want[2]).reshape(-1), "loss_pose of batch %d" % j)
        same(N(fc7), N(want[3]), "fc7 of batch %d" % j)
        same(N(ptanh), N(want[4]), "poses_tanh of batch %d" % j)


@pytest.mark.parametrize("train", [False, True])
def test_batch_pipeline_rgbd_matches_cpu_reference(gpu, train):
    """vgg16_convs.py:99-126: second tower `*_p` on the depth blob, 1024-channel concat in front of
    score_conv4 / score_conv5. train=True adds the Hough layer's training mode with gt poses, so
    hard_label and average_distance_loss run on real targets."""
    import torch
    from cpu_reference import run_cpu_pipeline, vgg16_convs_cpu
    from posecnn_amd import dist as pdist, fcn
    from posecnn_amd.networks import vgg16_convs
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    B, H, W = 2, 240, 320
    kw = dict(vertex_reg_2d=True, pose_reg=True, trainable=False, is_train=train, seed=3, init="he", with_losses=True)
    net = vgg16_convs("RGBD", 22, 64, (1.0,), 1.0, -1.0, device=gpu, **kw)
    synth.init_calibrated(net)
    cpu = vgg16_convs_cpu("RGBD", 22, 64, (1.0,), 1.0, -1.0, **kw)
    K = config.DEMO_INTRINSICS.copy(); K[:2] *= W / 640.0
    rng = np.random.default_rng(2)
    data, data_p = _rgbd_inputs(rng, B, H, W)
    planted_np, scenes = synth.make_planted_batch(17, B, H=H, W=W, K=K, n_obj=3)
    gt = synth.make_gt_poses(scenes, K, seed=3) if train else None
    pts = synth.make_model_points(22, 256)
    planted = {k: torch.from_numpy(v).to(gpu) for k, v in planted_np.items()}
    with torch.no_grad():
        det = fcn.im_segment_batch(net, T(gpu, data), K, config.LOV_EXTENTS, pts, config.LOV_SYMMETRY, data_p=T(gpu, data_p),
                                   planted=planted, with_losses=True, gt_poses=None if gt is None else T(gpu, gt))
        rows, counts = pdist.all_gather_detections(det.rows, det.count)
    flat = pdist.flatten_gathered(rows, counts)
    g_rois, g_poses = fcn.finalize_batch(flat, flat.shape[0])
    assert "conv5_3_p" in net.layers and net.vars["score_conv4/weights"].shape[1] == 1024
    cpu.share_weights(net)
    ref = run_cpu_pipeline(cpu, data, K, config.LOV_EXTENTS, pts, config.LOV_SYMMETRY, planted=planted_np, data_p=data_p, gt_poses=gt)

    # north_star, literally (VERDICT r3 "Next" #1; the network carries synth.init_calibrated's realistic scales):
    # label maps bit-exact, quaternions / translations within 1e-4 ABSOLUTE, against the CPU restatement of the same graph
    flips = int((det.label_2d.cpu().numpy() != ref["label_2d"]).sum())
    assert flips == 0, "%d label pixels differ from the CPU restatement" % flips
    want_cls = sorted((b, o[0]) for b, s in enumerate(scenes) for o in s["objects"] if (s["label_lowres"] == o[0]).sum() * 64 > 500)
    assert sorted((int(r[0]), int(r[1])) for r in g_rois) == want_cls
    assert g_rois.shape == ref["final_rois"].shape
    og = np.lexsort((g_rois[:, 1], g_rois[:, 0])); oc = np.lexsort((ref["final_rois"][:, 1], ref["final_rois"][:, 0]))
    gr, gp, cr, cp = g_rois[og], g_poses[og], ref["final_rois"][oc], ref["final_poses"][oc]
    assert np.array_equal(gr[:, :2], cr[:, :2])
    trans_d = float(np.abs(gp[:, 4:] - cp[:, 4:]).max())
    box_d = float(np.abs(gr[:, 2:6] - cr[:, 2:6]).max())
    quat_d = float(np.abs(gp[:, :4] - cp[:, :4]).max())
    print("\nrgbd pipeline vs cpu restatement [train=%s]: 0 label flips, box diff %.3g px, trans diff %.3g m (depths %.2f-%.2f m), quat diff %.3g"
          % (train, box_d, trans_d, float(cp[:, 6].min()), float(cp[:, 6].max()), quat_d))
    assert box_d < 1e-3 and np.array_equal(gr[:, 6], cr[:, 6])     # boxes and vote counts
    assert trans_d < 1e-4 and quat_d < 1e-4
    n = int(det.count.item()) * (9 if train else 1)
    w_gpu = net.get_output("poses_weight")[:n].cpu().numpy()
    loss_gpu = float(net.get_output("loss_pose"))
    if train:
        assert n == ref["rois"].shape[0] and n % 9 == 0
        assert np.array_equal(w_gpu, ref["poses_weight"])              # same rows matched a gt pose
        same(net.get_output("poses_target")[:n].cpu().numpy(), ref["poses_target"], "poses_target")
        assert (w_gpu.sum(axis=1) > 0).sum() >= n // 2
        assert loss_gpu > 0 and abs(loss_gpu - float(np.ravel(ref["loss_pose"])[0])) <= 1e-4 * max(1.0, abs(float(np.ravel(ref["loss_pose"])[0])))
    else:
        assert loss_gpu == 0.0 and not w_gpu.any()   # is_train = 0: no targets -> ADL skips every row
    hl = net.get_output("gt_label_weight").cpu().numpy()
    assert hl.shape == (B, H, W, 22) and set(np.unique(hl)) <= {0.0, 1.0}


# ---- fc6 / fc7 on capacity-sized rows --------------------------------------------------------------
@pytest.mark.parametrize("M,K,N,count,relu", [(200, 25088, 256, 77, True), (3024, 4096, 4096, 468, True), (64, 128, 64, 64, False),
                                             (130, 1024, 192, 0, True), (65, 640, 128, None, True), (1, 256, 64, 1, False),
                                             (336, 4096, 128, 70, True), (200, 25088, 256, 200, True), (600, 2048, 1024, 513, False)])
def test_fc_rows_matches_float64_and_skips_padding(gpu, M, K, N, count, relu):
    """`Network.fc` (network.py:392-422) as one fp32-MFMA kernel over a capacity-sized row buffer: rows below
    the device-side count equal x @ W + b (f32 roundoff against a float64 reference, same as the library
    GEMM), rows at or past it are exactly zero whatever the buffer held (NaN poison)."""
    import torch
    from posecnn_amd import ops
    torch.backends.cuda.matmul.allow_tf32 = False
    g = torch.Generator(device="cpu").manual_seed(M + K)
    x = torch.randn((M, K), generator=g).to(gpu)
    w = (torch.randn((K, N), generator=g) / K ** 0.5).to(gpu)
    b = torch.randn((N,), generator=g).to(gpu)
    n = M if count is None else count
    if n < M:
        x[n:] = float("nan")      # padding rows must never reach the result
    cnt = None if count is None else torch.tensor([count], dtype=torch.int32, device=gpu)
    y = ops.fc_rows(x, w.t().contiguous(), b, relu, num_rows=cnt)
    assert y.shape == (M, N)
    ref = x[:n].double() @ w.double() + b.double()
    if relu:
        ref = torch.relu(ref)
    lib = torch.addmm(b, x[:n], w)
    if relu:
        lib = torch.relu(lib)
    if n:
        scale = float(ref.abs().max())
        err = float((y[:n].double() - ref).abs().max())
        err_lib = float((lib.double() - ref).abs().max())
        assert err <= max(3.0 * err_lib, 4e-6 * scale), (err, err_lib, scale)   # f32 summation-order noise, K up to 25088
    assert not y[n:].cpu().numpy().view(np.uint32).any()
    # split-K (few live rows, decided on the device) sums its partial products in a fixed order: run to run identical
    y2 = ops.fc_rows(x, w.t().contiguous(), b, relu, num_rows=cnt)
    assert torch.equal(y, y2)


def test_fc_rows_addend_is_a_conv_over_a_concatenation(gpu):
    """conv(1,1) over concat(a, b) (the RGB-D heads, vgg16_convs.py:104-113) == fc_rows(a, Wa, bias,
    addend=fc_rows(b, Wb, 0)): checked against the float64 product and the framework's convolution of
    the materialised concatenation; and through the network: `head_gemm` on/off agree."""
    import torch
    import torch.nn.functional as F
    from posecnn_amd import ops
    g = torch.Generator(device="cpu").manual_seed(31)
    B, h, w, C, N = 2, 15, 20, 512, 64
    a = torch.randn((B, h, w, C), generator=g).to(gpu)
    b = torch.randn((B, h, w, C), generator=g).to(gpu)
    W = (torch.randn((N, 2 * C), generator=g) / (2 * C) ** 0.5).to(gpu)
    bias = torch.randn((N,), generator=g).to(gpu)
    part = ops.fc_rows(b.reshape(-1, C), W[:, C:].contiguous(), torch.zeros_like(bias), relu=False)
    y = ops.fc_rows(a.reshape(-1, C), W[:, :C].contiguous(), bias, relu=True, addend=part).view(B, h, w, N)
    cat = torch.cat([a, b], dim=3)
    ref = torch.relu(cat.double().reshape(-1, 2 * C) @ W.double().t() + bias.double()).view(B, h, w, N)
    lib = torch.relu(F.conv2d(cat.permute(0, 3, 1, 2), W.view(N, 2 * C, 1, 1), bias)).permute(0, 2, 3, 1)
    err, err_lib = float((y.double() - ref).abs().max()), float((lib.double() - ref).abs().max())
    assert err <= max(2.0 * err_lib, 2e-6 * float(ref.abs().max())), (err, err_lib)
    # the addend also rides through the split-K reduction (few rows, long K: partial products + fixed-order sum)
    M, K2, N2 = 100, 2048, 128
    x = torch.randn((M, K2), generator=g).to(gpu)
    W2 = (torch.randn((N2, K2), generator=g) / K2 ** 0.5).to(gpu)
    b2 = torch.randn((N2,), generator=g).to(gpu)
    add = torch.randn((M, N2), generator=g).to(gpu)
    cnt = torch.tensor([77], dtype=torch.int32, device=gpu)
    y2 = ops.fc_rows(x, W2, b2, relu=True, num_rows=cnt, addend=add)
    ref2 = torch.relu(x[:77].double() @ W2.double().t() + b2.double() + add[:77].double())
    assert float((y2[:77].double() - ref2).abs().max()) <= 4e-6 * float(ref2.abs().max()) + 1e-6
    assert not y2[77:].cpu().numpy().view(np.uint32).any()


def test_fc_layer_uses_the_row_count(gpu):
    """Network.fc routes to the MFMA kernel when `rows_count` is set and gives the library result on the
    live rows (fc6 -> fc7 chain on ROI-pooled shaped input)."""
    import torch
    from posecnn_amd.networks import Network

    class Tiny(Network):
        def setup(self):
            pass
    net = Tiny(device=gpu, trainable=False)
    x = torch.randn((40, 7, 7, 64), device=gpu)
    net.layers = {"in": x}
    with torch.no_grad():
        net.feed("in").fc(128, height=7, width=7, channel=64, name="fa").fc(64, num_in=128, name="fb")
        want = net.get_output("fb").clone()
        net.rows_count = torch.tensor([25], dtype=torch.int32, device=gpu)
        net.feed("in").fc(128, height=7, width=7, channel=64, name="fa").fc(64, num_in=128, name="fb")
        got = net.get_output("fb")
    assert float((got[:25] - want[:25]).abs().max()) <= 1e-5 * float(want.abs().max())
    assert float(got[25:].abs().max()) == 0.0


# ---- realistic geometry: the reference's demo depth frames and real YCB models (tests/golden/make_demo_fixtures.py) ----
def _demo_fixture():
    import os
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    fr = np.load(os.path.join(g, "demo_frames.npz"))
    mo = np.load(os.path.join(g, "lov_models.npz"))
    return fr["depth"], fr["label"], mo["points"], mo["extents"]


def _vertex_from_depth(label, depth, C=22):
    """What the vertex head is trained to emit (lib/gt_synthesize_layer/minibatch.py:583-594): per foreground pixel
    the unit direction to its object's centre and the log of its depth — here from the REAL depth image, with a
    deterministic per-pixel angular perturbation so that cones are not perfectly aligned."""
    H, W = label.shape
    v = np.zeros((H, W, 3 * C), F)
    yy, xx = np.mgrid[0:H, 0:W]
    wob = (((xx * 73856093) ^ (yy * 19349663)) % 1000 / 1000.0 - 0.5) * 0.12     # +-0.06 rad
    for c in np.unique(label):
        if c == 0:
            continue
        m = label == c
        cy, cx = yy[m].mean(), xx[m].mean()
        ang = np.arctan2(cy - yy[m], cx - xx[m]) + wob[m]
        v[m, 3 * c] = np.cos(ang)
        v[m, 3 * c + 1] = np.sin(ang)
        v[m, 3 * c + 2] = np.log(np.maximum(depth[m].astype(np.float64) / config.DEMO_FACTOR_DEPTH, 0.25))
    return v


@pytest.mark.parametrize("frame", [0, 1, 2, 3, 4])
def test_hough_on_demo_frame_geometry(gpu, frame):
    """Full-size parity (480x640, C = 22, skip 10, labelThreshold 500, the file's extents and the demo intrinsics) on
    masks and depths taken from the reference's own demo frames, both vote_threshold branches."""
    depth, label, points, ext = _demo_fixture()
    assert np.array_equal(ext, config.LOV_EXTENTS)
    lab = label[frame].astype(np.int32)[None]
    ver = _vertex_from_depth(lab[0], depth[frame])[None]
    meta = config.make_meta_data(config.DEMO_INTRINSICS)[None]
    for vote_thr, per_thr in ((-1.0, 0.02), (40.0, 0.0002)):
        want = oracle.hough_voting(lab, ver, ext, meta, None, 0, vote_thr, per_thr, 10, padded=True)
        got = run_gpu(gpu, lab, ver, ext, meta, None, 0, vote_thr, per_thr, 10)
        compare(got, want)
        assert int(got[5][1]) >= (6 if vote_thr < 0 else 1)


def test_average_distance_on_real_models(gpu):
    """average_distance_loss with the real 2620-point YCB models (data/LOV/models/*/points.xyz), including the two
    classes lov.py:38 marks symmetric (16 wood block, 21 foam brick: nearest-neighbour search over all 2620 points)."""
    from posecnn_amd import ops
    _, _, points, _ = _demo_fixture()
    assert points.shape == (22, 2620, 3)
    rng = np.random.default_rng(5)
    R, C = 9, 22
    pred = np.zeros((R, 4 * C), F); tgt = np.zeros((R, 4 * C), F); wgt = np.zeros((R, 4 * C), F)
    for n, c in enumerate((1, 16, 5, 21, 11, 16, 14, 21, 3)):
        pred[n, 4 * c:4 * c + 4] = np.tanh(rng.standard_normal(4)).astype(F)
        tgt[n, 4 * c:4 * c + 4] = synth.random_unit_quats(rng, 1)[0]
        wgt[n, 4 * c:4 * c + 4] = 1
    loss, diff = ops.average_distance_loss(T(gpu, pred), T(gpu, tgt), T(gpu, wgt), T(gpu, points), T(gpu, config.LOV_SYMMETRY), 0.01)
    wl, wd = oracle.average_distance(pred, tgt, wgt, points, config.LOV_SYMMETRY, 0.01)
    assert wl[0] > 0
    same(N(loss), wl, "loss")
    same(N(diff), wd, "bottom_diff")
