#!/usr/bin/env python
"""bench.py — PoseCNN single-frame inference throughput on MI355X (frames/s) + Hough-vote roofline.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches it under
torch.distributed.run (one rank per GPU, RCCL). A "step" is one pass of the hot path over one
batch of `--batch` (default 16) synthetic 640x480 frames PER GPU (weak scaling): VGG16 backbone +
label/vertex heads (PyTorch-ROCm, fp32) -> softmax/argmax -> Hough voting -> ROI pooling ->
fc6/7/8 + tanh -> (hard_label, average_distance_loss) -> all-gather of the fixed-size detection
buffer -> host NMS / pose assembly. Inputs are resident in HBM when the timed region starts.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from posecnn_amd import _lib, config, dist as pdist, fcn, synth  # noqa: E402
from posecnn_amd.networks import vgg16_convs  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 matrix peak
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def build_net(dev, input_format, seed=3):
    net = vgg16_convs(input_format, 22, 64, (1.0,), 1.0, -1.0, vertex_reg_2d=True, pose_reg=True,
                      trainable=False, is_train=False, device=dev, seed=seed, init="he")
    synth.init_planted_heads(net)
    return net


def make_inputs(dev, first, B, H, W, input_format, nbuf):
    """nbuf distinct synthetic batches, resident on the device."""
    bufs = []
    g = torch.Generator(device="cpu").manual_seed(1234 + first)
    K = config.DEMO_INTRINSICS.copy()
    K[:2] *= W / 640.0
    for i in range(nbuf):
        im = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8).float()
        data = (im - torch.from_numpy(config.PIXEL_MEANS)).contiguous().to(dev)  # BGR - PIXEL_MEANS (test.py:60)
        data_p = None
        if input_format == "RGBD":
            depth = torch.randint(0, 3000, (B, H, W, 1), generator=g).float()
            d = (torch.clamp(depth / 2000.0, 0, 1) * 255).expand(B, H, W, 3)
            data_p = (d - torch.from_numpy(config.PIXEL_MEANS)).contiguous().to(dev)
        planted_np, scenes = synth.make_planted_batch(first + i * B, B, H=H, W=W, K=K)
        planted = {k: torch.from_numpy(v).to(dev) for k, v in planted_np.items()}
        bufs.append((data, data_p, planted, scenes))
    return bufs, K


def cpu_baseline(K, H, W, input_format, net_gpu, max_seconds=25.0, max_frames=6):
    """The same pipeline on the host: PyTorch-CPU fp32 dense layers + the C oracle for the custom
    layers ("port": the TF1 reference cannot run here). Bounded sample, all host threads."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cpu_reference import run_cpu_pipeline, vgg16_convs_cpu
    net = vgg16_convs_cpu(input_format, 22, 64, (1.0,), 1.0, -1.0, vertex_reg_2d=True, pose_reg=True,
                          trainable=False, is_train=False, init="he")
    net.share_weights(net_gpu)
    pts = synth.make_model_points(22, config.NUM_MODEL_POINTS)
    g = torch.Generator(device="cpu").manual_seed(99)
    threads = torch.get_num_threads()
    done, t_total = 0, 0.0
    for i in range(max_frames + 1):
        im = torch.randint(0, 256, (1, H, W, 3), generator=g, dtype=torch.uint8).float()
        data = (im - torch.from_numpy(config.PIXEL_MEANS)).numpy()
        planted_np, _ = synth.make_planted_batch(5000 + i, 1, H=H, W=W, K=K)
        t0 = time.perf_counter()
        out = run_cpu_pipeline(net, data, K, config.LOV_EXTENTS, pts, config.LOV_SYMMETRY, planted=planted_np)
        dt = time.perf_counter() - t0
        if i == 0:
            continue  # first frame pages in libraries / warms the thread pool
        done += 1
        t_total += dt
        if t_total > max_seconds:
            break
    # the Hough layer alone on the last frame's label / vertex maps: the GPU-kernel semantics (the parity
    # target, OpenMP) and the reference's own CPU kernel semantics (H7: ray marching, what demo.sh runs
    # without a GPU; single-threaded like the original, a different algorithm — not a parity target)
    import oracle
    meta1 = config.make_meta_data(K)[None]
    lab, ver = out["label_2d"], out["vertex_pred"]
    t0 = time.perf_counter(); oracle.hough_voting(lab, ver, config.LOV_EXTENTS, meta1, None, 0, -1.0, 0.02, 10)
    hough_port_ms = 1000 * (time.perf_counter() - t0)
    t0 = time.perf_counter(); rows_h7 = oracle.hough_cpu_kernel(lab, ver, config.LOV_EXTENTS, meta1)
    hough_h7_ms = 1000 * (time.perf_counter() - t0)
    # what the frame rate would be with the reference's own (cheaper, different) CPU Hough kernel in place
    # of the GPU-kernel semantics: an estimate from the pieces measured above
    per_frame_ms = 1000.0 * t_total / done
    alt = 1000.0 / max(per_frame_ms - hough_port_ms + hough_h7_ms, 1e-3)
    return {"value": done / t_total, "unit": "frames/s", "cores": int(threads), "kind": "port",
            "value_with_reference_cpu_kernel_hough_estimate": alt,
            "hough_ms_per_frame": {"gpu_kernel_semantics_openmp": hough_port_ms,
                                   "reference_cpu_kernel_semantics_1_thread": hough_h7_ms,
                                   "reference_cpu_kernel_detections": int(rows_h7.shape[0])},
            "sample": "%d synthetic 640x480 frames, batch 1, same graph/weights: PyTorch-CPU fp32 (%d threads) "
                      "+ C oracle (OpenMP) for hough/roi_pool/softmax; %d detections on the last frame"
                      % (done, threads, out["final_rois"].shape[0]),
            "seconds": t_total}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="frames per GPU per step")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--input", default="COLOR", choices=["COLOR", "RGBD"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-losses", action="store_true")
    ap.add_argument("--nbuf", type=int, default=2)
    ap.add_argument("--prewarm-seconds", type=float, default=8.0, help="untimed engine / clock warm-up before the W warm-up steps")
    ap.add_argument("--no-dual-pool", action="store_true", help="A/B switch: conv4_3 -> pool4 as two passes")
    ap.add_argument("--blas", default="hipblas", choices=["default", "hipblas", "hipblaslt"],
                    help="library behind the fp32 GEMMs (Winograd planes, fc6-8); see tools/probe_bmm.py")
    a = ap.parse_args()

    rank, world, local = pdist.init_from_env()
    assert world == a.gpus or world == 1 and a.gpus == 1, "launch with torch.distributed.run --nproc-per-node %d" % a.gpus
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if a.blas != "default":
        torch.backends.cuda.preferred_blas_library(a.blas)
    torch.backends.cudnn.benchmark = True          # MIOpen find: pick the fastest fp32 conv kernels
    torch.backends.cuda.matmul.allow_tf32 = False  # fp32 like the reference; no reduced precision
    torch.backends.cudnn.allow_tf32 = False

    B, H, W = a.batch, a.height, a.width
    net = build_net(dev, a.input)
    if a.no_dual_pool:
        net.dual_pool = frozenset()
    bufs, K = make_inputs(dev, 100000 * rank, B, H, W, a.input, a.nbuf)
    pts = torch.from_numpy(synth.make_model_points(22, config.NUM_MODEL_POINTS)).to(dev)
    feed_cache = None
    last = {}

    drain = pdist.HostDrain(depth=2)

    def launch(i):
        """Enqueue one batch: backbone + heads + Hough voting + RoI/pose branch + all-gather + async D2H.
        No host synchronisation in here."""
        nonlocal feed_cache
        data, data_p, planted, _ = bufs[i % len(bufs)]
        if feed_cache is None:
            feed_cache = fcn._feed(net, data, data_p, K, config.LOV_EXTENTS, pts, config.LOV_SYMMETRY, 22, dev)
        det = fcn.im_segment_batch(net, data, K, config.LOV_EXTENTS, pts, config.LOV_SYMMETRY, data_p=data_p,
                                   planted=planted, feed_cache=feed_cache, with_losses=not a.no_losses)
        packed = pdist.all_gather_packed(det.rows, det.count, frame_offset=rank * B)
        last["det"] = det
        return drain.submit(packed)

    def finish(ticket):
        """Wait for THAT batch's detections and post-process them on the host (class-aware NMS, pose rows)."""
        flat = drain.collect(ticket)
        rois, poses = fcn.finalize_batch(flat, flat.shape[0])
        last["rois"] = rois
        return rois.shape[0]

    def run(first, n):
        """n batches, software-pipelined by one: batch i+1 is enqueued before batch i is collected, so
        the D2H latency and the host NMS of batch i overlap the kernels of batch i+1. Every batch is
        launched AND finished inside the call."""
        ndet, pending = 0, None
        for i in range(n):
            h0 = time.perf_counter()
            t = launch(first + i)
            last["host_launch_s"] = last.get("host_launch_s", 0.0) + time.perf_counter() - h0
            if pending is not None:
                ndet += finish(pending)
            pending = t
        if pending is not None:
            ndet += finish(pending)
        return ndet

    with torch.no_grad():
        # untimed engine warm-up before the W contract warm-up steps: MIOpen find, library handles,
        # allocator pools — and the GPU's power state: under sustained load the clocks keep rising for
        # several seconds (measured on fresh boxes: 962 frames/s with no pre-warm, 1023 with 2 s,
        # 1132 with 8 s — the steady state a throughput job runs in)
        t_pre = time.perf_counter()
        pre = 0
        while time.perf_counter() - t_pre < a.prewarm_seconds:
            run(pre, 4)
            torch.cuda.synchronize()
            pre += 4
        run(0, a.warmup)
        torch.cuda.synchronize()
        _lib.profile_enable(True)   # HIP events around every library kernel, on the launch stream
        net.conv_timing = []        # ... and around every MIOpen convolution of the trunk
        pdist.barrier()
        torch.cuda.synchronize()
        last["host_launch_s"] = 0.0
        t0 = time.perf_counter()
        ndet = run(a.warmup, a.steps)
        torch.cuda.synchronize()
        pdist.barrier()
        t1 = time.perf_counter()
        kern = _lib.profile_report()
        _lib.profile_enable(False)
        conv_ms = sum(e0.elapsed_time(e1) for _, _, _, e0, e1 in net.conv_timing)
        conv_flops = sum(f for _, f, _, _, _ in net.conv_timing)          # executed (Winograd layers: 16 GEMMs)
        conv_direct_flops = sum(f for _, _, f, _, _ in net.conv_timing)   # what a direct convolution would execute
        net.conv_timing = None
    elapsed = pdist.max_over_ranks(t1 - t0, dev)

    if rank != 0:
        pdist.shutdown()
        return
    frames = B * world * a.steps
    ms_per_step = 1000.0 * elapsed / a.steps
    # Hough-vote roofline: algorithmic bytes per launch = B frames x (4*H*W + 12*N_fg + 56*R)
    # (SURVEY.md §8d A_hough), over the live HIP-event duration of hv_vote_kernel.
    lab = last["det"].label_2d
    n_fg = int((lab > 0).sum().item())
    n_rows = int(last["det"].count.item())
    alg_bytes = 4 * H * W * B + 12 * n_fg + 56 * n_rows
    hv = kern.get("hv_vote_kernel", {"avg_us": float("nan"), "calls": 0})
    achieved = alg_bytes / (hv["avg_us"] * 1e-6) / 1e9 if hv["calls"] else float("nan")
    hough_us = sum(v["avg_us"] for k, v in kern.items() if k.startswith("hv_"))
    # HBM traffic of the kernel from PMC counters: collected offline in separate --pmc passes (they
    # cannot share a run with the timed region) at this same workload; see profiles/README.md
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_hough_pmc.json")))["hv_vote_kernel"]
        if B == 16 and H == 480 and W == 640:
            traffic = int((2.0 * pmc["FETCH_SIZE_KB"] + pmc["WRITE_SIZE_KB"]) * 1024)
    except Exception:
        pass
    # brute-force-equivalent pair predicates of the reference kernel: sum_c ceil(N_c/skip)*H*W
    per_class = torch.bincount((lab.reshape(B, -1).long() + 22 * torch.arange(B, device=dev).unsqueeze(1)).flatten(),
                               minlength=22 * B).reshape(B, 22)[:, 1:]
    pairs = float(((per_class + net.skip_pixels - 1) // net.skip_pixels * (per_class > 500)).sum().item()) * H * W
    # HBM-bound kernels of the library: algorithmic bytes per step / live event time per step
    act = lambda div, ch: 4.0 * B * (H // div) * (W // div) * ch
    towers = 2 if a.input == "RGBD" else 1
    hbm = {
        "conv3x3_c3_bias_relu_kernel": towers * (act(1, 3) + act(1, 64)),
        "hard_label_fwd_kernel": 4.0 * B * H * W * (2 + 22),
        "upscore_softmax_argmax_kernel": 4.0 * B * H * W * (22 + 1) + act(8, 22),
    }
    if net.winograd_tile == 4 and net.winograd_min_channels == 64:
        # F(4x4,3x3) transforms of conv1_2 ... conv5_3: the input transform reads X and writes 2.25 X,
        # the output transform reads 2.25 Y and writes Y (or Y/4 where the max-pool is fused)
        x_in = (act(1, 64) + act(2, 64) + act(2, 128) + act(4, 128) + 2 * act(4, 256) + act(8, 256) + 2 * act(8, 512)
                + 3 * act(16, 512))
        y_all = (act(1, 64) + 2 * act(2, 128) + 3 * act(4, 256) + 3 * act(8, 512) + 3 * act(16, 512))
        y_written = y_all - 0.75 * (act(1, 64) + act(2, 128) + act(4, 256))
        if net.fuse_first_conv_into_winograd and net.fused_first_conv:
            # conv1_1 runs inside conv1_2's input transform: reads the frame, writes V (2.25 x [H,W,64])
            x_in -= act(1, 64)
            hbm["conv3x3_c3_wino43_kernel"] = towers * (act(1, 3) + 2.25 * act(1, 64))
        fused_gemm = net.winograd_fused_gemm   # conv1_2 (+pool) and conv2_1 end in the fused MFMA kernel instead
        if fused_gemm:
            y_all -= act(1, 64) + act(2, 128)
            y_written -= 0.25 * act(1, 64) + act(2, 128)
        hbm["wino43_input_kernel"] = towers * 3.25 * x_in
        hbm["wino43_output_kernel"] = towers * (2.25 * y_all + y_written)

    def us(k):  # per step, all template instances of a kernel together
        t = sum(v["avg_us"] * v["calls"] for n, v in kern.items() if n == k or n.startswith(k + "<"))
        return t / a.steps if t else None
    others = []
    for k, byt in hbm.items():
        t = us(k)
        if t and byt:
            others.append({"kernel": k, "bound": "hbm", "achieved": byt / (t * 1e-6) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                           "frac": byt / (t * 1e-6) / 1e9 / HBM_PEAK_GBPS, "us_per_step": round(t, 1)})
    # the Winograd output transforms (library kernels) belong to the convolutions' time
    wino_out_ms = sum(v["avg_us"] * v["calls"] for k, v in kern.items() if k.startswith("wino") and "_output" in k) / 1e3   # incl. wino43_gemm_output_kernel
    conv_ms += wino_out_ms
    conv_tflops = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms else None
    t_fused = us("wino43_gemm_output_kernel")
    if t_fused and net.winograd_tile == 4 and net.winograd_min_channels == 64:
        tiles = lambda div: B * ((H // div + 3) // 4) * ((W // div + 3) // 4)
        fl = towers * 2.0 * 36 * 64 * (tiles(1) * 64 + tiles(2) * 128)     # conv1_2 and conv2_1, executed flops
        others.append({"kernel": "wino43_gemm_output_kernel", "bound": "mfma", "achieved": fl / (t_fused * 1e-6) / 1e12,
                       "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fl / (t_fused * 1e-6) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                       "us_per_step": round(t_fused, 1)})
    out = {
        "metric": "RGB-D frames/sec (640x480, 21 YCB classes)",
        "value": frames / elapsed, "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic (random RGB frames; random He-init VGG16; planted 1/8-res scene "
                                "so the heads emit 5 objects/frame — DESIGN.md §synthetic workload)",
        "config": {"workload": "configs[2]: batch=%d/GPU 640x480 full pipeline (vgg16_convs %s + hough_voting + "
                               "roi_pool + fc6-8 + hard_label + average_distance_loss + all-gather + NMS)" % (B, a.input),
                   "global_batch": B * world, "per_gpu_batch": B, "height": H, "width": W, "num_classes": 22,
                   "input_format": a.input, "parallelism": "dp%d (frames sharded, 1 all-gather of detections)" % world,
                   "detections_per_step": ndet / a.steps},
        "roofline": {"kernel": "hv_vote_kernel", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                     "traffic_note": "bytes/launch = (2*FETCH_SIZE + WRITE_SIZE) from separate rocprofv3 --pmc passes (profiles/r01_hough_pmc.json)",
                     "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_us": hv["avg_us"], "launches": hv["calls"],
                     "note": "Hough voting is VALU/LDS bound, not HBM bound (SURVEY.md §8d): compulsory traffic is ~2 MB/frame",
                     "hough_sequence_us": hough_us, "hough_GBps_whole_sequence": alg_bytes / (hough_us * 1e-6) / 1e9 if hough_us else None,
                     "pair_predicates_equiv_per_launch": pairs,
                     "pair_predicates_equiv_per_s": pairs / (hv["avg_us"] * 1e-6) if hv["calls"] else None},
        "roofline_other": others,
        # the roofline object above is the Hough kernel BASELINE.json asks for; by time per step the
        # largest hand-written kernel of the library is this one (its own entry is in roofline_other)
        "dominant_library_kernel": (max(others, key=lambda o: o["us_per_step"])["kernel"] if others else None),
        "backbone": {"what": "the fp32 convolutions of the VGG16 trunk + heads: 3x3 layers with >= %d input channels as Winograd F(%dx%d,3x3) "
                             "(gfx950 transform kernels + library fp32 batched GEMM on MFMA), the rest as MIOpen/CK direct convolutions; "
                             "achieved = EXECUTED flops / (transforms + GEMMs + direct convs) time" % (
                                 net.winograd_min_channels, net.winograd_tile, net.winograd_tile), "bound": "mfma",
                     "direct_conv_equivalent_TFLOPs": conv_direct_flops / (conv_ms * 1e-3) / 1e12 if conv_ms else None,
                     "achieved": conv_tflops, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": conv_tflops / FP32_MFMA_PEAK_TFLOPS if conv_tflops else None,
                     "ms_per_step": conv_ms / a.steps, "share_of_step": conv_ms / a.steps / ms_per_step},
        "prewarm_seconds": a.prewarm_seconds,
        "host_launch_ms_per_step": 1000.0 * last["host_launch_s"] / a.steps,
        "kernels_us": {k: round(v["avg_us"], 2) for k, v in sorted(kern.items())},
    }
    if world == 1 and not a.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(K, H, W, a.input, net)
            out["cpu_baseline"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        except Exception as e:  # the baseline is reported, never required for the GPU number
            out["cpu_baseline"] = {"error": repr(e)}
    print(json.dumps(out), flush=True)
    pdist.shutdown()


if __name__ == "__main__":
    main()
