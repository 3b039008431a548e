#!/usr/bin/env python
"""bench.py — PoseCNN inference throughput on MI355X (frames/s) + Hough-vote roofline.

Contract: `python bench.py --gpus N --steps K --warmup W`. With N > 1 and no WORLD_SIZE in the
environment it re-executes itself under `python -m torch.distributed.run` (one rank per GPU, RCCL);
under a launcher it reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*. Rank 0 prints ONE JSON line.

A "step" is one pass of the hot path over one batch of `--batch` (default 16) synthetic frames PER
GPU (weak scaling). Default workload = BASELINE.json configs[2]: 640x480 RGB-D (two VGG16 towers),
21 YCB classes, full pipeline: image blobs in PINNED HOST memory -> H2D on a side stream (inside
the timed region, pipelined one batch ahead) -> backbone + label/vertex heads -> softmax/argmax ->
Hough voting (training mode: 9 rows per maximum + pose targets from planted ground-truth poses) ->
ROI pooling -> fc6/7/8 + tanh -> hard_label + average_distance_loss on those rows -> all-gather of
the fixed-size detection buffer -> D2H -> host NMS / pose assembly.

  --config linemod   BASELINE configs[4]: 960x1280, LINEMOD 13 objects (C = 14), batch 4
  --batch 1 --latency   BASELINE configs[1]-style per-frame latency (p50 / p99), COLOR or RGBD
  --dry-run          distributed plumbing only (gloo, CPU): used by tests/test_bench_launch.py
"""
import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 matrix peak
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_ACHIEVABLE_GBPS = 6290.0    # ... measured float4 copy


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="ycb", choices=["ycb", "linemod"])
    ap.add_argument("--batch", type=int, default=None, help="frames per GPU per step (ycb 16, linemod 4)")
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--input", default="RGBD", choices=["COLOR", "RGBD"])
    ap.add_argument("--losses", default="train", choices=["train", "test", "none"],
                    help="train: Hough is_train=1 with planted gt poses, hard_label + average_distance_loss on real "
                         "targets (configs[2]); test: is_train=0 graph with the loss layers evaluated (no targets -> "
                         "ADL skips every row); none: pure inference")
    ap.add_argument("--resident-inputs", action="store_true", help="A/B: frames already in HBM (no H2D in the timed region)")
    ap.add_argument("--raw-inputs", action="store_true",
                    help="upload the frames as the sensor delivers them (uint8 BGR, uint16 depth: 0.9 + 0.6 MB per frame instead of "
                         "2 x 3.7 MB of f32 blobs); the first trunk kernel forms the blobs of lib/fcn/test.py:56-74 itself, bit for bit")
    ap.add_argument("--latency", action="store_true", help="per-frame synchronous loop; reports p50/p99 latency")
    ap.add_argument("--graph", action="store_true", help="replay the step from a hipGraph (see posecnn_amd/pipeline.py)")
    ap.add_argument("--graph-upload", action="store_true", default=None,
                    help="with --graph: the H2D copy of the frame(s) is a node of the captured graph (pinned host buffer -> the graph's "
                         "device slot) instead of a side-stream copy fenced by events: one launch per frame, nothing to hand over "
                         "between streams. Default in --latency mode (a synchronous single-frame loop has nothing to overlap the "
                         "upload with); throughput runs keep the side-stream uploader, which copies batch i+1 under batch i's kernels")
    ap.add_argument("--streams", type=int, default=None,
                    help="HIP streams the batches alternate over (default 3 since round 4; 1 with --graph): batch i+1's trunk overlaps batch "
                         "i's heads / Hough / RoI tail, and the fused first-layers kernel (one workgroup per CU, matrix pipe half idle) "
                         "shares the chip with the other batches' kernels: 680.6 / 757.9 / 768.0 frames/s on 1 / 2 / 3 streams. Round 3 found and fixed the race this mode used to "
                         "expose (an s_waitcnt vmcnt(0) missing in front of the barrier that recycles the MFMA kernels' LDS ring; "
                         "tools/debug_streams.py). --graph --streams 2 is legal (each graph owns its scratch) but measured slower: "
                         "689 vs 722 frames/s")
    ap.add_argument("--backproject-grid", type=int, default=None,
                    help="also run the backprojecting layer on each batch's head features (G^3 voxels per frame); "
                         "default 128 for --config linemod (configs[4] names it), 0 = off otherwise")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fused-conv12", action="store_true",
                    help="A/B: conv1_1 + conv1_2's input transform and conv1_2 as the two round-3 kernels instead of the fused one")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the short configs[1] (batch-1 latency) and configs[4] (LINEMOD 1280x960 + backproject) runs whose "
                         "summaries the default single-GPU invocation appends under `secondary`")
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed regions of --steps steps run back to back: the FIRST is `value` / `ms_per_step` (the driver's contract, as in "
                         "every earlier round); all of them give value_median / value_min / value_max, so one run says whether a 2 %% "
                         "change is a change (VERDICT r5 #3: five boxes read 793-827 on one build)")
    ap.add_argument("--nbuf", type=int, default=2, help="distinct synthetic batches cycled through")
    ap.add_argument("--prewarm-seconds", type=float, default=8.0,
                    help="untimed sustained-load warm-up between the cold and the headline measurement")
    ap.add_argument("--blas", default="hipblas", choices=["default", "hipblas", "hipblaslt"],
                    help="library behind torch's own fp32 GEMMs — none is left on the default inference path; matters only "
                         "for the non-default graph switches (strict_numerics, fused_heads=False); see tools/probe_bmm.py")
    ap.add_argument("--force-process-group", action="store_true",
                    help="initialise RCCL and run the detection all-gather through it even at world size 1")
    ap.add_argument("--backend", default="auto", choices=["auto", "nccl", "gloo"],
                    help="process-group backend (auto: RCCL on GPUs). gloo + --shared-device is the one-GPU rehearsal of the "
                         "multi-rank path: real kernels, real shard offsets, real launcher, the detection block staged through the host")
    ap.add_argument("--shared-device", action="store_true",
                    help="every rank computes on cuda:0 (RCCL refuses two ranks on one device: use --backend gloo)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: launch / rendezvous / collective / JSON plumbing on gloo")
    ap.add_argument("--master-port", type=int, default=0)
    a = ap.parse_args(argv)
    # PCNN_BENCH_INPUTS = raw | resident | pinned: the frame hand-over of a line whose command is not ours to change (the driver's
    # `bench.py --gpus 8 --steps K --warmup W`): raw = --raw-inputs (24.6 MB per step and rank instead of 118), resident =
    # --resident-inputs. Unset: what the flags say (default: pinned f32 blobs, the same per-rank workload as the 1-GPU line)
    env_in = os.environ.get("PCNN_BENCH_INPUTS", "").strip().lower()
    if env_in == "raw":
        a.raw_inputs, a.resident_inputs = True, False
    elif env_in == "resident":
        a.resident_inputs, a.raw_inputs = True, False
    elif env_in == "pinned":
        a.resident_inputs = a.raw_inputs = False
    elif env_in:
        raise SystemExit("bench.py: PCNN_BENCH_INPUTS must be raw, resident or pinned (got %r)" % env_in)
    if a.streams is None:
        a.streams = 1 if a.graph else 3
    if a.graph_upload is None:
        a.graph_upload = bool(a.graph and a.latency)
    a.graph_upload = bool(a.graph_upload and a.graph and not a.resident_inputs)
    return a


def respawn_command(a, argv):
    """The command line `python bench.py --gpus N` turns itself into when it is not already running
    under a launcher: one rank per GPU on this node, rendezvous on 127.0.0.1."""
    port = a.master_port
    if not port:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def dry_run(a, pdist):
    """Everything around the GPU work, on gloo/CPU: rendezvous, per-rank shard bookkeeping, the
    all-gather of (fake) detections, host drain, barrier + MAX-over-ranks timing, rank-0 JSON."""
    import numpy as np
    import torch
    rank, world, _ = pdist.init_from_env(backend="gloo", force=a.force_process_group)
    B = a.batch or 16
    drain = pdist.HostDrain(depth=2)
    cap = 32
    rows = torch.zeros((cap, pdist.DET_COLS))
    rows[:3, 0] = torch.arange(3, dtype=torch.float32)   # local image index
    rows[:3, 1] = 1 + rank
    rows[:3, 6] = 1.0
    count = torch.tensor([3], dtype=torch.int32)
    seen, ranks_seen = 0, 0
    lanes = max(1, a.streams)   # --streams N: N batches in flight, their collectives issued in batch order from alternating lanes
    inflight = []               # (step, work handle | None, gathered buffer)

    def issue(step):
        r_ = rows.clone()
        r_[:3, 2] = float(step)                      # tag: which batch this block belongs to
        buf = pdist.pack_detections(r_, count, rank * B)
        if not torch.distributed.is_initialized():
            return step, None, buf.unsqueeze(0)
        flat = torch.empty((world * buf.shape[0], buf.shape[1]), dtype=buf.dtype)
        work = torch.distributed.all_gather_into_tensor(flat, buf, async_op=True)   # every rank issues these in the same order
        return step, work, flat.view(world, buf.shape[0], buf.shape[1])

    def retire(item):
        nonlocal seen, ranks_seen
        step, work, packed = item
        if work is not None:
            work.wait()
        flat = drain.collect(drain.submit(packed))
        assert flat.shape == (3 * world, pdist.DET_COLS)
        assert sorted(set(np.round(flat[:, 0]).astype(int))) == sorted(r * B + i for r in range(world) for i in range(3))
        assert set(np.round(flat[:, 2]).astype(int)) == {step}, "blocks of different batches were mixed"
        ranks_seen = max(ranks_seen, len(set((flat[:, 0] // B).astype(int).tolist())))
        seen += flat.shape[0]

    pdist.barrier()
    t0 = time.perf_counter()
    for step in range(a.steps):
        inflight.append(issue(step))
        if len(inflight) >= lanes:
            retire(inflight.pop(0))
    while inflight:
        retire(inflight.pop(0))
    local = time.perf_counter() - t0
    pdist.barrier()
    elapsed = pdist.max_over_ranks(time.perf_counter() - t0, torch.device("cpu"))
    rank_elapsed = pdist.gather_over_ranks(local, torch.device("cpu"))
    if rank == 0:
        print(json.dumps({"metric": "dry-run (no GPU work)", "value": B * world * a.steps / max(elapsed, 1e-9), "unit": "frames/s",
                          "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "dry_run": True,
                          **spread_fields(B * world * a.steps, [max(elapsed, 1e-9)], a.steps),
                          "per_rank": per_rank_fields(rank_elapsed, a.steps, 1 << 20),
                          "detections_gathered_per_step": seen / a.steps, "ranks_seen": ranks_seen, "batches_in_flight": lanes,
                          "process_group": bool(torch.distributed.is_initialized())}), flush=True)
    pdist.shutdown()


def spread_fields(frames, elapsed_all, steps):
    """value_median / value_min / value_max over the R timed regions (frames/s; the first region is `value`)."""
    vals = sorted(frames / e for e in elapsed_all)
    med = vals[len(vals) // 2] if len(vals) % 2 else 0.5 * (vals[len(vals) // 2 - 1] + vals[len(vals) // 2])
    return {"repeats": len(vals), "value_median": med, "value_min": vals[0], "value_max": vals[-1],
            "value_spread_rel": (vals[-1] - vals[0]) / med if med else None,
            "value_all": [round(frames / e, 2) for e in elapsed_all],
            "ms_per_step_all": [round(1000.0 * e / steps, 4) for e in elapsed_all],
            "repeats_note": "R timed regions of K steps back to back, each bracketed by barrier + synchronize, MAX over ranks; `value` / "
                            "`ms_per_step` are the FIRST region (the contract), the rest is its spread on this box"}


def per_rank_fields(rank_elapsed, steps, h2d_bytes):
    """Per-rank step times of the headline region (min / median / max over ranks, not only the MAX the contract asks for) and
    the host-to-device rate each rank sustained — so that a multi-GPU run explains itself."""
    ms = sorted(1000.0 * e / steps for e in rank_elapsed)
    med = ms[len(ms) // 2] if len(ms) % 2 else 0.5 * (ms[len(ms) // 2 - 1] + ms[len(ms) // 2])
    out = {"ranks": len(ms), "ms_per_step_min": ms[0], "ms_per_step_median": med, "ms_per_step_max": ms[-1],
           "ms_per_step_by_rank": [round(1000.0 * e / steps, 4) for e in rank_elapsed]}
    if h2d_bytes:
        out["h2d_GBps_per_rank"] = [round(h2d_bytes / (e / steps) / 1e9, 3) for e in rank_elapsed]
        out["h2d_note"] = ("bytes uploaded per step / this rank's own step time: the host-to-device rate every rank sustained at the same "
                           "time (8 ranks share the host's memory and PCIe root complexes)")
    return out


def make_host_inputs(first, B, H, W, C, input_format, nbuf, extents, K, train, raw=False):
    """nbuf distinct synthetic batches: image blobs as pinned host tensors (what feed_dict holds in
    lib/fcn/test.py:151-170), the planted 1/8-resolution scene and its gt poses as numpy."""
    import numpy as np
    import torch
    from posecnn_amd import config, pipeline, synth
    g = torch.Generator(device="cpu").manual_seed(1234 + first)
    host, aux = [], []
    for i in range(nbuf):
        im8 = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8)
        im = im8.float()
        data = (im - torch.from_numpy(config.PIXEL_MEANS)).float().contiguous()  # BGR - PIXEL_MEANS (test.py:60; float32 -= float64)
        data_p = None
        if input_format == "RGBD":
            depth_i = torch.randint(0, 3000, (B, H, W, 1), generator=g)
            depth = depth_i.float()
            d = (torch.clamp(depth / 2000.0, 0, 1) * 255).expand(B, H, W, 3)     # test.py:70-74
            data_p = (d - torch.from_numpy(config.PIXEL_MEANS)).float().contiguous()
        if raw:   # the same frames before _get_image_blob: what a camera driver hands over
            data = im8.contiguous()
            data_p = None if data_p is None else torch.from_numpy(depth_i.numpy().astype(np.uint16).reshape(B, H, W))
        planted_np, scenes = synth.make_planted_batch(first + i * B, B, H=H, W=W, K=K, C=C, extents=extents)
        gt = synth.make_gt_poses(scenes, K, seed=first + i) if train else None
        host.append((pipeline.pin(data), pipeline.pin(data_p)))
        aux.append((planted_np, gt, scenes))
    return host, aux


def cpu_baseline(a, K, H, W, C, extents, symmetry, net_gpu, train, max_seconds=25.0, max_frames=6):
    """The same graph on the host: PyTorch-CPU fp32 dense layers + the C oracle (OpenMP) for the
    custom layers ("port": the TF1 reference cannot run here). Bounded sample, all host threads."""
    import numpy as np
    import torch
    from posecnn_amd import config, synth
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cpu_reference import run_cpu_pipeline, vgg16_convs_cpu
    net = vgg16_convs_cpu(a.input, C, 64, (1.0,), 1.0, -1.0, vertex_reg_2d=True, pose_reg=True,
                          trainable=False, is_train=train, init="he", with_losses=a.losses != "none")
    net.share_weights(net_gpu)
    pts = synth.make_model_points(C, config.NUM_MODEL_POINTS, extents=extents)
    g = torch.Generator(device="cpu").manual_seed(99)
    threads = torch.get_num_threads()
    done, t_total = 0, 0.0
    out = None
    for i in range(max_frames + 1):
        im = torch.randint(0, 256, (1, H, W, 3), generator=g, dtype=torch.uint8).float()
        data = (im - torch.from_numpy(config.PIXEL_MEANS)).float().numpy()
        data_p = None
        if a.input == "RGBD":
            depth = torch.randint(0, 3000, (1, H, W, 1), generator=g).float()
            data_p = ((torch.clamp(depth / 2000.0, 0, 1) * 255).expand(1, H, W, 3) - torch.from_numpy(config.PIXEL_MEANS)).float().numpy()
        planted_np, scenes = synth.make_planted_batch(5000 + i, 1, H=H, W=W, K=K, C=C, extents=extents)
        gt = synth.make_gt_poses(scenes, K, seed=i) if train else None
        t0 = time.perf_counter()
        out = run_cpu_pipeline(net, data, K, extents, pts, symmetry, planted=planted_np, data_p=data_p, gt_poses=gt)
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
    t0 = time.perf_counter(); oracle.hough_voting(lab, ver, extents, meta1, None, 0, -1.0, 0.02, 10)
    hough_port_ms = 1000 * (time.perf_counter() - t0)
    t0 = time.perf_counter(); rows_h7 = oracle.hough_cpu_kernel(lab, ver, extents, meta1)
    hough_h7_ms = 1000 * (time.perf_counter() - t0)
    per_frame_ms = 1000.0 * t_total / done
    alt = 1000.0 / max(per_frame_ms - hough_port_ms + hough_h7_ms, 1e-3)
    return {"value": done / t_total, "unit": "frames/s", "cores": int(threads), "kind": "port",
            "value_with_reference_cpu_kernel_hough_estimate": alt,
            "hough_ms_per_frame": {"gpu_kernel_semantics_openmp": hough_port_ms,
                                   "reference_cpu_kernel_semantics_1_thread": hough_h7_ms,
                                   "reference_cpu_kernel_detections": int(rows_h7.shape[0])},
            "sample": "%d synthetic %dx%d %s frames, batch 1, same graph/weights/loss mode as the GPU run: PyTorch-CPU "
                      "fp32 (%d threads) + C oracle (OpenMP) for hough/roi_pool/softmax/hard_label/average_distance; "
                      "%d detections on the last frame" % (done, W, H, a.input, threads, out["final_rois"].shape[0]),
            "seconds": t_total}


def secondary_configs(timeout=150):
    """BASELINE configs[1] and configs[4] as short runs of this same script (fresh processes, after the headline
    measurement; ~20 s each), so that the driver's one command records them too (VERDICT r3 "Next" #6). Each entry is
    a summary of that run's own JSON line; a failed run is reported as such, never fatal for the headline."""
    import subprocess
    runs = {
        "configs[1]": ["--latency", "--batch", "1", "--input", "COLOR", "--losses", "none", "--graph", "--raw-inputs",
                       "--steps", "100", "--warmup", "5", "--prewarm-seconds", "2"],
        "configs[4]": ["--config", "linemod", "--steps", "8", "--warmup", "3", "--prewarm-seconds", "2"],
    }
    out = {}
    for name, flags in runs.items():
        cmd = [sys.executable, os.path.abspath(__file__)] + flags + ["--no-cpu-baseline", "--no-secondary"]
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            j = json.loads(line[-1]) if line else None
        except Exception as e:
            out[name] = {"error": repr(e), "flags": " ".join(flags)}
            continue
        if j is None:
            out[name] = {"error": "rc %d: %s" % (r.returncode, r.stderr[-400:]), "flags": " ".join(flags)}
            continue
        e = {"flags": " ".join(flags), "wall_s": round(time.perf_counter() - t0, 1), "workload": j["config"]["workload"],
             "frames_s": j["value"], "ms_per_step": j["ms_per_step"], "steps": j["steps"], "dtype": j["dtype"]}
        if "latency" in j:
            e.update({"p50_ms": j["latency"]["p50_ms"], "p99_ms": j["latency"]["p99_ms"], "min_ms": j["latency"]["min_ms"],
                      "latency_samples": j["latency"]["samples"], "frames_s_note": "pipelined throughput of the same loop; p50 / p99 are "
                      "synchronous per-frame times (upload -> kernels -> D2H -> host NMS)"})
        for o in j.get("roofline_other", []):
            if o["kernel"] == "backproject_fused_kernel":
                e["backproject"] = {k: o[k] for k in ("bound", "achieved", "peak", "unit", "frac", "us_per_step")}
        if j.get("roofline_dominant"):
            e["roofline_dominant"] = {k: j["roofline_dominant"][k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "us_per_step")}
        out[name] = e
    return out


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    a = parse_args(argv)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        cmd = respawn_command(a, argv)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.execv(cmd[0], cmd)

    from posecnn_amd import dist as pdist
    if a.dry_run:
        return dry_run(a, pdist)

    import numpy as np
    import torch
    from posecnn_amd import _lib, config, fcn, ops, pipeline, synth
    from posecnn_amd.networks import vgg16_convs

    if a.shared_device:
        assert a.backend == "gloo", "--shared-device needs --backend gloo (RCCL refuses duplicate devices)"
    rank, world, local = pdist.init_from_env(backend=None if a.backend == "auto" else a.backend, force=a.force_process_group)
    assert world == a.gpus or a.gpus == 1, "WORLD_SIZE=%d but --gpus %d" % (world, a.gpus)
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    if a.shared_device:
        local = 0
    torch.cuda.set_device(local)
    # one process per GPU: this rank's host thread (and, by first touch, the pinned buffers it allocates below) on its GPU's NUMA node
    numa = pdist.pin_to_gpu_numa(local) if world > 1 and not a.shared_device else None
    dev = torch.device("cuda", local)
    if a.blas != "default":
        torch.backends.cuda.preferred_blas_library(a.blas)
    torch.backends.cudnn.benchmark = True          # MIOpen find: pick the fastest fp32 conv kernels
    torch.backends.cuda.matmul.allow_tf32 = False  # fp32 like the reference; no reduced precision
    torch.backends.cudnn.allow_tf32 = False

    if a.config == "linemod":
        C, extents, symmetry = 14, config.LINEMOD_EXTENTS, config.LINEMOD_SYMMETRY
        H, W, B = a.height or 960, a.width or 1280, a.batch or 4
        cfg_name = "configs[4]"
    else:
        C, extents, symmetry = 22, config.LOV_EXTENTS, config.LOV_SYMMETRY
        H, W, B = a.height or 480, a.width or 640, a.batch or 16
        cfg_name = "configs[2]" if (B, a.input) == (16, "RGBD") else ("configs[1]" if B == 1 else "configs[2]-like")
    train = a.losses == "train"
    K = config.DEMO_INTRINSICS.copy()
    K[:2] *= W / 640.0   # same rule as lib/fcn/test.py:130-131
    G3 = a.backproject_grid if a.backproject_grid is not None else (128 if a.config == "linemod" else 0)

    net = vgg16_convs(a.input, C, 64, (1.0,), 1.0, -1.0, vertex_reg_2d=True, pose_reg=True,
                      trainable=False, is_train=train, device=dev, seed=3, init="he", with_losses=False)
    # (with_losses=False: the graph itself adds no loss layers — the log-softmax `prob` feeds only loss_cls,
    # which nobody fetches here; im_segment_batch evaluates hard_label and average_distance_loss on request)
    synth.init_calibrated(net)
    if a.no_fused_conv12:
        net.fused_conv12 = False
    host, aux = make_host_inputs(100000 * rank, B, H, W, C, a.input, a.nbuf, extents, K, train, raw=a.raw_inputs)
    planted = [{k: torch.from_numpy(v).to(dev) for k, v in p.items()} for p, _, _ in aux]
    gts = [None if g is None else torch.from_numpy(g).to(dev) for _, g, _ in aux]
    pts = torch.from_numpy(synth.make_model_points(C, config.NUM_MODEL_POINTS, extents=extents)).to(dev)
    bp = None
    if G3 > 0:
        # the backprojecting layer (lib/backprojecting_layer, Network.backproject network.py:224-226) on this
        # batch's own tensors: 64-channel head features at full resolution + the class probabilities, lifted into
        # a G^3 voxel grid with a synthetic depth map (BASELINE configs[4]: "stress HBM on backprojecting")
        ident = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]], np.float32)
        m3 = config.make_meta_data(K, voxel_step=(6.0 / G3, 6.0 / G3, 7.0 / G3), voxel_min=(-3, -3, -3),
                                   pose_world2live=ident, pose_live2world=ident).reshape(1, 1, 1, 48)
        gen = torch.Generator(device="cpu").manual_seed(99)
        bp = {"meta": torch.from_numpy(np.repeat(m3, B, axis=0)).to(dev),
              "depth": (1.5 + 0.5 * torch.rand((B, H, W, 1), generator=gen)).to(dev),
              "label_3d": torch.zeros((B, G3, G3, G3, C), device=dev)}
    if a.resident_inputs:
        resident = []
        for hb in host:
            slot = pipeline.alloc_adjacent(hb, dev)
            for d_, h_ in zip(slot, hb):
                if d_ is not None:
                    d_.copy_(h_)
            resident.append(slot)
        uploader = None
    else:
        # (--graph: one captured step per device slot, so slot k always holds host batch k: nbuf = slots = 2)
        uploader = None if a.graph_upload else pipeline.FrameUploader(host, dev, depth=1 if a.graph else 2)
    # --graph-upload: one device slot per distinct host batch; the copy into it is part of the captured step
    gslots = [pipeline.alloc_adjacent(hb, dev) for hb in host] if a.graph_upload else None
    h2d_bytes = sum(t.numel() * t.element_size() for t in host[0] if t is not None)
    if a.graph:
        assert len(host) == 2 and not a.resident_inputs, "--graph uses --nbuf 2 and pinned-host inputs"
    feed_cache = None
    last = {}
    drain = pdist.HostDrain(depth=2, spin=a.latency)
    seq = {"i": 0}
    ticket_batch = {}
    loss_log = torch.zeros(max(1, a.steps + 8), dtype=torch.float32, device=dev)

    graphs = {}

    def step(data, data_p, k):
        nonlocal feed_cache
        if feed_cache is None:
            feed_cache = fcn._feed(net, data, data_p, K, extents, pts, symmetry, C, dev)
        det = fcn.im_segment_batch(net, data, K, extents, pts, symmetry, data_p=data_p,
                                   planted=planted[k], feed_cache=feed_cache,
                                   with_losses=a.losses != "none", gt_poses=gts[k], frame_offset=rank * B)
        if bp is not None:
            feat = ops.deconv_bilinear(net.get_output("dropout"), int(16 * net.scale), int(8 * net.scale))   # `upscore` [B,H,W,64]
            top = ops.backproject(feat, net.get_output("prob_normalized"), bp["depth"], bp["meta"], bp["label_3d"], G3, 3, 0.02)
            last["backproject"] = top[0]
        return det

    # (--graph with --streams 2: every GraphedStep warms up and captures on a stream of its own, so the library scratch
    #  it bakes in — keyed by stream — is private to it and the two device slots' graphs may replay concurrently
    #  (tests/test_gpu_round3.py::test_two_graphs_replaying_concurrently_equal_the_eager_steps). It is not the default:
    #  two whole-step graphs interleaving kernel by kernel measured 689 frames/s against 722 for one graph stream.)
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(device=dev) for _ in range(max(1, a.streams) - 1)]
    multi = {"on": len(streams) > 1}

    def launch(_unused):
        """Enqueue one batch: (H2D wait) backbone + heads + Hough voting + RoI/pose branch + losses +
        all-gather + async D2H. No host synchronisation in here. With --streams > 1 consecutive batches
        go to different HIP streams (everything a batch allocates, reads and writes is stream-local:
        caching-allocator pools, library workspaces, upload slots and the drain are keyed by stream or
        fenced by events)."""
        first = seq["i"] == 0
        with torch.cuda.stream(streams[seq["i"] % len(streams)] if multi["on"] and not first else streams[0]):
            t = _launch()
        if first and len(streams) > 1:
            torch.cuda.synchronize()   # batch 0 fills the per-network caches (filter transforms, packed weights) every stream reads
        return t

    def _launch():
        i = seq["i"]
        seq["i"] += 1
        k = i % len(planted)
        if gslots is not None:
            data, data_p = gslots[k]
        else:
            data, data_p = uploader.get(i) if uploader is not None else resident[i % len(resident)]

        def upload(k=k):
            for d_, h_ in zip(gslots[k], host[k]):
                if d_ is not None:
                    d_.copy_(h_, non_blocking=True)
        if a.graph and last.get("graph_ready"):
            if k not in graphs:   # capture once per device slot (its tensors are the graph's static inputs)
                def fn(data=data, data_p=data_p, k=k):
                    if gslots is not None:
                        upload(k)     # a memcpy node: pinned host buffer k -> device slot k
                    d = step(data, data_p, k)
                    return d.rows, d.count, d.label_2d, net.get_output("poses_weight"), d.packed
                graphs[k] = pipeline.GraphedStep(fn, warmup=1, device=dev)
            rows, count, label_2d, pw, packed_ = graphs[k].replay()
            det = fcn.Detections(rows, count, label_2d, packed_)
            last["poses_weight"] = pw
        else:
            if gslots is not None:
                upload(k)
            det = step(data, data_p, k)
            last["poses_weight"] = net.layers.get("poses_weight")
        if uploader is not None:
            uploader.release(i)
        packed = pdist.all_gather_packed(det.rows, det.count, frame_offset=rank * B, packed=det.packed)
        last["det"] = det
        if last.get("record") is not None and "loss_pose" in net.layers and not (a.graph and last.get("graph_ready")):
            # (outputs_equal_serial: this batch's scalar pose loss — a function of all 9 x count rows of poses_tanh —
            # parked in a preallocated device slot on the batch's own stream; read back after the timed region)
            loss_log[i % loss_log.shape[0]].copy_(net.layers["loss_pose"].reshape(()))
        t = drain.submit(packed)
        ticket_batch[t] = i
        return t

    def finish(ticket):
        """Wait for THAT batch's detections and post-process them on the host (class-aware NMS, pose rows)."""
        flat = drain.collect(ticket)
        if last.get("record") is not None:
            last["record"].append((ticket_batch[ticket], flat))   # (collect() already copied out of the pinned slot)
        rois, poses = fcn.finalize_batch(flat, flat.shape[0])
        last["rois"] = rois
        if flat.shape[0]:   # detections carry GLOBAL frame indices (rank * B + local): whose frames reached this rank?
            last["ranks_seen"] = max(last.get("ranks_seen", 0), len(set((flat[:, 0] // B).astype(int).tolist())))
        return rois.shape[0]

    def run(n):
        """n batches, software-pipelined by one: batch i+1 is enqueued before batch i is collected, so
        the D2H latency and the host NMS of batch i overlap the kernels of batch i+1. Every batch is
        launched AND finished inside the call."""
        ndet, pending = 0, None
        for _ in range(n):
            h0 = time.perf_counter()
            t = launch(None)
            last["host_launch_s"] = last.get("host_launch_s", 0.0) + time.perf_counter() - h0
            if pending is not None:
                ndet += finish(pending)
            pending = t
        if pending is not None:
            ndet += finish(pending)
        return ndet

    def timed(n):
        pdist.barrier()
        torch.cuda.synchronize()
        last["host_launch_s"] = 0.0
        t0 = time.perf_counter()
        ndet = run(n)
        torch.cuda.synchronize()
        last["local_elapsed"] = time.perf_counter() - t0     # this rank's own K steps (before it waits for the others)
        pdist.barrier()
        return pdist.max_over_ranks(time.perf_counter() - t0, dev), ndet

    def check_equal_serial(first_timed):
        """Outside the timed region: the nbuf distinct batches once more, SERIALLY on one stream (one batch in flight,
        device idle in between), and every batch of the timed multi-stream / graph run held to them bit for bit —
        the gathered detection rows (boxes, scores, quaternions, translations: labels -> Hough -> RoI pooling -> fc6-8)
        and the scalar pose loss (every row of poses_tanh). VERDICT r4 #1: round 3's LDS-ring race lived in exactly
        the mode the headline is measured in; the reference loop has one frame in flight (lib/fcn/test.py:1867-1888)."""
        rec, last["record"] = last["record"], None
        torch.cuda.synchronize()
        timed_loss = loss_log.cpu().numpy().copy()
        was_multi, was_graph = multi["on"], last.get("graph_ready")
        multi["on"], last["graph_ready"] = False, False
        serial, serial_loss = {}, {}
        try:
            while len(serial) < len(planted):
                i = seq["i"]
                last["record"] = []
                finish(launch(None))
                torch.cuda.synchronize()
                serial.setdefault(i % len(planted), last["record"][0][1])
                serial_loss.setdefault(i % len(planted), float(loss_log[i % loss_log.shape[0]].item()))
        finally:
            multi["on"], last["graph_ready"], last["record"] = was_multi, was_graph, None
        bad = []
        for i, flat in rec:
            want = serial[i % len(planted)]
            if flat.shape != want.shape or flat.tobytes() != want.tobytes():
                bad.append("detections of timed batch %d" % (i - first_timed))
            elif a.losses != "none" and not a.graph and np.float32(timed_loss[i % loss_log.shape[0]]).tobytes() != np.float32(serial_loss[i % len(planted)]).tobytes():
                bad.append("loss_pose of timed batch %d" % (i - first_timed))
        ok = len(rec) == a.steps and not bad
        if not ok:
            print("bench.py: outputs_equal_serial FAILED: %d batches recorded, mismatches: %s" % (len(rec), bad[:8]), file=sys.stderr)
        return {"ok": ok, "batches_compared": len(rec), "mismatches": len(bad),
                "rows_per_batch": [int(serial[k].shape[0]) for k in sorted(serial)]}

    lat = None
    with torch.no_grad():
        # (1) the contract as written, on a cold process: W untimed warm-up steps (MIOpen find, library
        # handles, allocator pools), then K timed steps -> value_cold
        run(a.warmup)
        torch.cuda.synchronize()
        if a.graph:
            last["graph_ready"] = True      # everything is warm: from here on the step is a hipGraph replay
            run(2 * len(host))              # (captures one graph per device slot, untimed)
            torch.cuda.synchronize()
        elapsed_cold, _ = timed(a.steps)
        # (2) the steady state a throughput job runs in: under sustained load the GPU's clocks keep rising
        # for several seconds (measured round 1: 962 / 1023 / 1132 frames/s after 0 / 2 / 8 s). Untimed.
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < a.prewarm_seconds:
            run(4)
            torch.cuda.synchronize()
        sep_profile = a.graph or len(streams) > 1
        if not sep_profile:
            _lib.profile_enable(True)   # HIP events around every library kernel, on the launch stream
            net.conv_timing = []        # ... and around every remaining framework convolution / GEMM of the trunk
        last["record"] = []       # the timed run's own detections, as the host received them (7 KB per step, already copied)
        first_timed = seq["i"]
        elapsed, ndet = timed(a.steps)
        host_launch_ms = 1000.0 * last["host_launch_s"] / a.steps
        rank_elapsed = pdist.gather_over_ranks(last["local_elapsed"], dev)   # the headline region, rank by rank
        equal_serial = check_equal_serial(first_timed)
        if sep_profile:
            # per-kernel events cannot be recorded inside a graph replay, and with two streams a kernel's
            # event pair also spans whatever the other stream ran in between: time the same kernels once
            # more, eagerly on one stream
            last["graph_ready"] = False
            multi["on"] = False
            net.conv_timing = []
            _lib.profile_enable(True)
            run(a.steps)
            torch.cuda.synchronize()
            multi["on"] = len(streams) > 1
        kern = _lib.profile_report()
        _lib.profile_enable(False)
        if a.graph:
            last["graph_ready"] = True
        conv_ms = sum(e0.elapsed_time(e1) for _, _, _, e0, e1 in net.conv_timing)
        conv_flops = sum(f for _, f, _, _, _ in net.conv_timing)
        conv_direct_flops = sum(f for _, _, f, _, _ in net.conv_timing)
        net.conv_timing = None
        # (3) the same region R - 1 more times (same barrier + sync + MAX-over-ranks bracket, kernel timing off again): the
        # spread of the number, measured where the number is. `value` stays the FIRST region.
        elapsed_all = [elapsed]
        for _ in range(max(1, a.repeats) - 1):
            elapsed_all.append(timed(a.steps)[0])
        if a.latency:
            # per-frame latency: one batch at a time, upload -> kernels -> D2H -> host NMS, synchronously
            times = []
            for _ in range(max(a.steps, 50)):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                finish(launch(None))
                times.append(1000.0 * (time.perf_counter() - t0))
            times.sort()
            lat = {"batch": B, "p50_ms": times[len(times) // 2], "p99_ms": times[min(len(times) - 1, int(0.99 * len(times)))],
                   "min_ms": times[0], "samples": len(times)}

    ag_us = None
    if torch.distributed.is_initialized():   # every rank takes part (collective), before the other ranks leave
        try:
            ag_us = pdist.time_all_gather(last["det"].rows, last["det"].count)
        except Exception as e:
            ag_us = "failed: %r" % (e,)
    if rank != 0:
        pdist.shutdown()
        return
    frames = B * world * a.steps
    ms_per_step = 1000.0 * elapsed / a.steps
    det = last["det"]
    lab = det.label_2d
    # Hough-vote roofline: algorithmic bytes per launch = B frames x (4*H*W + 12*N_fg + 56*R)
    # (SURVEY.md §8d A_hough), over the live HIP-event duration of the vote kernel.
    n_fg = int((lab > 0).sum().item())
    n_rows = int(det.count.item()) * (9 if train else 1)
    alg_bytes = 4 * H * W * B + 12 * n_fg + 56 * n_rows
    vote_name = next((k for k in sorted(kern) if k.startswith("hv_vote")), "hv_vote_kernel")
    hv = kern.get(vote_name, {"avg_us": float("nan"), "calls": 0})
    achieved = alg_bytes / (hv["avg_us"] * 1e-6) / 1e9 if hv["calls"] else float("nan")
    hough_us = sum(v["avg_us"] for k, v in kern.items() if k.startswith("hv_"))
    # PMC counters: collected offline in separate rocprofv3 --pmc passes (they cannot share a run with the timed region) at this
    # same workload (tools/collect_pmc_step.sh -> profiles/r06_step_pmc.json); every kernel's entry is reported only while
    # the source file the kernel lives in still hashes to what the counters were collected on
    import hashlib
    pmc, pmc_src = {}, None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r06_step_pmc.json")))
        pmc_src = "profiles/r06_step_pmc.json"
    except Exception:
        pass
    sha_now = {}

    def pmc_of(kernel):
        """The counters of `kernel` if they belong to this build and this workload, else (None, why)."""
        ent = pmc.get(kernel)
        if not ent or (B, H, W, a.input) != (16, 480, 640, "RGBD"):
            return None, "no counters for this kernel / workload"
        f = ent.get("_src")
        if f not in sha_now:
            sha_now[f] = hashlib.sha256(open(os.path.join(ROOT, "posecnn_amd", "csrc", f), "rb").read()).hexdigest()[:16]
        if pmc.get("_source_sha16", {}).get(f) != sha_now[f]:
            print("bench.py: %s was collected on %s %s, this build is %s -> its counters are not reported; re-collect with "
                  "tools/collect_pmc_step.sh" % (pmc_src, f, pmc.get("_source_sha16", {}).get(f), sha_now[f]), file=sys.stderr)
            return None, "STALE: %s belongs to another build of %s" % (pmc_src, f)
        return ent, pmc_src

    clock_hz = 1e3 * torch.cuda.get_device_properties(dev).clock_rate if hasattr(torch.cuda.get_device_properties(dev), "clock_rate") else 2.4e9
    simds = 4 * torch.cuda.get_device_properties(dev).multi_processor_count
    ent, traffic_src = pmc_of(vote_name)
    traffic = int((2.0 * ent["FETCH_SIZE_KB"] + ent["WRITE_SIZE_KB"]) * 1024) if ent and "FETCH_SIZE_KB" in ent else None

    def issue_fracs(ent, avg_us):
        """What the kernel's REAL bound looks like: share of the chip's vector-ALU issue slots it used — a wave64
        VALU instruction occupies its SIMD-32 for 2 cycles (MI355X_MICROARCH.md) —, and the instruction mix."""
        if not ent or not avg_us or "SQ_INSTS_VALU" not in ent:
            return {}
        cyc = simds * clock_hz * avg_us * 1e-6
        valu = ent["SQ_INSTS_VALU"] - ent.get("SQ_INSTS_MFMA", 0)     # (SQ_INSTS_VALU counts the MFMAs too)
        tot = float(ent["SQ_INSTS_VALU"] + ent.get("SQ_INSTS_SALU", 0) + ent.get("SQ_INSTS_LDS", 0))
        o = {"valu_frac": 2.0 * valu / cyc, "valu_insts_per_launch": valu,
             "lds_inst_share": ent.get("SQ_INSTS_LDS", 0) / tot if tot else None,
             "salu_inst_share": ent.get("SQ_INSTS_SALU", 0) / tot if tot else None,
             "valu_frac_note": "(SQ_INSTS_VALU - SQ_INSTS_MFMA) x 2 issue cycles / (%d SIMDs x %.2f GHz x the live launch duration)" % (simds, clock_hz / 1e9)}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in ent and ent.get("SQ_INSTS_MFMA"):
            o["mfma_busy_share"] = ent["SQ_VALU_MFMA_BUSY_CYCLES"] / cyc
            o["valu_per_mfma"] = valu / float(ent["SQ_INSTS_MFMA"])
            o["mfma_busy_note"] = ("SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x clock x duration): the matrix pipe's busy share; every other vector "
                                   "instruction of either wave on the SIMD is paid in matrix time (SQ_VALU_MFMA_COEXEC_CYCLES = 0 on gfx950, DESIGN §3.2c)")
        if "SQ_WAIT_ANY" in ent and ent.get("SQ_WAVE_CYCLES"):
            o["wave_cycles_waiting"] = ent["SQ_WAIT_ANY"] / float(ent["SQ_WAVE_CYCLES"])
        if "SQ_ACTIVE_INST_VALU" in ent:
            # round 6: the vector ALUs' busy share as the counter itself has it. SQ_ACTIVE_INST_VALU counts, like SQ_WAVE_CYCLES, in
            # units of 4 cycles (MI355X_MICROARCH.md); summed over the SIMDs and divided by SIMDs x clock x duration it is the share
            # of the launch with a vector instruction executing. For hv_vote it reads ~1.0 where valu_frac (2 issue cycles per
            # instruction) reads 0.5: a dependent stream holds the pipe ~4 cycles per instruction (profiles/r05_valu_rate_probe.txt).
            o["valu_active_share"] = 4.0 * ent["SQ_ACTIVE_INST_VALU"] / cyc
        return o
    adl_rows = int((last["poses_weight"].sum(dim=1) > 0).sum().item()) if a.losses != "none" and last.get("poses_weight") is not None else 0

    # HBM-bound kernels of the library: algorithmic bytes per step / live event time per step
    act = lambda div, ch: 4.0 * B * (H // div) * (W // div) * ch
    towers = 2 if a.input == "RGBD" else 1
    hbm = {
        "hard_label_fwd_kernel": 4.0 * B * H * W * (2 + C),
        "upscore_softmax_argmax": 4.0 * B * H * W * (C + 1) + act(8, C),     # (generic kernel or the compile-time-C instance)
    }
    if a.losses != "none" and "hard_label_fwd_kernel" not in kern:
        # the Hardlabel op rides in the label head's launch (pcnn_upscore_softmax_argmax_hard_fwd): its label map in,
        # its C-wide weights out, in the same kernel's bytes
        hbm["upscore_softmax_argmax"] += 4.0 * B * H * W * (1 + C)
    hbm.update(net.hbm_table(B, H, W)) if hasattr(net, "hbm_table") else None
    if G3 > 0:   # writes data + flag [G^3, 64] and label [G^3, C], reads label_3d [G^3, C]
        hbm["backproject_fused_kernel"] = 4.0 * B * G3 ** 3 * (2 * 64 + 2 * C)

    def us(k):  # per step, all template instances of a kernel together
        t = sum(v["avg_us"] * v["calls"] for n, v in kern.items() if n == k or n.startswith(k + "<") or n.startswith(k + "_kernel")
                or n.startswith(k + "_fixed_kernel"))
        return t / a.steps if t else None
    others = []
    for k, byt in hbm.items():
        t = us(k)
        if t and byt:
            others.append({"kernel": k, "bound": "hbm", "achieved": byt / (t * 1e-6) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                           "frac": byt / (t * 1e-6) / 1e9 / HBM_PEAK_GBPS, "us_per_step": round(t, 1)})
    # MFMA kernels of the library (Winograd-domain GEMMs with fused output transform): executed flops / event time
    mfma_flops = net.mfma_table(B, H, W) if hasattr(net, "mfma_table") else {}
    lib_conv_ms, lib_conv_flops = 0.0, 0.0
    for k, fl in mfma_flops.items():
        t = us(k)
        if t and fl:
            others.append({"kernel": k, "bound": "mfma", "achieved": fl / (t * 1e-6) / 1e12, "peak": FP32_MFMA_PEAK_TFLOPS,
                           "unit": "TFLOP/s", "frac": fl / (t * 1e-6) / 1e12 / FP32_MFMA_PEAK_TFLOPS, "us_per_step": round(t, 1)})
            lib_conv_ms += t / 1e3
            lib_conv_flops += fl
    # fc6 / fc7 on the live rows + the head 1x1 products (all fc_rows_mfma_kernel): executed flops of the rows that exist
    live_rows = int(last["det"].count.item()) * (9 if train else 1)
    head_rows = B * (H // 8) * (W // 8)
    fc_flops = 2.0 * live_rows * (7 * 7 * 512 * 4096 + 4096 * 4096) \
        + 2.0 * head_rows * 512 * (64 * towers + 128) * 1.25      # conv4_3 heads + conv5_3 heads (a quarter of the pixels)
    t_fc = us("fc_rows_mfma_kernel")
    if t_fc:
        others.append({"kernel": "fc_rows_mfma_kernel", "bound": "mfma", "achieved": fc_flops / (t_fc * 1e-6) / 1e12, "peak": FP32_MFMA_PEAK_TFLOPS,
                       "unit": "TFLOP/s", "frac": fc_flops / (t_fc * 1e-6) / 1e12 / FP32_MFMA_PEAK_TFLOPS, "us_per_step": round(t_fc, 1),
                       "note": "fc6 + fc7 on %d live rows and the 1x1 head products" % live_rows})
    # the trunk as a whole: every kernel whose name says it belongs to a convolution
    trunk_ms = sum(v["avg_us"] * v["calls"] for k, v in kern.items() if k.startswith(("wino", "conv3x3", "conv12", "bias_"))) / 1e3 / a.steps \
        + conv_ms / a.steps
    direct_flops_step = towers * 2.0 * B * (H * W) * 9 * (64 * 3 + 64 * 64 + (128 * 64 + 128 * 128) / 4 + (256 * 128 + 2 * 256 * 256) / 16
                                                         + (512 * 256 + 2 * 512 * 512) / 64 + 3 * 512 * 512 / 256)
    name_of = "RGB-D" if a.input == "RGBD" else "RGB"
    workload = ("%s: batch=%d/GPU %dx%d %s, %d classes; vgg16_convs (%d tower%s) + hough_voting(is_train=%d) + roi_pool + fc6-8"
                % (cfg_name, B, W, H, name_of, C - 1, towers, "s" if towers > 1 else "", 1 if train else 0))
    if a.losses != "none":
        workload += " + hard_label + average_distance_loss (%d rows with pose targets)" % adl_rows
    if G3 > 0:
        workload += " + upscore deconv + backproject into a %d^3 grid (%.1f GB of voxel features per step)" % (G3, 4.0 * B * G3 ** 3 * (2 * 64 + C) / 1e9)
    workload += " + all-gather + D2H + NMS; inputs from %s" % ("HBM (resident)" if a.resident_inputs else ("pinned host memory as raw uint8 / uint16 frames (H2D inside the timed region; the blobs are formed in the first kernel)" if a.raw_inputs else "pinned host memory (H2D inside the timed region)"))
    out = {
        "metric": "%s frames/sec (%dx%d, %d classes)" % (name_of, W, H, C - 1),
        "value": frames / elapsed, "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic (random frames; seeded VGG16 weights calibrated to O(1) activations — synth.init_calibrated, the "
                                "weights the parity tests assert north_star's tolerance on; planted 1/8-res scene so the heads emit "
                                "5 objects/frame with known poses — DESIGN.md §5)",
        "config": {"workload": workload, "global_batch": B * world, "per_gpu_batch": B, "height": H, "width": W,
                   "num_classes": C, "input_format": a.input, "losses": a.losses,
                   "inputs": "resident" if a.resident_inputs else ("pinned-host-raw" if a.raw_inputs else "pinned-host"),
                   "h2d_MB_per_step": None if a.resident_inputs else round(h2d_bytes / 1e6, 1),
                   "parallelism": "dp%d (frames sharded, 1 all-gather of detections)" % world,
                   "detections_per_step": ndet / a.steps, "adl_rows_with_targets": adl_rows},
        **spread_fields(frames, elapsed_all, a.steps),
        "value_cold": frames / elapsed_cold,
        "value_note": "value: after %g s of untimed sustained-load pre-warm (GPU clocks ramp for seconds); value_cold: the same K "
                      "steps right after the W warm-up steps of a fresh process" % a.prewarm_seconds,
        "roofline": {"kernel": vote_name, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                     "traffic_note": "bytes/launch = (2*FETCH_SIZE + WRITE_SIZE) from separate rocprofv3 --pmc passes (%s)" % traffic_src,
                     "real_bound": "vector-ALU issue + latency (valu_frac, lds_inst_share below): the HBM fraction is what north_star asks for, not what limits the kernel",
                     "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_us": hv["avg_us"], "launches": hv["calls"],
                     "note": "Hough voting is VALU/LDS bound, not HBM bound (SURVEY.md §8d): compulsory traffic is ~2 MB/frame",
                     "hough_sequence_us": hough_us, "hough_GBps_whole_sequence": alg_bytes / (hough_us * 1e-6) / 1e9 if hough_us else None,
                     "frac_whole_sequence": alg_bytes / (hough_us * 1e-6) / 1e9 / HBM_PEAK_GBPS if hough_us else None,
                     "sequence_note": "SURVEY.md §8d defines the Hough-vote rate over the whole launch sequence (hist + scatter + vote + select + emit); "
                                      "`frac` above is the vote kernel alone",
                     **issue_fracs(ent, hv["avg_us"] if hv["calls"] else None)},
        # the kernel the step actually spends its time in, against ITS roof (the headline `roofline` key is the Hough vote
        # kernel north_star names — 0.8 % of the step)
        "roofline_dominant": (dict(max(others, key=lambda o: o["us_per_step"]), share_of_step=max(o["us_per_step"] for o in others) / 1e3 / ms_per_step,
                                   **issue_fracs(pmc_of(max(others, key=lambda o: o["us_per_step"])["kernel"])[0],
                                                 (lambda k_: (us(k_) or 0) * a.steps / max(1, sum(v["calls"] for n, v in kern.items() if n == k_ or n.startswith(k_ + "<"))))
                                                 (max(others, key=lambda o: o["us_per_step"])["kernel"])))
                              if others else None),
        "roofline_other": others,
        "dominant_library_kernel": (max(others, key=lambda o: o["us_per_step"])["kernel"] if others else None),
        "backbone": {"what": "all convolution kernels of the VGG16 trunk(s): gfx950 Winograd F(4x4,3x3) transform + fp32-MFMA kernels "
                             "of libposecnn_hip.so plus whatever still goes to the framework (conv_timing)",
                     "bound": "mfma", "ms_per_step": trunk_ms, "share_of_step": trunk_ms / ms_per_step,
                     "library_mfma_TFLOPs": lib_conv_flops / (lib_conv_ms * 1e-3) / 1e12 if lib_conv_ms else None,
                     "framework_conv_ms_per_step": conv_ms / a.steps,
                     "framework_conv_TFLOPs": conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms else None,
                     "direct_conv_equivalent_TFLOPs": direct_flops_step / (trunk_ms * 1e-3) / 1e12 if trunk_ms else None,
                     "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s"},
        "outputs_equal_serial": equal_serial["ok"],
        "outputs_equal_serial_detail": dict(equal_serial, what="every batch of the timed run (detection rows as gathered + loss_pose) == the same "
                                            "batch run serially on one stream after the timed region, bit for bit"),
        "prewarm_seconds": a.prewarm_seconds,
        "host_launch_ms_per_step": host_launch_ms,
        "step_submission": ("hipGraph replay (one graph per device slot" + (", the frame's H2D copy a node of it)" if a.graph_upload else ")") if a.graph else "eager launches")
                           + ("; batches alternate over %d HIP streams (per-kernel times from a separate single-stream pass)" % len(streams) if len(streams) > 1 else ""),
        "kernels_us": {k: round(v["avg_us"], 2) for k, v in sorted(kern.items())},
        "kernel_calls_per_step": {k: round(v["calls"] / a.steps, 2) for k, v in sorted(kern.items())},
    }
    if lat is not None:
        out["latency"] = lat
    out["process_group"] = ({"backend": torch.distributed.get_backend(), "ranks_seen": int(last.get("ranks_seen", 0)),
                             "world_size": world, "collective": "all_gather_into_tensor of the packed detection block, once per step"}
                            if torch.distributed.is_initialized() else
                            {"backend": None, "ranks_seen": 1, "world_size": 1, "collective": "none (single process; --force-process-group runs it through RCCL)"})
    out["per_rank"] = per_rank_fields(rank_elapsed, a.steps, 0 if a.resident_inputs else h2d_bytes)
    out["process_group"]["shared_device"] = bool(a.shared_device)
    out["process_group"]["host_numa"] = ({"node": numa[0], "cpus": numa[1]} if numa else None)
    # the collective by itself: latency of one all-gather of the packed block, outside the timed region. A
    # single process has no communicator during the measurement; a 1-rank RCCL group is created here, after it, for this number only.
    try:
        made = False
        if not torch.distributed.is_initialized() and world == 1 and not a.latency and not a.no_secondary:
            pdist.init_from_env(force=True)
            made = True
            ag_us = pdist.time_all_gather(det.rows, det.count)
        if isinstance(ag_us, str):
            raise RuntimeError(ag_us)
        out["process_group"]["all_gather_us"] = ag_us
        out["process_group"]["all_gather_note"] = ("one all_gather_into_tensor of the (cap+1) x 14 f32 block per rank, %s, back to back, outside the timed region%s"
                                                   % (torch.distributed.get_backend() if torch.distributed.is_initialized() else "no group",
                                                      "; 1-rank communicator created after the measurement for this number only" if made else ""))
    except Exception as e:
        out["process_group"]["all_gather_us"] = None
        out["process_group"]["all_gather_note"] = "failed: %r" % (e,)
    if (world == 1 and not a.no_secondary and not a.latency and not a.graph and a.config == "ycb" and cfg_name == "configs[2]"
            and "WORLD_SIZE" not in os.environ):
        out["secondary"] = secondary_configs()
    if world == 1 and not a.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(a, K, H, W, C, extents, symmetry, net, train)
            out["cpu_baseline"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        except Exception as e:  # the baseline is reported, never required for the GPU number
            out["cpu_baseline"] = {"error": repr(e)}
    # the JSON line is the LAST thing this process writes: tear the process group down first and flush the C
    # stdio buffer (RCCL prints a version banner through printf, which would otherwise land after the line)
    pdist.shutdown()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
