"""CPU tests of bench.py's launch logic: `python bench.py --gpus N` with no
WORLD_SIZE must start N ranks by itself, rendezvous on 127.0.0.1, run the detection all-gather and
print one JSON line from rank 0. `--dry-run` keeps everything except the GPU work and uses gloo."""
import json
import os
import subprocess

import pytest
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["OMP_NUM_THREADS"] = "1"
    return env


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_respawn_command_shape():
    sys.path.insert(0, ROOT)
    import bench
    argv = ["--gpus", "8", "--steps", "3", "--warmup", "1"]
    cmd = bench.respawn_command(bench.parse_args(argv), argv)
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-len(argv) - 1] == BENCH and cmd[-len(argv):] == argv


def test_gpus2_self_spawns_two_ranks_and_gathers_on_gloo():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run", "--streams", "2"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["process_group"] is True
    assert out["detections_gathered_per_step"] == 6   # 3 rows from each of the 2 ranks
    assert out["ranks_seen"] == 2 and out["batches_in_flight"] == 2   # --streams 2: two collectives in flight, issued in the same order on every rank
    # round 6: the spread fields and the per-rank step times travel with the line at world > 1
    pr = out["per_rank"]
    assert pr["ranks"] == 2 and len(pr["ms_per_step_by_rank"]) == 2 and len(pr["h2d_GBps_per_rank"]) == 2
    assert pr["ms_per_step_min"] <= pr["ms_per_step_median"] <= pr["ms_per_step_max"]
    assert out["repeats"] == 1 and out["value_min"] <= out["value_median"] <= out["value_max"]


def test_inputs_env_override(monkeypatch):
    """the frame hand-over of the driver's fixed `bench.py --gpus 8 ...` line is selectable through the environment."""
    import bench
    for val, want in (("raw", (True, False)), ("resident", (False, True)), ("pinned", (False, False)), ("", (False, False))):
        monkeypatch.setenv("PCNN_BENCH_INPUTS", val)
        a = bench.parse_args(["--gpus", "8"])
        assert (a.raw_inputs, a.resident_inputs) == want, val
    monkeypatch.setenv("PCNN_BENCH_INPUTS", "floppy")
    with pytest.raises(SystemExit):
        bench.parse_args([])


def test_spread_and_per_rank_fields():
    import bench
    f = bench.spread_fields(320, [0.40, 0.42, 0.39, 0.41, 0.40], 20)
    assert f["repeats"] == 5 and f["value_all"][0] == 800.0 and abs(f["value_median"] - 800.0) < 1e-9
    assert abs(f["value_min"] - 320 / 0.42) < 1e-9 and abs(f["value_max"] - 320 / 0.39) < 1e-9
    assert f["ms_per_step_all"] == [20.0, 21.0, 19.5, 20.5, 20.0]
    r = bench.per_rank_fields([0.40, 0.44, 0.42, 0.48], 20, 118_000_000)
    assert r["ms_per_step_min"] == 20.0 and r["ms_per_step_max"] == 24.0 and abs(r["ms_per_step_median"] - 21.5) < 1e-9
    assert r["h2d_GBps_per_rank"][0] == 5.9
    assert "h2d_GBps_per_rank" not in bench.per_rank_fields([0.4], 20, 0)


def test_gpus2_single_lane_and_three_lanes():
    for lanes in (1, 3):
        r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "5", "--warmup", "0", "--dry-run", "--streams", str(lanes)],
                           env=_env(), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        out = _json_line(r.stdout)
        assert out["ranks_seen"] == 2 and out["batches_in_flight"] == lanes and out["detections_gathered_per_step"] == 6


def test_under_a_launcher_env_is_respected():
    # what the driver does for N > 1: torch.distributed.run ... bench.py --gpus N
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "2",
                        "--warmup", "0", "--dry-run"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert _json_line(r.stdout)["n_gpus"] == 2


def test_world1_force_process_group_runs_the_collective():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "2", "--warmup", "0", "--dry-run",
                        "--force-process-group"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 1 and out["process_group"] is True and out["detections_gathered_per_step"] == 3


def test_step_pmc_file_belongs_to_this_build_of_the_kernels():
    """bench.py's `roofline.traffic`, `valu_frac` and `roofline_dominant.mfma_busy_share` come
    from PMC counters collected in separate rocprofv3 passes (profiles/r06_step_pmc.json, tools/collect_pmc_step.sh). The file
    records the hash of every kernel source it was collected on; bench.py drops a kernel's counters when its source
    differs from the build it runs — and this test fails for EVERY source the file names (round 5 checked the Hough and trunk
    files only, and a comment-only commit to average_distance.hip silently dropped the loss kernels' counters)."""
    import hashlib
    import json
    pmc = json.load(open(os.path.join(ROOT, "profiles", "r06_step_pmc.json")))
    assert {"hough_voting.hip", "wino_mfma.hip", "conv_first.hip", "winograd.hip", "average_distance.hip", "roi_pool.hip",
            "fc_mfma.hip", "upscore.hip"} <= set(pmc["_source_sha16"])
    for src, want in pmc["_source_sha16"].items():
        sha = hashlib.sha256(open(os.path.join(ROOT, "posecnn_amd", "csrc", src), "rb").read()).hexdigest()[:16]
        assert want == sha, "profiles/r06_step_pmc.json was collected on another build of %s: re-run tools/collect_pmc_step.sh" % src
    for kernel in ("hv_vote_kernel", "wino43_mfma_kernel", "conv12_wino43_fused_kernel", "adl_terms_kernel", "roi_pool_add2_rows"):
        ent = pmc[kernel]
        assert ent["_src"] in pmc["_source_sha16"]
        assert ent["SQ_INSTS_VALU"] > 0 and ent["FETCH_SIZE_KB"] > 0 and ent["WRITE_SIZE_KB"] > 0, kernel
    assert pmc["wino43_mfma_kernel"]["SQ_VALU_MFMA_BUSY_CYCLES"] > 0 and pmc["wino43_mfma_kernel"]["SQ_INSTS_MFMA"] > 0
