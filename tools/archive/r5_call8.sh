# round 5, GPU call 8: roi_pool_add2 batched scans (parity), configs[1] latency with either first-layers kernel
set -x
O=/root/repo/gpurun_out/r5h; mkdir -p $O
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_round4.py tests/test_gpu_round2.py tests/test_gpu_pipeline.py -x -q -k "roi_pool or pipeline" > $O/pytest.log 2>&1
tail -3 $O/pytest.log
for v in 1 0 1 0; do
  PCNN_CONV12_PAIRS=$v timeout 300 python bench.py --latency --batch 1 --input COLOR --losses none --graph --raw-inputs --steps 200 --warmup 5 --prewarm-seconds 2 --no-cpu-baseline --no-secondary > $O/latency_pairs$v.$RANDOM.json 2>> $O/latency.err
done
python - <<'PY'
import glob, json
for f in sorted(glob.glob("/root/repo/gpurun_out/r5h/latency_pairs*.json")):
    j = json.loads([l for l in open(f) if l.startswith("{")][-1])
    k = [n for n in j["kernels_us"] if n.startswith(("conv12", "roi_pool"))]
    print(f.split("/")[-1], j["latency"], {n: j["kernels_us"][n] for n in k})
PY
timeout 300 python tools/bench_roi_pool.py > $O/roi_pool.txt 2>&1
tail -5 $O/roi_pool.txt
