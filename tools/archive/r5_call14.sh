set -x
O=/root/repo/gpurun_out/r5n; mkdir -p $O
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_round4.py -x -q -k "block_maps" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for v in scalar packed scalar packed; do
  if [ $v = packed ]; then export PCNN_WINO_MODE=4; else unset PCNN_WINO_MODE; fi
  timeout 600 python tools/bench_wino_mfma.py --no-library > $O/layers_$v.$RANDOM.json 2>> $O/layers.err
done
for v in scalar packed scalar packed; do
  if [ $v = packed ]; then export PCNN_WINO_MODE=4; else unset PCNN_WINO_MODE; fi
  timeout 600 python bench.py --no-cpu-baseline --no-secondary --prewarm-seconds 4 > $O/bench_$v.$RANDOM.json 2>> $O/bench.err
done
unset PCNN_WINO_MODE
python - <<'PY'
import glob, json
for f in sorted(glob.glob("/root/repo/gpurun_out/r5n/layers_*.json")):
    a = json.load(open(f)); print(f.split("/")[-1], a["total"]["mfma_ms"], {k: v["mfma_ms"] for k, v in a["layers"].items()})
for f in sorted(glob.glob("/root/repo/gpurun_out/r5n/bench_*.json")):
    j = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(f.split("/")[-1], round(j["value"], 1), round(j["ms_per_step"], 3), j["kernels_us"].get("wino43_mfma_kernel"), j["outputs_equal_serial"])
PY
