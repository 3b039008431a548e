set -x
O=/root/repo/gpurun_out/r5o; mkdir -p $O
cd /root/repo
for v in 0 1 0 1; do
  PCNN_WINO_IN_NT=$v timeout 600 python tools/bench_wino_mfma.py --no-library > $O/layers_nt$v.$RANDOM.json 2>> $O/layers.err
done
for v in 0 1 0 1; do
  PCNN_WINO_IN_NT=$v timeout 600 python bench.py --no-cpu-baseline --no-secondary --prewarm-seconds 4 > $O/bench_nt$v.$RANDOM.json 2>> $O/bench.err
done
python - <<'PY'
import glob, json
for f in sorted(glob.glob("/root/repo/gpurun_out/r5o/layers_*.json")):
    a = json.load(open(f)); print(f.split("/")[-1], a["total"], {k: v["input_transform_ms"] for k, v in a["layers"].items()})
for f in sorted(glob.glob("/root/repo/gpurun_out/r5o/bench_*.json")):
    j = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(f.split("/")[-1], round(j["value"], 1), round(j["ms_per_step"], 3), j["kernels_us"].get("wino43_input_kernel"), j["kernels_us"].get("wino43_mfma_kernel"), j["outputs_equal_serial"])
PY
