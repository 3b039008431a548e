set -x
O=/root/repo/gpurun_out/r5aj; mkdir -p $O
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q -k "wino or winograd or trunk or golden or pipeline or variant" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python tools/bench_wino_mfma.py --no-library > $O/layers.json 2> $O/layers.err
for i in 1 2; do
  timeout 600 python bench.py --no-cpu-baseline --no-secondary --prewarm-seconds 4 > $O/bench.$RANDOM.json 2>> $O/bench.err
done
python - <<'PY'
import glob, json
a = json.load(open("/root/repo/gpurun_out/r5aj/layers.json")); print(a["total"], {k: v["input_transform_ms"] for k, v in a["layers"].items()})
for f in sorted(glob.glob("/root/repo/gpurun_out/r5aj/bench.*.json")):
    j = json.loads([l for l in open(f) if l.startswith("{")][-1])
    k = j["kernels_us"]
    print(f.split("/")[-1], round(j["value"], 1), round(j["ms_per_step"], 3), {n: v for n, v in k.items() if "wino" in n}, [(o["kernel"], round(o["frac"], 3), o["us_per_step"]) for o in j["roofline_other"] if "input" in o["kernel"]], j["outputs_equal_serial"])
PY
