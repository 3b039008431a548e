set -x
O=/root/repo/gpurun_out/r5al; mkdir -p $O
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q -k "hough or golden or pipeline or three_streams" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for i in 1 2; do
  timeout 600 python bench.py --no-cpu-baseline --no-secondary --prewarm-seconds 4 > $O/bench.$RANDOM.json 2>> $O/bench.err
done
python - <<'PY'
import glob, json
for f in sorted(glob.glob("/root/repo/gpurun_out/r5al/bench.*.json")):
    j = json.loads([l for l in open(f) if l.startswith("{")][-1])
    k = j["kernels_us"]
    print(f.split("/")[-1], round(j["value"], 1), round(j["ms_per_step"], 3), {n: v for n, v in k.items() if "hv_" in n}, j["outputs_equal_serial"])
PY
