set -x
O=/root/repo/gpurun_out/r5ah; mkdir -p $O
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q -k "head or pipeline or merged or golden" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for i in 1 2; do
  timeout 600 python bench.py --no-cpu-baseline --no-secondary --prewarm-seconds 4 > $O/bench.$RANDOM.json 2>> $O/bench.err
done
timeout 300 python bench.py --latency --batch 1 --input COLOR --losses none --no-cpu-baseline --graph --raw-inputs --steps 200 --no-secondary > $O/latency.json 2>> $O/bench.err
python - <<'PY'
import glob, json
for f in sorted(glob.glob("/root/repo/gpurun_out/r5ah/bench.*.json")):
    j = json.loads([l for l in open(f) if l.startswith("{")][-1])
    k = j["kernels_us"]
    print(f.split("/")[-1], round(j["value"], 1), round(j["ms_per_step"], 3), {n: v for n, v in k.items() if "head" in n}, j["outputs_equal_serial"])
j = json.loads([l for l in open("/root/repo/gpurun_out/r5ah/latency.json") if l.startswith("{")][-1]); print(j["latency"])
PY
