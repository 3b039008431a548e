set -x
O=/root/repo/gpurun_out/r5ag; mkdir -p $O
cd /root/repo
T=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.$T.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --no-secondary --prewarm-seconds 4 > $O/bench_short.$T.json 2>> $O/bench.err
python - <<'PY'
import glob, json
for f in sorted(glob.glob("/root/repo/gpurun_out/r5ag/bench_*.json")):
    j = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(f.split("/")[-1], round(j["value"], 1), round(j["ms_per_step"], 3), j["outputs_equal_serial"])
PY
