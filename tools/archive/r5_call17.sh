set -x
O=/root/repo/gpurun_out/r5r; mkdir -p $O
cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q -k "average_distance or adl or golden or training or batch_pipeline_rgbd or loss" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python tools/probe_adl.py > $O/probe_adl.json 2> $O/probe.err; cat $O/probe_adl.json
for i in 1 2; do
  timeout 600 python bench.py --no-cpu-baseline --no-secondary --prewarm-seconds 4 > $O/bench.$RANDOM.json 2>> $O/bench.err
done
python - <<'PY'
import glob, json
for f in sorted(glob.glob("/root/repo/gpurun_out/r5r/bench.*.json")):
    j = json.loads([l for l in open(f) if l.startswith("{")][-1])
    k = j["kernels_us"]
    print(f.split("/")[-1], round(j["value"], 1), round(j["ms_per_step"], 3), k.get("adl_terms_kernel"), j["outputs_equal_serial"])
PY
