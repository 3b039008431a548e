set -x
O=/root/repo/gpurun_out/r5am; mkdir -p $O
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q -k "deconv or upscore or linemod or backproject or heads or training" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python bench.py --config linemod --no-cpu-baseline --no-secondary > $O/bench_linemod.json 2>> $O/bench.err
python - <<'PY'
import json
j = json.loads([l for l in open("/root/repo/gpurun_out/r5am/bench_linemod.json") if l.startswith("{")][-1])
k = j["kernels_us"]
print(round(j["value"], 1), round(j["ms_per_step"], 3), {n: v for n, v in k.items() if "deconv" in n or "backproject" in n}, j["outputs_equal_serial"])
PY
