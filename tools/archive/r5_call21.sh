set -x
O=/root/repo/gpurun_out/r5ae; mkdir -p $O
cd /root/repo
for v in nt plain nt plain; do
  cp tools/lib_$v.so posecnn_amd/libposecnn_hip.so
  timeout 600 python bench.py --no-cpu-baseline --no-secondary --prewarm-seconds 4 > $O/bench_$v.$RANDOM.json 2>> $O/bench.err
done
python - <<'PY'
import glob, json
for f in sorted(glob.glob("/root/repo/gpurun_out/r5ae/bench_*.json")):
    j = json.loads([l for l in open(f) if l.startswith("{")][-1])
    k = j["kernels_us"]
    print(f.split("/")[-1], round(j["value"], 1), round(j["ms_per_step"], 3), {n: v for n, v in k.items() if "upscore" in n or "roi" in n or "hv_vote" in n or "conv12" in n}, round(sum(k.values())), j["outputs_equal_serial"])
PY
