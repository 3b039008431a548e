set -x
O=/root/repo/gpurun_out/r5ai; mkdir -p $O
cd /root/repo
for v in 64 128 256 64 128 256; do
  cp tools/lib_seg$v.so posecnn_amd/libposecnn_hip.so
  timeout 600 python bench.py --no-cpu-baseline --no-secondary --prewarm-seconds 3 --steps 10 > $O/bench_seg$v.$RANDOM.json 2>> $O/bench.err
done
cp tools/lib_seg128.so posecnn_amd/libposecnn_hip.so
python - <<'PY'
import glob, json
for f in sorted(glob.glob("/root/repo/gpurun_out/r5ai/bench_*.json")):
    j = json.loads([l for l in open(f) if l.startswith("{")][-1])
    k = j["kernels_us"]
    print(f.split("/")[-1], round(j["value"], 1), round(j["ms_per_step"], 3), {n: v for n, v in k.items() if "upscore" in n}, j["outputs_equal_serial"])
PY
