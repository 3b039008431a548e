# round 5, GPU call 7: the two-workgroups-per-CU first-layers kernel: equality with the unfused pair, A/B against round 4's kernel
set -x
O=/root/repo/gpurun_out/r5g; mkdir -p $O
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round2.py -x -q -k "conv1_1_conv1_2 or raw_frame or network_on_raw" > $O/pytest.log 2>&1
tail -3 $O/pytest.log
for v in 1 0 1 0; do
  PCNN_CONV12_PAIRS=$v timeout 600 python bench.py --no-cpu-baseline --no-secondary --prewarm-seconds 4 > $O/bench_pairs$v.$RANDOM.json 2>> $O/bench.err
done
python - <<'PY'
import glob, json
for f in sorted(glob.glob("/root/repo/gpurun_out/r5g/bench_pairs*.json")):
    j = json.loads([l for l in open(f) if l.startswith("{")][-1])
    k = [n for n in j["kernels_us"] if n.startswith("conv12")]
    print(f.split("/")[-1], round(j["value"], 1), round(j["ms_per_step"], 3), {n: j["kernels_us"][n] for n in k}, j["outputs_equal_serial"])
PY
