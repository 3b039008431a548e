# round 5, GPU call 9: direct MFMA epilogue (equality with the staged one, per-layer and step A/B), ADL packed scan parity, literal pipeline tests
set -x
O=/root/repo/gpurun_out/r5i; mkdir -p $O
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round2.py tests/test_gpu_ops.py tests/test_gpu_pipeline.py tests/test_gpu_round3.py -x -q -k "wino or conv1_1 or average_distance or adl or pipeline or fused_heads or trunk or golden" > $O/pytest.log 2>&1
tail -4 $O/pytest.log
timeout 600 python tools/bench_wino_mfma.py --no-library > $O/layers_mfma_direct.json 2> $O/layers.err
PCNN_WINO_MODE=4 timeout 600 python tools/bench_wino_mfma.py --no-library > $O/layers_mfma_staged.json 2>> $O/layers.err
for v in direct staged direct staged; do
  if [ $v = staged ]; then export PCNN_WINO_MODE=4; else unset PCNN_WINO_MODE; fi
  timeout 600 python bench.py --no-cpu-baseline --no-secondary --prewarm-seconds 4 > $O/bench_$v.$RANDOM.json 2>> $O/bench.err
done
unset PCNN_WINO_MODE
python - <<'PY'
import glob, json
for f in sorted(glob.glob("/root/repo/gpurun_out/r5i/bench_*.json")):
    j = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(f.split("/")[-1], round(j["value"], 1), round(j["ms_per_step"], 3), j["kernels_us"].get("wino43_mfma_kernel"), j["kernels_us"].get("adl_terms_kernel"), j["outputs_equal_serial"])
for t in ("direct", "staged"):
    a = json.load(open("/root/repo/gpurun_out/r5i/layers_mfma_%s.json" % t))
    print(t, a["total"], {k: v["mfma_ms"] for k, v in a["layers"].items()})
PY
