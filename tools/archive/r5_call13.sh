set -x
O=/root/repo/gpurun_out/r5m; mkdir -p $O
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round3.py tests/test_gpu_pipeline.py -x -q -k "merged_head or head_lowres or pipeline or single_frame or latency or fc_rows" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for v in 1 2; do
timeout 300 python bench.py --latency --batch 1 --input COLOR --losses none --graph --raw-inputs --steps 200 --warmup 5 --prewarm-seconds 2 --no-cpu-baseline --no-secondary > $O/latency_b1_$v.json 2> $O/latency.err
python -c "
import json
j=json.loads([l for l in open('$O/latency_b1_$v.json') if l.startswith('{')][-1]); print(j['latency'], j['value'], sorted(j['kernel_calls_per_step'].items()))"
done
