set -x
O=/root/repo/gpurun_out/r5l; mkdir -p $O
cd /root/repo
for v in 1 2 3; do
timeout 300 python bench.py --latency --batch 1 --input COLOR --losses none --graph --raw-inputs --steps 200 --warmup 5 --prewarm-seconds 2 --no-cpu-baseline --no-secondary > $O/latency_b1_$v.json 2> $O/latency.err
python -c "
import json
j=json.loads([l for l in open('$O/latency_b1_$v.json') if l.startswith('{')][-1]); print(j['latency'], j['value'], j['step_submission'])"
done
tail -3 $O/latency.err
timeout 600 python -m pytest tests/test_gpu_round2.py -x -q -k "graph" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
