# round 5, GPU call 11: latency with the all-pairs NMS / spin drain; full GPU suite
set -x
O=/root/repo/gpurun_out/r5k; mkdir -p $O
cd /root/repo
timeout 300 python bench.py --latency --batch 1 --input COLOR --losses none --graph --raw-inputs --steps 200 --warmup 5 --prewarm-seconds 2 --no-cpu-baseline --no-secondary > $O/latency_b1.json 2> $O/latency.err
python -c "
import json
j=json.loads([l for l in open('$O/latency_b1.json') if l.startswith('{')][-1]); print(j['latency'], j['value'])"
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
