# round 5, GPU call 2: backproject hit-path variants (parity + timing), then the whole GPU suite on the pruned library
set -x
O=/root/repo/gpurun_out/r5b; mkdir -p $O
cd /root/repo
for v in 0 1 2 3; do
  PCNN_BP_VARIANT=$v timeout 300 python -m pytest tests/test_gpu_round4.py -x -q -k "backproject_at_grid_128" > $O/pytest_bp_v$v.log 2>&1
  PCNN_BP_VARIANT=$v timeout 300 python tools/bench_backproject.py --grids 256,128 > $O/bp_v$v.json 2>> $O/bp.err
done
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
