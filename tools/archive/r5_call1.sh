# round 5, GPU call 1: new equality tests, backproject parity + A/B, baseline bench on this box, PMC of backproject and conv12
set -x
O=/root/repo/gpurun_out/r5a; mkdir -p $O
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round4.py -x -q -k "round5 or backproject or three_streams or bench_reports or two_ranks" > $O/pytest_r5.log 2>&1
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -k backproject > $O/pytest_bp.log 2>&1
timeout 300 python tools/bench_backproject.py > $O/bp_nt1.json 2> $O/bp.err
PCNN_BP_NT=0 timeout 300 python tools/bench_backproject.py > $O/bp_nt0.json 2>> $O/bp.err
timeout 600 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
bash tools/pmc_kernel.sh r5a_bp_pmc backproject_fused -- python /root/repo/tools/bench_backproject.py --once --grids 256 > $O/pmc_bp.log 2>&1
ls -la $O
