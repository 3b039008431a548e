# round 5, GPU call 5: backproject (compacted hits) parity + timing + PMC on the parity scene; head kernel v2; ADL packed scan; 4 streams
set -x
O=/root/repo/gpurun_out/r5e; mkdir -p $O
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_ops.py tests/test_gpu_round4.py tests/test_gpu_round3.py tests/test_gpu_pipeline.py tests/test_gpu_round2.py tests/test_gpu_training.py -x -q -k "not three_streams and not two_ranks and not bench_reports" > $O/pytest.log 2>&1
tail -5 $O/pytest.log
timeout 300 python tools/bench_backproject.py --grids 256,128 > $O/bp.json 2> $O/bp.err
timeout 600 python bench.py --no-cpu-baseline --no-secondary > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python bench.py --no-cpu-baseline --no-secondary --streams 4 > $O/bench_streams4.json 2> $O/bench_streams4.err
bash tools/pmc_kernel.sh r5e_bp_smooth_pmc backproject_fused -- python /root/repo/tools/bench_backproject.py --once --grids 256 --kinds smooth > $O/pmc_bp.log 2>&1
