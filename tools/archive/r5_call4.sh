# round 5, GPU call 4: backproject v5 (class quads), head / fc8 / normalize / packed-block kernels, whole-pipeline parity, step
set -x
O=/root/repo/gpurun_out/r5d; mkdir -p $O
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_ops.py tests/test_gpu_round4.py tests/test_gpu_round3.py tests/test_gpu_pipeline.py tests/test_gpu_round2.py -x -q -k "not three_streams and not two_ranks and not bench_reports" > $O/pytest.log 2>&1
tail -5 $O/pytest.log
timeout 300 python tools/bench_backproject.py --grids 256,128 > $O/bp.json 2> $O/bp.err
timeout 600 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python /root/repo/bench.py --steps 10 --warmup 3 --prewarm-seconds 3 --no-cpu-baseline --no-secondary > $O/bench_traced.json 2> $O/prof.log
python /root/repo/tools/rocprof_summary.py $O/prof/bench_results.db --marker hv_emit_kernel --steps 8 > $O/bench_kernel_stats.csv 2> $O/kernel_stats.err
rm -rf $O/prof
