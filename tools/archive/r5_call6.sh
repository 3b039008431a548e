# round 5, GPU call 6: MFMA heads at every batch size (parity of the single-frame paths + configs[1] latency), headline again
set -x
O=/root/repo/gpurun_out/r5f; mkdir -p $O
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_pipeline.py -x -q -k "not three_streams and not two_ranks and not bench_reports" > $O/pytest.log 2>&1
tail -3 $O/pytest.log
timeout 300 python bench.py --latency --batch 1 --input COLOR --losses none --graph --raw-inputs --steps 200 --warmup 5 --prewarm-seconds 2 --no-cpu-baseline --no-secondary > $O/latency_b1.json 2> $O/latency.err
timeout 600 python bench.py --no-cpu-baseline --no-secondary > $O/bench_default.json 2> $O/bench_default.err
