# round 5, GPU call 3: backproject with LDS match lists; conv12 / mfma latency hoists (probe A/B, equality tests, per-layer bench); step
set -x
O=/root/repo/gpurun_out/r5c; mkdir -p $O
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_round4.py tests/test_gpu_round2.py -x -q -k "backproject or conv1_1_conv1_2 or winograd43_mfma or block_maps or first_conv" > $O/pytest.log 2>&1
tail -3 $O/pytest.log
timeout 300 python tools/bench_backproject.py --grids 256,128 > $O/bp.json 2> $O/bp.err
tools/conv12_probe 1 > $O/conv12_probe_hoist.txt 2>&1
tools/conv12_probeNO_HOIST 1 > $O/conv12_probe_nohoist.txt 2>&1
tools/conv12_probe 1 >> $O/conv12_probe_hoist.txt 2>&1
tools/conv12_probeNO_HOIST 1 >> $O/conv12_probe_nohoist.txt 2>&1
timeout 600 python tools/bench_wino_mfma.py --no-library > $O/layers_mfma.json 2> $O/layers_mfma.err
timeout 600 python bench.py --no-cpu-baseline --no-secondary > $O/bench_default.json 2> $O/bench_default.err
