# round 5, GPU call 10: conv12 packed-f32 transforms (equality), step, PMC of one step
set -x
O=/root/repo/gpurun_out/r5j; mkdir -p $O
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round2.py tests/test_gpu_round5.py -x -q -k "conv1_1_conv1_2 or raw_frame or network_on_raw or head_lowres or fc_rows_cols or pose_l2" > $O/pytest.log 2>&1
tail -3 $O/pytest.log
tools/conv12_probe 1 > $O/conv12_probe.txt 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-secondary > $O/bench_default.json 2> $O/bench_default.err
bash tools/collect_pmc_step.sh $O/pmc > $O/pmc.log 2>&1
tail -40 $O/pmc.log
