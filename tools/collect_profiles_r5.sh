# Round-5 profile set (final build): bash tools/collect_profiles_r5.sh r5q -> gpurun_out/r5q/* (copy what is quoted into profiles/r05_*).
set -x
O=/root/repo/gpurun_out/${1:-r5q}; mkdir -p $O
cd /root/repo
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python bench.py --streams 1 --no-cpu-baseline --no-secondary > $O/bench_streams1.json 2> $O/bench_streams.err
python bench.py --graph --no-cpu-baseline --no-secondary > $O/bench_graph.json 2> $O/bench_graph.err
python bench.py --latency --batch 1 --input COLOR --losses none --no-cpu-baseline --graph --raw-inputs --steps 200 --no-secondary > $O/bench_latency_b1_inference_raw.json 2> $O/bench_latency.err
python bench.py --config linemod --no-cpu-baseline --no-secondary > $O/bench_linemod.json 2> $O/bench_linemod.err
python bench.py --force-process-group --no-cpu-baseline --no-secondary > $O/bench_rccl_world1.json 2> $O/bench_rccl_world1.err
python bench.py --gpus 2 --backend gloo --shared-device --steps 10 --warmup 3 --prewarm-seconds 2 --no-cpu-baseline --no-secondary > $O/bench_two_ranks_one_gpu.json 2> $O/bench_two_ranks.err
python tools/soak_streams.py --rounds 8 > $O/streams_soak.txt 2>&1
python tools/bench_backproject.py > $O/backproject.json 2> $O/backproject.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python /root/repo/bench.py --steps 10 --warmup 3 --prewarm-seconds 4 --no-cpu-baseline --no-secondary > $O/bench_traced.json 2> $O/prof.log
python /root/repo/tools/rocprof_summary.py $O/prof/bench_results.db --marker hv_emit_kernel --steps 8 > $O/bench_kernel_stats.csv 2> $O/kernel_stats.err
rm -rf $O/prof
cd /root/repo
python tools/bench_wino_mfma.py --no-library > $O/layers_mfma.json 2> $O/layers_mfma.err
ls -la $O
