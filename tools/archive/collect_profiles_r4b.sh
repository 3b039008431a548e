# Round-4 profile set, final build (second half of the round: conv1_1 on the matrix cores, single-tower fusion, the opt-in
# one-wave trunk kernel). bash tools/collect_profiles_r4b.sh r4q -> gpurun_out/r4q/* (copy what is quoted into profiles/r04_*).
# The PMC sets (Hough, wino43_mfma conv4_2) are not re-collected: those kernels are unchanged (source-hash provenance, §6).
set -x
O=/root/repo/gpurun_out/${1:-r4q}; mkdir -p $O
cd /root/repo
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --streams 1 --no-cpu-baseline --no-secondary > $O/bench_streams1.json 2> $O/bench_streams.err
python bench.py --streams 2 --no-cpu-baseline --no-secondary > $O/bench_streams2.json 2>> $O/bench_streams.err
python bench.py --no-fused-conv12 --no-cpu-baseline --no-secondary > $O/bench_unfused_conv12.json 2> $O/bench_unfused.err
python bench.py --graph --no-cpu-baseline --no-secondary > $O/bench_graph.json 2> $O/bench_graph.err
python bench.py --latency --batch 1 --input COLOR --losses none --no-cpu-baseline --graph --raw-inputs --steps 200 > $O/bench_latency_b1_inference_raw.json 2> $O/bench_latency.err
python bench.py --latency --batch 1 --input COLOR --losses test --no-cpu-baseline --graph --steps 200 > $O/bench_latency_b1.json 2>> $O/bench_latency.err
python bench.py --config linemod --no-cpu-baseline --no-secondary > $O/bench_linemod.json 2> $O/bench_linemod.err
python bench.py --force-process-group --no-cpu-baseline --no-secondary > $O/bench_rccl_world1.json 2> $O/bench_rccl_world1.err
tools/conv12_probe 1 > $O/conv12_probe.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python /root/repo/bench.py --steps 10 --warmup 3 --prewarm-seconds 4 --no-cpu-baseline --no-secondary > $O/bench_traced.json 2> $O/prof.log
python /root/repo/tools/rocprof_summary.py $O/prof/bench_results.db --marker hv_emit_kernel --steps 8 > $O/bench_kernel_stats.csv 2> $O/kernel_stats.err
rm -rf $O/prof
cd /root/repo
python tools/bench_wino_mfma.py --no-library > $O/layers_mfma.json 2> $O/layers_mfma.err
python tests/parity_study.py --frames 64 --out $O/parity_study.json > /dev/null 2> $O/parity.err
ls -la $O
