# Round-4 profile set (final build). bash tools/collect_profiles_r4.sh r4p  -> gpurun_out/r4p/*  (copy what is quoted into profiles/r04_*)
set -x
O=/root/repo/gpurun_out/${1:-r4p}; mkdir -p $O
cd /root/repo
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --streams 1 --no-cpu-baseline --no-secondary > $O/bench_streams1.json 2> $O/bench_streams1.err
python bench.py --streams 3 --no-cpu-baseline --no-secondary > $O/bench_streams3.json 2> $O/bench_streams3.err
python bench.py --no-fused-conv12 --no-cpu-baseline --no-secondary > $O/bench_unfused_conv12.json 2> $O/bench_unfused.err
python bench.py --no-fused-conv12 --streams 1 --no-cpu-baseline --no-secondary > $O/bench_unfused_conv12_streams1.json 2>> $O/bench_unfused.err
tools/conv12_probe > $O/conv12_probe.txt 2>&1
for v in NO_B1 NO_A1; do echo "== ablation $v" >> $O/conv12_probe.txt; tools/conv12_probe$v >> $O/conv12_probe.txt 2>&1; done
python bench.py --graph --no-cpu-baseline > $O/bench_graph.json 2> $O/bench_graph.err
python bench.py --latency --batch 1 --input COLOR --losses none --no-cpu-baseline --graph --raw-inputs --steps 200 > $O/bench_latency_b1_inference_raw.json 2> $O/bench_latency.err
python bench.py --latency --batch 1 --input COLOR --losses test --no-cpu-baseline --graph --steps 200 > $O/bench_latency_b1.json 2>> $O/bench_latency.err
python bench.py --config linemod --no-cpu-baseline > $O/bench_linemod.json 2> $O/bench_linemod.err
python bench.py --force-process-group --no-cpu-baseline --no-secondary > $O/bench_rccl_world1.json 2> $O/bench_rccl_world1.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python /root/repo/bench.py --steps 10 --warmup 3 --prewarm-seconds 4 --no-cpu-baseline --no-secondary > $O/bench_traced.json 2> $O/prof.log
python /root/repo/tools/rocprof_summary.py $O/prof/bench_results.db --marker hv_emit_kernel --steps 8 > $O/bench_kernel_stats.csv 2> $O/kernel_stats.err
rm -rf $O/prof
cd /root/repo
bash tools/collect_pmc_hough.sh $O > $O/hough_pmc.log 2>&1
LAYERS=conv4_2 bash tools/collect_pmc_r3.sh ${1:-r4p}/wino_pmc_conv4_2 > $O/wino_pmc.log 2>&1
python tools/bench_wino_mfma.py --no-library > $O/layers_mfma.json 2> $O/layers_mfma.err
python tools/bench_wino_mfma.py --no-library --batch 1 --groups 1 > $O/layers_mfma_batch1.json 2>> $O/layers_mfma.err
python tools/bench_ops.py > $O/ops.json 2> $O/ops.err
python tools/bench_icp.py > $O/icp.json 2> $O/icp.err
python tests/parity_study.py --frames 64 --out $O/parity_study.json > /dev/null 2> $O/parity.err
ls -la $O
