# Round-6 profile set: bash tools/collect_profiles_r6.sh [dir under gpurun_out]   (on the GPU box; copy what is quoted into profiles/r06_*)
# = the step's PMC counters (tools/collect_pmc_step.sh: what bench.py's roofline fields read) FIRST, then the bench lines that read them,
# the traced run, the per-layer / per-op tools, the GPU suite and smoke(). Every rocprofv3 call is wrapped in `timeout`.
set -x
R=$(cd $(dirname $0)/.. && pwd)
O=$R/gpurun_out/${1:-r6q}; mkdir -p $O
cd $R
timeout 900 bash tools/collect_pmc_step.sh $O/pmc > $O/pmc.log 2>&1
cp $O/pmc/step_pmc.json profiles/r06_step_pmc.json; cp $O/pmc/step_pmc.csv profiles/r06_step_pmc.csv    # (on the box: so that the bench lines below quote THIS build's counters)
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python bench.py --streams 1 --no-cpu-baseline --no-secondary > $O/bench_streams1.json 2> $O/bench_streams.err
python bench.py --graph --no-cpu-baseline --no-secondary > $O/bench_graph.json 2> $O/bench_graph.err
python bench.py --resident-inputs --no-cpu-baseline --no-secondary > $O/bench_resident.json 2> $O/bench_resident.err
python bench.py --raw-inputs --no-cpu-baseline --no-secondary > $O/bench_raw.json 2> $O/bench_raw.err
python bench.py --latency --batch 1 --input COLOR --losses none --no-cpu-baseline --graph --raw-inputs --steps 200 --no-secondary > $O/bench_latency_b1_inference_raw.json 2> $O/bench_latency.err
python bench.py --config linemod --no-cpu-baseline --no-secondary > $O/bench_linemod.json 2> $O/bench_linemod.err
python bench.py --force-process-group --no-cpu-baseline --no-secondary > $O/bench_rccl_world1.json 2> $O/bench_rccl_world1.err
python bench.py --gpus 2 --backend gloo --shared-device --steps 10 --warmup 3 --prewarm-seconds 2 --no-cpu-baseline --no-secondary > $O/bench_two_ranks_one_gpu.json 2> $O/bench_two_ranks.err
python tools/soak_streams.py --rounds 8 > $O/streams_soak.txt 2>&1
python tools/bench_backproject.py > $O/backproject.json 2> $O/backproject.err
python tools/bench_fc_rows.py > $O/fc_rows.txt 2>&1
python tools/bench_ops.py > $O/ops.json 2> $O/ops.err
python tools/bench_wino_mfma.py --no-library > $O/layers_mfma.json 2> $O/layers_mfma.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py --steps 10 --warmup 3 --prewarm-seconds 4 --repeats 1 --no-cpu-baseline --no-secondary > $O/bench_traced.json 2> $O/prof.log
python $R/tools/rocprof_summary.py $O/prof/bench_results.db --marker hv_emit_kernel --steps 8 > $O/bench_kernel_stats.csv 2> $O/kernel_stats.err
rm -rf $O/prof
cd $R
python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
ls -la $O
