set -x
O=/root/repo/gpurun_out/${1:-r2p}; mkdir -p $O
cd /root/repo
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --streams 1 --no-cpu-baseline > $O/bench_streams1.json 2> $O/bench_streams1.err
python bench.py --streams 1 --graph --no-cpu-baseline > $O/bench_graph.json 2> $O/bench_graph.err
python bench.py --streams 1 --resident-inputs --no-cpu-baseline > $O/bench_resident.json 2> $O/bench_resident.err
python bench.py --streams 1 --latency --batch 1 --input COLOR --losses test --no-cpu-baseline --graph > $O/bench_latency_b1.json 2> $O/bench_latency_b1.err
python bench.py --config linemod --no-cpu-baseline > $O/bench_linemod.json 2> $O/bench_linemod.err
python bench.py --force-process-group --no-cpu-baseline > $O/bench_rccl_world1.json 2> $O/bench_rccl_world1.err
python bench.py --input COLOR --losses test --resident-inputs --no-cpu-baseline > $O/bench_color_test_resident.json 2> $O/bench_color.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python /root/repo/bench.py --steps 10 --warmup 3 --prewarm-seconds 4 --no-cpu-baseline > $O/bench_traced.json 2> $O/prof.log
python /root/repo/tools/rocprof_summary.py $O/prof/bench_results.db --marker hv_emit_kernel --steps 8 > $O/kernel_stats.csv 2> $O/kernel_stats.err
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -c1-8 | tr ' ' '_')
  rocprofv3 --pmc $set --kernel-trace -d $O/pmc_$tag -o pmc -- python /root/repo/tools/bench_ops.py --ops hough --iters 3 > $O/pmc_$tag.log 2>&1
  python /root/repo/tools/pmc_summary.py $O/pmc_$tag --match hv_ > $O/pmc_$tag.csv 2>> $O/pmc_$tag.log
done
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE"; do
  tag=w_$(echo $set | cut -c1-10 | tr ' ' '_')
  rocprofv3 --pmc $set --kernel-trace -d $O/pmc_$tag -o pmc -- python /root/repo/tools/bench_wino_mfma.py --layers conv3_2,conv4_2 --no-library > $O/pmc_$tag.log 2>&1
  python /root/repo/tools/pmc_summary.py $O/pmc_$tag --match wino > $O/pmc_$tag.csv 2>> $O/pmc_$tag.log
done
cd /root/repo
python tools/bench_wino_mfma.py --no-library > $O/layers_mfma.json 2> $O/layers_mfma.err
python tools/bench_ops.py > $O/ops.json 2> $O/ops.err
python tools/bench_wino_mfma.py --no-library --batch 1 --groups 1 > $O/layers_mfma_batch1.json 2>> $O/layers_mfma.err
python tools/bench_fc_rows.py > $O/fc_rows.txt 2>&1
python tools/bench_roi_pool.py > $O/roi_pool.txt 2>&1
rm -rf $O/prof/*.db $O/pmc_*/ 2>/dev/null
ls -la $O
