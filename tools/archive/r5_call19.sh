set -x
O=/root/repo/gpurun_out/r5y; mkdir -p $O
cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q -k "average_distance or adl or golden or training or batch_pipeline_rgbd or loss" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python tools/probe_adl.py > $O/probe_adl.json 2> $O/probe.err; cat $O/probe_adl.json | tr -d "\n "
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAVES"; do
  i=$((i+1))
  ADL_PROBE_ONLY="63 live, all symmetric" timeout 300 rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o pmc -- python /root/repo/tools/probe_adl.py > $O/p$i.log 2>&1
  python /root/repo/tools/pmc_summary.py $O/p$i > $O/p$i.csv 2>> $O/p$i.log
  grep adl_terms $O/p$i.csv
  rm -rf $O/p$i
done
