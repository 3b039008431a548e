# cache / latency counters of backproject_fused_kernel at G = 256 (profiles/r06_backproject_trace.txt §2). Every pass under `timeout`: a
# TA_* set (TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum) aborted rocprofv3 with signal 6 and hung the call for 15 minutes.
cd /tmp; export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/bp_pmc; mkdir -p $O
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum TCC_EA_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  timeout 180 rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o pmc -- python $R/tools/bench_backproject.py --once --grids 256 --kinds smooth > $O/p$i.log 2>&1
  python $R/tools/pmc_summary.py $O/p$i --match backproject_fused > $O/p$i.csv 2>> $O/p$i.log
  rm -rf $O/p$i
  cat $O/p$i.csv | tail -n +2
  tail -2 $O/p$i.log | grep -i "error\|invalid\|not" | head -3
done
