# Round-3 PMC collection for the trunk MFMA kernel (final build): separate --pmc passes (no tracing domains besides
# --kernel-trace), summaries -> gpurun_out/$1/*.csv.   bash tools/collect_pmc_r3.sh r3pmc
set -x
O=/root/repo/gpurun_out/${1:-r3pmc}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_IFETCH SQ_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o pmc -- python /root/repo/tools/bench_wino_mfma.py --layers ${LAYERS:-conv1_2,conv3_2,conv4_2} --no-library > $O/p$i.log 2>&1
  python /root/repo/tools/pmc_summary.py $O/p$i --match wino43_mfma > $O/p$i.csv 2>> $O/p$i.log
  rm -rf $O/p$i
done
cat $O/p*.csv | grep -v "^kernel,counter" > $O/wino_mfma_pmc.csv
cat $O/wino_mfma_pmc.csv
