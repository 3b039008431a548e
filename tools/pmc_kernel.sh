# bash tools/pmc_kernel.sh <tag> <kernel-name-substring> -- <command ...>     five --pmc passes, summary to gpurun_out/<tag>.csv
tag=$1; match=$2; shift 3
O=/root/repo/gpurun_out/$tag; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o pmc -- "$@" > $O/p$i.log 2>&1
  python /root/repo/tools/pmc_summary.py $O/p$i --match "$match" > $O/p$i.csv 2>> $O/p$i.log
  rm -rf $O/p$i
done
cat $O/p*.csv | grep -v "^kernel,counter" > /root/repo/gpurun_out/$tag.csv
cat /root/repo/gpurun_out/$tag.csv
