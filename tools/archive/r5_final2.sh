# round 5, final evidence run on the final build (after the average_distance / label head / roi_pool work): step PMC, profile
# set, the three ADL / VALU probes, full GPU suite, smoke
set -x
O=/root/repo/gpurun_out/r5fin3; mkdir -p $O
cd /root/repo
bash tools/collect_pmc_step.sh $O/pmc > $O/pmc.log 2>&1
bash tools/collect_profiles_r5.sh r5fin3 > $O/collect.log 2>&1
python tools/probe_adl.py > $O/adl_probe.json 2> $O/adl_probe.err
tools/adl_stamp_probe 1 > $O/adl_stamp_probe.txt 2>&1; tools/adl_stamp_probe 0 >> $O/adl_stamp_probe.txt 2>&1
tools/valu_rate_probe > $O/valu_rate_probe.txt 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
