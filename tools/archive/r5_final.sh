# round 5, final evidence run on the final build: step PMC, profile set, full GPU suite, smoke
set -x
O=/root/repo/gpurun_out/r5q; mkdir -p $O
cd /root/repo
bash tools/collect_pmc_step.sh $O/pmc > $O/pmc.log 2>&1
bash tools/collect_profiles_r5.sh r5q > $O/collect.log 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
