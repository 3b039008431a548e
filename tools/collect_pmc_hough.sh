#!/bin/bash
# Hough PMC counters of THIS build -> profiles/r04_hough_pmc.{csv,json} (bench.py's `roofline.traffic`; the JSON records
# the hash of csrc/hough_voting.hip it belongs to — tests/test_bench_launch.py fails when the kernel changes without a
# re-collection). Separate rocprofv3 --pmc passes with --kernel-trace only (MI355X_MICROARCH.md, HBM section).
#   bash tools/collect_pmc_hough.sh gpurun_out/r4pmc      (on the GPU box), then copy the two files into profiles/
O=${1:-gpurun_out/r4pmc}; mkdir -p $O; O=$(cd $O && pwd)
R=$(cd $(dirname $0)/.. && pwd)
cd /tmp && export TMPDIR=/tmp
: > $O/hough_pmc.csv
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -c1-8 | tr ' ' '_')
  rocprofv3 --pmc $set --kernel-trace -d $O/pmc_$tag -o pmc -- python $R/tools/bench_ops.py --ops hough --iters 3 > $O/pmc_$tag.log 2>&1
  python $R/tools/pmc_summary.py $O/pmc_$tag --match hv_ > $O/pmc_$tag.csv 2>> $O/pmc_$tag.log
  if [ -s $O/hough_pmc.csv ]; then tail -n +2 $O/pmc_$tag.csv >> $O/hough_pmc.csv; else cat $O/pmc_$tag.csv > $O/hough_pmc.csv; fi
  rm -rf $O/pmc_$tag
done
python - <<PY
import csv, hashlib, json
rows = list(csv.DictReader(open("$O/hough_pmc.csv")))
out = {"_source": "rocprofv3 --pmc <set> --kernel-trace -- python tools/bench_ops.py --ops hough --iters 3 (three separate passes: SQ set, "
                  "FETCH_SIZE, WRITE_SIZE; tools/collect_pmc_hough.sh), B=16 640x480 C=22; per-dispatch averages",
       "_units": "FETCH_SIZE / WRITE_SIZE in KB per dispatch (rocprofv3); gfx950: FETCH_SIZE under-reports wide (16 B/lane) reads by 2x "
                 "(MI355X_MICROARCH.md HBM section) -> bench.py doubles it",
       "_kernel_source_sha16": hashlib.sha256(open("$R/posecnn_amd/csrc/hough_voting.hip", "rb").read()).hexdigest()[:16],
       "_kernel_source_note": "sha256(posecnn_amd/csrc/hough_voting.hip)[:16] of the build these counters were collected on; bench.py "
                              "reports \`traffic\` only while the file still hashes to this"}
for r in rows:
    k = r["kernel"].split("<")[0]
    name = r["counter"] + ("_KB" if r["counter"] in ("FETCH_SIZE", "WRITE_SIZE") else "")
    v = float(r["avg_per_dispatch"])
    out.setdefault(k, {})[name] = v if name.endswith("_KB") else int(round(v))
json.dump(out, open("$O/hough_pmc.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if not k.startswith("_")}, indent=1))
PY
