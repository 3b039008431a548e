#!/bin/bash
# PMC counters of the kernels of ONE bench.py step (configs[2], single stream) of THIS build -> $O/step_pmc.json (copy to
# profiles/r06_step_pmc.json: bench.py's roofline.traffic / valu_frac / mfma_busy_share read it, gated per kernel by the
# hash of the source file the kernel lives in — tests/test_bench_launch.py fails when a gated source changes without a
# re-collection). Separate rocprofv3 --pmc passes with --kernel-trace only (MI355X_MICROARCH.md, HBM / rocprofv3 section).
#   bash tools/collect_pmc_step.sh gpurun_out/r6pmc        (on the GPU box)
O=${1:-gpurun_out/r6pmc}; mkdir -p $O; O=$(cd $O && pwd)
R=$(cd $(dirname $0)/.. && pwd)
cd /tmp && export TMPDIR=/tmp
: > $O/step_pmc.csv
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_WAIT_ANY" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o pmc -- python $R/bench.py --steps 3 --warmup 1 --prewarm-seconds 0 --streams 1 --no-cpu-baseline --no-secondary > $O/p$i.log 2>&1
  python $R/tools/pmc_summary.py $O/p$i > $O/p$i.csv 2>> $O/p$i.log
  if [ -s $O/step_pmc.csv ]; then tail -n +2 $O/p$i.csv >> $O/step_pmc.csv; else cat $O/p$i.csv > $O/step_pmc.csv; fi
  rm -rf $O/p$i
done
python - <<PY
import csv, hashlib, json, os
R = "$R"
sha = lambda f: hashlib.sha256(open(os.path.join(R, "posecnn_amd", "csrc", f), "rb").read()).hexdigest()[:16]
src_of = {"hv_": "hough_voting.hip", "backproject": "backproject.hip", "deconv_": "upscore.hip", "det_": "heads_small.hip", "pose_": "heads_small.hip", "fc_skinny": "fc_skinny.hip", "wino43_mfma": "wino_mfma.hip", "conv12_": "conv_first.hip", "wino43_input": "winograd.hip",
          "fc_rows": "fc_mfma.hip", "adl_": "average_distance.hip", "roi_pool": "roi_pool.hip", "upscore_": "upscore.hip",
          "hard_label": "hard_label.hip", "head_lowres": "heads_small.hip"}
out = {"_source": "rocprofv3 --pmc <set> --kernel-trace -- python bench.py --steps 3 --warmup 1 --prewarm-seconds 0 --streams 1 (five separate passes; "
                  "tools/collect_pmc_step.sh): configs[2] B=16 640x480 RGB-D C=22; per-dispatch averages over every launch of the kernel in the run",
       "_units": "FETCH_SIZE / WRITE_SIZE in KB per dispatch (rocprofv3); gfx950: FETCH_SIZE under-reports wide (16 B/lane) reads by 2x "
                 "(MI355X_MICROARCH.md HBM section) -> bench.py doubles it. SQ_* are chip-wide sums per dispatch.",
       "_source_sha16": {f: sha(f) for f in sorted(set(src_of.values()))}}
for r in csv.DictReader(open("$O/step_pmc.csv")):
    k = r["kernel"].replace("void ", "").split("<")[0].strip().strip('"')
    if not any(k.startswith(p) for p in src_of):
        continue
    name = r["counter"] + ("_KB" if r["counter"] in ("FETCH_SIZE", "WRITE_SIZE") else "")
    e = out.setdefault(k, {"_src": next(v for p, v in src_of.items() if k.startswith(p))})
    # several template instances of one kernel: dispatch-weighted mean
    n, v = int(r["dispatches"]), float(r["avg_per_dispatch"])
    if name in e:
        n0 = e["_n_" + name]
        e[name] = (e[name] * n0 + v * n) / (n0 + n); e["_n_" + name] = n0 + n
    else:
        e[name] = v; e["_n_" + name] = n
for k, e in out.items():
    if not k.startswith("_"):
        e["dispatches"] = max(v for kk, v in e.items() if kk.startswith("_n_"))
        for kk in [kk for kk in e if kk.startswith("_n_")]:
            del e[kk]
        for kk in e:
            if kk not in ("_src", "dispatches") and not kk.endswith("_KB"):
                e[kk] = int(round(e[kk]))
json.dump(out, open("$O/step_pmc.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k in ("hv_vote_kernel", "wino43_mfma_kernel", "conv12_wino43_fused_kernel")}, indent=1))
PY
