# HISTORICAL (round 4): the kernel variants these switches selected (PCNN_WINO_MODE bits 1 / 3, PCNN_CONV12=2) left the library in round 5 -> tools/variants/
# Round-4 experiment: the one-wave-per-SIMD trunk kernel (wino43_mfma_w1_kernel, PCNN_WINO_MODE=2) against the 32-tile pairs
# (the library's choice) — alone (tools/wino_w1_probe, tools/mfma_bare) and in the whole step on 1 / 2 / 3 streams and at the
# LINEMOD configuration. bash tools/r4_w1_ab.sh <outdir>; what it printed for round 4 is profiles/r04_w1_pipeline_ab.txt,
# profiles/r04_wino_w1_probe.txt, profiles/r04_mfma_bare.txt.
O=${1:-gpurun_out/w1}; mkdir -p $O
tools/mfma_bare > $O/mfma_bare.txt 2>&1
tools/wino_w1_probe > $O/wino_w1_probe.txt 2>&1
for st in 1 2 3; do
  PCNN_WINO_MODE=2 python bench.py --streams $st --no-cpu-baseline --no-secondary > $O/w1_s$st.json 2>> $O/bench.err
  python bench.py --streams $st --no-cpu-baseline --no-secondary > $O/pairs_s$st.json 2>> $O/bench.err
done
PCNN_WINO_MODE=2 python bench.py --config linemod --no-cpu-baseline --no-secondary > $O/w1_linemod.json 2>> $O/bench.err
python bench.py --config linemod --no-cpu-baseline --no-secondary > $O/pairs_linemod.json 2>> $O/bench.err
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*_s?.json") + glob.glob("$O/*_linemod.json")):
    d = json.load(open(f))
    print(os.path.basename(f), round(d["value"], 1), round(d["ms_per_step"], 3), {k: v for k, v in d["kernels_us"].items() if "wino43_mfma" in k})
PY
