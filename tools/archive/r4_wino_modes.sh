# HISTORICAL (round 4): the kernel variants these switches selected (PCNN_WINO_MODE bits 1 / 3, PCNN_CONV12=2) left the library in round 5 -> tools/variants/
# Round-4 experiment: wino43_mfma_kernel variants (PCNN_WINO_MODE: 8 = round-3 kernel, 0 = zero-C first MFMAs, bit 0 = channel-block-major
# map where it applies; the priority and persistent-grid variants measured with this script are recorded in csrc/wino_mfma.hip). bash tools/r4_wino_modes.sh <outdir>
O=${1:-gpurun_out/r4c}; mkdir -p $O
for m in 8 0; do
  PCNN_WINO_MODE=$m python tools/bench_wino_mfma.py --no-library > $O/layers_mode$m.json 2>> $O/err.log
done
for m in 0 1; do
  PCNN_WINO_MODE=$m python tools/bench_wino_mfma.py --no-library --batch 1 --groups 1 > $O/layers_b1_mode$m.json 2>> $O/err.log
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/layers_*mode*.json")):
    d = json.load(open(f))
    print(os.path.basename(f), "total mfma ms", d["total"]["mfma_ms"], "TF", d["total"]["mfma_TFLOPs"], " ".join("%s %.3f" % (k, v["mfma_ms"]) for k, v in d["layers"].items()))
PY
