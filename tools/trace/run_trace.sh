#!/bin/bash
# bash tools/trace/run_trace.sh hv|wm|bp [layers for wm]   (on the GPU box: needs hipcc + a GPU; leaves the tree as it found it)
set -e
R=$(cd $(dirname $0)/../.. && pwd); cd $R; mkdir -p gpurun_out
case $1 in
  hv) P=hough_voting; ;;
  wm) P=wino_mfma; ;;
  bp) P=backproject; ;;
  *) echo "usage: $0 hv|wm|bp"; exit 2;;
esac
cp posecnn_amd/csrc/$P.hip /tmp/$P.hip.keep; cp posecnn_amd/libposecnn_hip.so /tmp/libposecnn_hip.so.keep
restore() { cp /tmp/$P.hip.keep posecnn_amd/csrc/$P.hip; cp /tmp/libposecnn_hip.so.keep posecnn_amd/libposecnn_hip.so; touch posecnn_amd/csrc/$P.o 2>/dev/null || true; }
trap restore EXIT
git apply tools/trace/$P.trace.patch
make -C posecnn_amd/csrc -j8 > /dev/null
case $1 in
  hv) PCNN_HV_TRACE=6 python tools/bench_ops.py --ops hough --iters 3 > /dev/null 2>&1; python tools/trace/analyze_trace.py hv gpurun_out/hv_trace.bin ;;
  wm) for L in ${2:-conv2_1 conv3_2 conv4_2 conv5_1}; do PCNN_WM_TRACE=6 python tools/bench_wino_mfma.py --layers $L --no-library > /dev/null 2>&1; python tools/trace/analyze_trace.py wm gpurun_out/wm_trace.bin $L; done ;;
  bp) PCNN_BP_TRACE=3 python tools/bench_backproject.py --grids 256 --kinds smooth --iters 4 > /dev/null 2>&1; python tools/trace/analyze_trace.py bp gpurun_out/bp_trace.bin ;;
esac
