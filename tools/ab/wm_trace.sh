# per-workgroup trace of wino43_mfma_kernel for a few layers (debug library tools/ab/libT.so)
cp posecnn_amd/libposecnn_hip.so /tmp/lib_keep.so
cp tools/ab/libT.so posecnn_amd/libposecnn_hip.so
for L in conv2_1 conv3_2 conv4_2 conv5_1; do
  PCNN_WM_TRACE=6 python tools/bench_wino_mfma.py --layers $L --no-library 2>&1 | tail -2 | cut -c1-400
  mv gpurun_out/wm_trace.bin gpurun_out/wm_trace_$L.bin
done
cp /tmp/lib_keep.so posecnn_amd/libposecnn_hip.so
