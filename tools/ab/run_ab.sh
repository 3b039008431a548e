# bash tools/ab/run_ab.sh '<command>'  — runs the command with tools/ab/libA.so and libB.so in turn as the product library (same box)
for v in A B A B; do
  cp tools/ab/lib$v.so posecnn_amd/libposecnn_hip.so
  echo "=== lib$v"
  eval "$1"
done
