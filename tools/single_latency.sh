#!/bin/bash
# single_latency.sh: the C++ drop-in demo (one image per call, main.cpp:199-209's pattern) unprofiled, under the variants that matter
# for a launch-bound call.  Prints "ms per pair (two detectAndCompute calls)" of the float path, three runs per variant.
# (the demo itself asks for 8 hardware queues unless GPU_MAX_HW_QUEUES is set)
R=$(pwd)
run() {
  local tag="$1"; shift
  local best=""
  for i in 1 2 3; do
    t=$(env "$@" $R/cuda-akaze_amd/hipakaze_demo 0 2>/dev/null | grep -m1 "Time of detection and computation" | awk '{print $NF}')
    best="$best $t"
  done
  echo "$tag: $best"
}
run "default                              " A=1
run "GPU_MAX_HW_QUEUES=4 (runtime default) " GPU_MAX_HW_QUEUES=4
run "HAK_SIDE_STREAMS=1                   " HAK_SIDE_STREAMS=1
run "HAK_LEVEL_HESS=0                     " HAK_LEVEL_HESS=0
run "HAK_LEVEL_MIN_STEPS=5                " HAK_LEVEL_MIN_STEPS=5
run "HAK_GRAPH=2 (graph replay)           " HAK_GRAPH=2
run "HAK_LEVEL_TILE=0 (round-2 kernels)   " HAK_LEVEL_TILE=0
run "HAK_SERIAL=1                         " HAK_SERIAL=1
