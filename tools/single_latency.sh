#!/bin/bash
# single_latency.sh: the C++ drop-in demo (one image per call, main.cpp:199-209's pattern) unprofiled, under the kernel-selection
# variants that matter for a launch-bound call.  Prints "ms per pair (two detectAndCompute calls)" of the float path per variant.
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
run "default                         "
run "HAK_GRAPH=2 (replay)            " HAK_GRAPH=2
run "GPU_MAX_HW_QUEUES=8             " GPU_MAX_HW_QUEUES=8
run "HAK_LEVEL_MIN_STEPS=1           " HAK_LEVEL_MIN_STEPS=1
run "HAK_LEVEL_MIN_STEPS=5           " HAK_LEVEL_MIN_STEPS=5
run "HAK_LEVEL_MIN_STEPS=12          " HAK_LEVEL_MIN_STEPS=12
run "HAK_LEVEL_TILE=0 (round-2 path) " HAK_LEVEL_TILE=0
run "HAK_SERIAL=1                    " HAK_SERIAL=1
run "HAK_GRAPH=0 (eager launches)    " HAK_GRAPH=0
run "HAK_GRAPH=0 HAK_LEVEL_TILE=0    " HAK_GRAPH=0 HAK_LEVEL_TILE=0
