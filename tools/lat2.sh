R=$(pwd)
run() { local tag="$1"; shift; local best=""; for i in 1 2 3; do t=$(env "$@" $R/cuda-akaze_amd/hipakaze_demo 0 2>/dev/null | grep -m1 "Time of detection and computation" | awk '{print $NF}'); best="$best $t"; done; echo "$tag: $best"; }
run "default (Q=8 hint)" A=1
run "Q=4" GPU_MAX_HW_QUEUES=4
run "Q=5" GPU_MAX_HW_QUEUES=5
run "Q=6" GPU_MAX_HW_QUEUES=6
run "Q=16" GPU_MAX_HW_QUEUES=16
run "default again" A=1
