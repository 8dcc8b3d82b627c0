R=$(pwd)
run() { local tag="$1"; shift; local best=""; for i in 1 2 3; do t=$(env "$@" $R/cuda-akaze_amd/hipakaze_demo 0 2>/dev/null | grep -m1 "Time of detection and computation" | awk '{print $NF}'); best="$best $t"; done; echo "$tag: $best"; }
run "default (2 side streams, 4 queues)" A=1
run "HAK_SIDE_STREAMS=3" HAK_SIDE_STREAMS=3
run "HAK_SIDE_STREAMS=1" HAK_SIDE_STREAMS=1
run "SIDE=3 Q=8" HAK_SIDE_STREAMS=3 GPU_MAX_HW_QUEUES=8
run "SIDE=2 Q=8" HAK_SIDE_STREAMS=2 GPU_MAX_HW_QUEUES=8
