R=$(pwd)
run() { local tag="$1"; shift; local best=""; for i in 1 2 3; do t=$(env "$@" $R/cuda-akaze_amd/hipakaze_demo 0 2>/dev/null | grep -m1 "Time of detection and computation" | awk '{print $NF}'); best="$best $t"; done; echo "$tag: $best"; }
run "default (min steps 8)" A=1
run "min steps 5" HAK_LEVEL_MIN_STEPS=5
run "min steps 6" HAK_LEVEL_MIN_STEPS=6
run "min steps 3" HAK_LEVEL_MIN_STEPS=3
run "level hess off" HAK_LEVEL_HESS=0
