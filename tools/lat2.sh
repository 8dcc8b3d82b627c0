R=$(pwd)
run() { local tag="$1"; shift; local best=""; for i in 1 2 3; do t=$(env "$@" $R/cuda-akaze_amd/hipakaze_demo 0 2>/dev/null | grep -m1 "Time of detection and computation" | awk '{print $NF}'); best="$best $t"; done; echo "$tag: $best"; }
run "NT=1024 (built)" A=1
run "NT=512" LD_PRELOAD=$R/build/ab/libhak_nt512.so
run "NT=256" LD_PRELOAD=$R/build/ab/libhak_nt256.so
run "NT=1024 (built)" A=1
