# A/B of the pipelined headline under env variants on ONE box: pairs/s (3 runs each)
run() { local tag="$1"; shift; local v=""; for i in 1 2; do v="$v $(env "$@" python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-configs --no-verify 2>/dev/null | tail -1 | python3 -c 'import sys,json; print(json.loads(sys.stdin.read())["value"])')"; done; echo "$tag: $v"; }
run "default" A=1
run "HAK_TAIL_FORK=0" HAK_TAIL_FORK=0
run "Q=4" GPU_MAX_HW_QUEUES=4
run "Q=4 TAIL_FORK=0" GPU_MAX_HW_QUEUES=4 HAK_TAIL_FORK=0
run "HAK_GRAPH=0" HAK_GRAPH=0
