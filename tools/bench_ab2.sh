run() { local tag="$1"; shift; local v=""; for i in 1 2; do v="$v $(env "$@" python3 bench.py --width 1280 --height 720 --pairs 64 --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --no-configs --no-verify 2>/dev/null | tail -1 | python3 -c 'import sys,json; print(json.loads(sys.stdin.read())["value"])')"; done; echo "$tag: $v"; }
run "720p default" A=1
run "720p Q=4" GPU_MAX_HW_QUEUES=4
run "720p Q=5" GPU_MAX_HW_QUEUES=5
run "720p HAK_GRAPH=0" HAK_GRAPH=0
run1080() { local tag="$1"; shift; local v=""; for i in 1 2; do v="$v $(env "$@" python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-configs --no-verify 2>/dev/null | tail -1 | python3 -c 'import sys,json; print(json.loads(sys.stdin.read())["value"])')"; done; echo "$tag: $v"; }
run1080 "1080p default" A=1
run1080 "1080p Q=4" GPU_MAX_HW_QUEUES=4
run1080 "1080p Q=5" GPU_MAX_HW_QUEUES=5
