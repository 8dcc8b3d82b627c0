run() { local tag="$1"; shift; for i in 1 2; do python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-configs --no-verify "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$tag', d['value'], 'pairs/s;', ' '.join(f\"{c['class']} {c['ms']:.2f}\" for c in r['classes']), 'kp', d['config']['keypoints_per_image'])"; done; }
run "default"
run "upright" --upright
