# serial per-class ms under env variants on ONE box
run() { local tag="$1"; shift; for i in 1 2; do env "$@" python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-configs --no-verify 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$tag', d['value'], 'pairs/s;', ' '.join(f\"{c['class']} {c['ms']:.2f}\" for c in r['classes']))"; done; }
run "waves 4096" HAK_STREAM_MIN_WAVES=4096
run "waves 3072" HAK_STREAM_MIN_WAVES=3072
run "waves 2048" HAK_STREAM_MIN_WAVES=2048
run "waves 1536" HAK_STREAM_MIN_WAVES=1536
run "waves 8192" HAK_STREAM_MIN_WAVES=8192
