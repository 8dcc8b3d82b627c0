# A/B of library builds on ONE box: serial per-class ms (bench roofline leg) and pipelined pairs/s.  usage: bench_ab.sh lib1.so lib2.so ...
for lib in "$@"; do
  for i in 1 2; do
    HAK_LIB=$PWD/$lib python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-configs --no-verify 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib', d['value'], 'pairs/s;', ' '.join(f\"{c['class']} {c['ms']:.2f}\" for c in r['classes']), 'copy', r['copy_ceiling_GBs'])"
  done
done
