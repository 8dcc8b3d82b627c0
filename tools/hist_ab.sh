for e in "HAK_HIST_MIN_BLOCKS=256" "HAK_HIST_MIN_BLOCKS=256 HAK_HIST_RPB_MAX=16" "HAK_HIST_MIN_BLOCKS=128 HAK_HIST_RPB_MAX=16" "HAK_HIST_MIN_BLOCKS=64 HAK_HIST_RPB_MAX=32" "HAK_HIST_MIN_BLOCKS=2048"; do echo "== $e"; env $e python bench.py --pair-call-leg 2>/dev/null | tail -1 | cut -c1-40; done
export HAK_BENCH_PMC=0
for e in "HAK_HIST_MIN_BLOCKS=2048" "HAK_HIST_MIN_BLOCKS=256" "HAK_HIST_MIN_BLOCKS=256 HAK_HIST_RPB_MAX=16" "HAK_HIST_MIN_BLOCKS=256 HAK_HIST_RPB_MAX=32" "HAK_HIST_MIN_BLOCKS=2048"; do
  env $e python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-configs --no-upload --no-fast --no-verify > /tmp/b.json 2>/dev/null
  python -c "
import json; d=json.load(open('/tmp/b.json')); print('$e', d['value'], {c['class']: c['ms'] for c in d['roofline']['classes']})"
done
