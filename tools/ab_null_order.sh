for e in "HAK_NULL_ORDER=1" "HAK_NULL_ORDER=0" "HAK_NULL_ORDER=1" "HAK_NULL_ORDER=0"; do
  echo "== $e"; env $e python bench.py --pair-call-leg 2>/dev/null | tail -1 | cut -c1-60; env $e python bench.py --single-pair-leg 2>/dev/null | tail -1 | cut -c1-80
done
export HAK_BENCH_PMC=0
for e in "HAK_NULL_ORDER=1" "HAK_NULL_ORDER=0" "HAK_NULL_ORDER=1" "HAK_NULL_ORDER=0"; do
  env $e python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-configs --no-upload --no-fast --no-verify --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e', d['value'])"
done
