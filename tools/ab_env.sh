#!/bin/bash
# ab_env.sh "VAR=val ..." ["VAR=val ..." ...]: the per-class serial times and pairs/s of the bench under each environment (run via gpurun)
export HAK_BENCH_PMC=0      # no rocprofv3 --pmc child runs inside an A/B (they are slow, and the variables under test would leak into them)
for e in "$@"; do
  env $e python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-configs --no-upload --no-fast --no-verify > /tmp/ab_env.json 2>/dev/null
  python -c "
import json; d=json.load(open('/tmp/ab_env.json')); c={x['class']:x['ms'] for x in d['roofline']['classes']}
print('$e', d['value'], d['roofline']['frac'], d['roofline']['launches_per_step'], c)"
done
