#!/bin/bash
# ab_libs.sh <lib> [<lib> ...]: per-class serial times and pairs/s of the bench with each build of the library (HAK_LIB), "-" = the
# built libhipakaze.so; two rounds (run via gpurun from the repo root)
export HAK_BENCH_PMC=0
for i in 1 2; do for v in "$@"; do
  if [ "$v" = "-" ]; then unset HAK_LIB; else export HAK_LIB=$PWD/cuda-akaze_amd/$v; fi
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-configs --no-upload --no-fast --no-verify > /tmp/ab_v.json 2>/dev/null
  python -c "
import json; d=json.load(open('/tmp/ab_v.json')); c={x['class']:x['ms'] for x in d['roofline']['classes']}
print('$v', d['value'], c)"
done; done
