#!/bin/bash
# pair_env.sh: pair-call latency (bench.py --pair-call-leg) and the three-call pattern under kernel-selection / scheduling environments (run via gpurun)
for e in "X=0" "HAK_LEVEL_MIN_BLOCKS=64" "HAK_LEVEL_MIN_BLOCKS=48" "HAK_LEVEL_MIN_BLOCKS=32" "HAK_LEVEL_MIN_BLOCKS=24" "HAK_LEVEL_MIN_BLOCKS=16" "HAK_LEVEL_MIN_BLOCKS=1" "HAK_LEVEL_MIN_BLOCKS=128" "HAK_LEVEL_MIN_BLOCKS=192" "X=1"; do
  echo "== $e"; env $e python bench.py --pair-call-leg 2>/dev/null | tail -1 | cut -c1-40; env $e python bench.py --single-pair-leg 2>/dev/null | tail -1 | cut -c1-60
done
