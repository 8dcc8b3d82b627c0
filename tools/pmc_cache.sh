#!/bin/bash
# Cache-path counters (TCP / TCC / TA) of the gather kernels, three separate --pmc passes without any trace option:
#   bash tools/pmc_cache.sh <tag>        (via gpurun, from the repo root)
set -u
TAG=${1:-cache}
R=$(pwd)
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P="python3 $R/bench.py --steps 1 --warmup 1 --serial --no-pipeline --no-cpu-baseline --no-roofline --no-configs --no-verify"
# at most 4 counters of one block per pass (TCC has 4 slots; a request the hardware cannot schedule aborts rocprofv3, which then
# sits until the timeout)
timeout 150 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_PENDING_STALL_CYCLES_sum \
    --output-format csv -d $OUT/A -- $P > $OUT/A.log 2>&1
timeout 150 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum \
    --output-format csv -d $OUT/B -- $P > $OUT/B.log 2>&1
timeout 150 rocprofv3 --pmc TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE \
    --output-format csv -d $OUT/C -- $P > $OUT/C.log 2>&1
for p in A B C; do python3 $R/tools/pmc_summary.py $OUT/$p > $OUT/$p.txt 2>&1; done
for p in A B C; do grep -c Kernel_Name -r $OUT/$p | tail -1; done
rm -rf $OUT/A $OUT/B $OUT/C
