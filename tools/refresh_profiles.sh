#!/bin/bash
# Regenerates the evidence under profiles/ on the GPU box (run via gpurun from the repo root):
#   bash tools/refresh_profiles.sh r05_a
# Every profiler run is wrapped in `timeout`; PMC passes are separate runs without any trace option.
set -u
TAG=${1:-r05_x}
R=$(pwd)
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
B="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-verify --no-upload --no-fast"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/default -o d -- $B > $OUT/default.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial -o s -- $B --serial --no-pipeline > $OUT/serial.log 2>&1
P="python3 $R/bench.py --steps 1 --warmup 1 --serial --no-pipeline --no-cpu-baseline --no-roofline --no-configs --no-verify --no-upload --no-fast"
# (the PMC passes need the roofline leg off -- its copy probe would be counted -- so the bench line of these runs carries no roofline)
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmcF -- $P > $OUT/pmcF.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmcW -- $P > $OUT/pmcW.log 2>&1
python3 $R/tools/pmc_traffic.py $OUT/pmcF $OUT/pmcW $OUT/pmc_traffic $OUT/pmcF.log > $OUT/pmc_traffic.log 2>&1
# SQ counters (two passes of 8): wave cycles / waits / active instruction classes, then instruction counts
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT \
    --output-format csv -d $OUT/sqA -- $P > $OUT/sqA.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES \
    --output-format csv -d $OUT/sqB -- $P > $OUT/sqB.log 2>&1
python3 $R/tools/pmc_summary.py --table $OUT/sqA $OUT/sqB > $OUT/sq_counters.txt 2>&1
# cache-path counters of the gather kernels (TCP / TCC / TA): at most 4 counters of one block per pass
timeout 150 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_PENDING_STALL_CYCLES_sum \
    --output-format csv -d $OUT/cA -- $P > $OUT/cA.log 2>&1
timeout 150 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum --output-format csv -d $OUT/cB -- $P > $OUT/cB.log 2>&1
timeout 150 rocprofv3 --pmc TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/cC -- $P > $OUT/cC.log 2>&1
python3 $R/tools/pmc_summary.py --cache $OUT/cA $OUT/cB $OUT/cC > $OUT/cache_counters.txt 2>&1
rm -rf $OUT/cA $OUT/cB $OUT/cC
find $OUT -name "*kernel_stats.csv" | head
tail -1 $OUT/bench.json | cut -c1-400
cat $OUT/pmc_traffic.log | cut -c1-600
# keep the merged-back payload small: the raw per-dispatch counter dumps are summarised above
rm -rf $OUT/pmcF $OUT/pmcW $OUT/sqA $OUT/sqB
find $OUT -name "*kernel_trace.csv" -delete
