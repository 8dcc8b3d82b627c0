#!/bin/bash
# Regenerates the evidence under profiles/ on the GPU box (run via gpurun from the repo root):
#   bash tools/refresh_profiles.sh r01_e
# Every profiler run is wrapped in `timeout`; PMC passes are separate runs without any trace option.
set -u
TAG=${1:-r02_x}
R=$(pwd)
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --upload --fast > $OUT/bench.json 2> $OUT/bench.err
B="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-verify"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/default -o d -- $B > $OUT/default.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial -o s -- $B --serial --no-pipeline > $OUT/serial.log 2>&1
P="python3 $R/bench.py --steps 1 --warmup 1 --serial --no-pipeline --no-cpu-baseline --no-roofline --no-configs --no-verify"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmcF -- $P > $OUT/pmcF.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmcW -- $P > $OUT/pmcW.log 2>&1
python3 $R/tools/pmc_traffic.py $OUT/pmcF $OUT/pmcW $OUT/pmc_traffic > $OUT/pmc_traffic.log 2>&1
find $OUT -name "*kernel_stats.csv" | head
tail -1 $OUT/bench.json | cut -c1-400
cat $OUT/pmc_traffic.log | cut -c1-400
