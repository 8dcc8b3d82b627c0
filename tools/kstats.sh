#!/bin/bash
# kstats.sh <tag> [bench args]: rocprofv3 kernel stats of one serial bench run (via gpurun, from the repo root)
TAG=$1; shift
R=$(pwd)
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-verify --no-upload --no-fast --serial --no-pipeline "$@" > $OUT/log.txt 2>&1
[ -n "$KEEP_TRACE" ] || find $OUT -name "*kernel_trace.csv" -delete
python3 - $OUT <<'P'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:26]:
    n=r['Name'].replace('(anonymous namespace)::','').split('(')[0]
    print(f"{n[:44]:44s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:9.1f} us {float(r['TotalDurationNs'])/1e6:8.2f} ms")
P
