#!/bin/bash
# counters of k_match_mfma in one serial bench sequence (run via gpurun from the repo root)
R=$(pwd); OUT=$R/gpurun_out/mm_pmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P="python3 $R/bench.py --steps 1 --warmup 1 --serial --no-pipeline --no-cpu-baseline --no-roofline --no-configs --no-verify --no-upload --no-fast"
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/a -- $P > $OUT/a.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/b -- $P > $OUT/b.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o t -- $P > $OUT/t.log 2>&1
python3 - $OUT <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
for sub in ('a','b'):
    for f in glob.glob(out+'/'+sub+'/**/*counter_collection.csv',recursive=True):
        acc=collections.defaultdict(float); n=0
        for r in csv.DictReader(open(f)):
            if 'k_match_mfma' in r['Kernel_Name']:
                acc[r['Counter_Name']]+=float(r['Counter_Value'])
        print(sub, dict(acc))
for f in glob.glob(out+'/t/**/*kernel_stats.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'match' in r['Name']: print(r['Name'][:40], r['Calls'], r['AverageNs'])
PY
find $OUT -name "*.csv" -size +1M -delete
