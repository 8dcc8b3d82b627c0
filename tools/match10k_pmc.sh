#!/bin/bash
# match10k_pmc.sh <tag>: SQ counters of k_match_mfma in the 10k x 10k call (run via gpurun from the repo root)
TAG=${1:-match10k_pmc}
R=$(pwd); OUT=$R/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P="python3 $R/tools/match10k.py --ctx --iters 10"
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/a -- $P > $OUT/a.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/b -- $P > $OUT/b.log 2>&1
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_I8 SQ_WAIT_INST_ANY SQ_LDS_ADDR_CONFLICT --output-format csv -d $OUT/c -- $P > $OUT/c.log 2>&1
python3 - $OUT <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
for sub in ('a','b','c'):
    for f in glob.glob(out+'/'+sub+'/**/*counter_collection.csv',recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
        for r in csv.DictReader(open(f)):
            k='knn' if 'ILb1' in r['Kernel_Name'] or '<true>' in r['Kernel_Name'] else '1nn' if 'k_match_mfma' in r['Kernel_Name'] else None
            if k: acc[k][r['Counter_Name']]+=float(r['Counter_Value']); n[(k,r['Counter_Name'])]+=1
        for k,v in acc.items():
            print(sub,k,{c: round(x/n[(k,c)]) for c,x in v.items()})
PY
tail -3 $OUT/c.log
find $OUT -name "*.csv" -size +1M -delete
