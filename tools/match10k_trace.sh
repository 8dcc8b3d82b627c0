#!/bin/bash
# match10k_trace.sh <tag>: rocprofv3 kernel trace + stats of tools/match10k.py (via gpurun, from the repo root)
TAG=$1; shift
R=$(pwd)
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
python3 $R/tools/match10k.py "$@" > $OUT/sync.json 2> $OUT/sync.err; cat $OUT/sync.json
python3 $R/tools/match10k.py --ctx "$@" > $OUT/sync_ctx.json 2>> $OUT/sync.err; cat $OUT/sync_ctx.json
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o m -- python3 $R/tools/match10k.py "$@" > $OUT/log.txt 2>&1
python3 - $OUT <<'P'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    n=r['Name'].replace('(anonymous namespace)::','').split('(')[0]
    print(f"{n[:60]:60s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:9.2f} us {float(r['TotalDurationNs'])/1e6:8.3f} ms")
# timeline of one call (last hak_match call = kernels between the two last k_match_finish)
t=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=sorted(csv.DictReader(open(t)), key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'].split('(')[0].replace('void ','').replace('(anonymous namespace)::','') for r in rows]
idx=[i for i,nm in enumerate(names) if nm.startswith('k_match_finish')]
if len(idx)>=2:
    a,b=idx[-2]+1,idx[-1]+1
    t0=int(rows[a]['Start_Timestamp'])
    print('one hak_match call:')
    for i in range(a,b):
        s,e=int(rows[i]['Start_Timestamp'])-t0,int(rows[i]['End_Timestamp'])-t0
        print(f"   {names[i][:50]:50s} start {s/1e3:8.2f} us  dur {(e-s)/1e3:8.2f} us  grid {rows[i].get('Grid_Size_X','?')}x{rows[i].get('Grid_Size_Y','?')} wg {rows[i].get('Workgroup_Size_X','?')}")
P
