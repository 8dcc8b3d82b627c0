#!/bin/bash
# gpu_idle.sh <tag>: kernel trace of one default (pipelined) bench run; prints how much of the timed region no kernel was running
TAG=$1; shift
R=$(pwd)
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-configs --no-verify --no-roofline "$@" > $OUT/log.txt 2>&1
python3 - $OUT <<'P'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
iv = []
for r in csv.DictReader(open(f)):
    iv.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
iv.sort()
# timed region: from the first k_base_stream after 2/8 of the launches to the end
n = len(iv)
t0, t1 = iv[n // 4][0], iv[-1][1]
cur_s, cur_e, busy = None, None, 0
gaps = []
for s, e, k in iv:
    if e < t0: continue
    s = max(s, t0)
    if cur_e is None: cur_s, cur_e = s, e; continue
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, k)); cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
wall = t1 - t0
print(f"wall {wall/1e6:.2f} ms  busy {busy/1e6:.2f} ms  idle {100*(1-busy/wall):.2f} %  gaps {len(gaps)}")
gaps.sort(reverse=True)
for g, k in gaps[:12]: print(f"  gap {g/1e3:8.1f} us before {k[:60]}")
P
find $OUT -name "*kernel_trace.csv" -delete
