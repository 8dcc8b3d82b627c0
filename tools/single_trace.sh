#!/bin/bash
# single_trace.sh <tag> [demo args]: rocprofv3 kernel trace + stats of the C++ drop-in demo (one image per call, the call pattern
# of the reference's main.cpp:199-209), then the timeline of ONE detectAndCompute call (tools/single_timeline.py).
# Run via gpurun from the repo root.  The program itself follows `--` (no shell / env hop under the profiler).
TAG=$1; shift
R=$(pwd)
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
[ -n "$HAK_TRACE_ENV" ] && export $HAK_TRACE_ENV
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- $R/cuda-akaze_amd/hipakaze_demo 0 "$@" > $OUT/log.txt 2>&1
tail -12 $OUT/log.txt
python3 $R/tools/single_timeline.py $OUT | tee $OUT/timeline.txt
