#!/bin/bash
# isa_dump.sh <kernel-file.hip> <out.s> [extra flags]  -- device ISA of one csrc file with the library's flags
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
SRC=$1; OUT=$2; shift 2
cd $R/cuda-akaze_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math "$@" -S --cuda-device-only -o $OUT $SRC 2>&1 | grep -v "argument unused" || true
