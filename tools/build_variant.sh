#!/bin/bash
# build_variant.sh <name> <extra hipcc flags...>: builds libhipakaze with extra flags into build/ab/libhak_<name>.so (A/B runs:
# HAK_LIB=build/ab/libhak_<name>.so python bench.py ..., or LD_PRELOAD for the C++ demo).  build/ is git-ignored but travels to the GPU box.
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
mkdir -p $T/cuda-akaze_amd $T/include $R/build/ab
cp -r $R/cuda-akaze_amd/csrc $T/cuda-akaze_amd/
cp $R/include/*.h $T/include/
rm -f $T/cuda-akaze_amd/csrc/*.o
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result"
make -C $T/cuda-akaze_amd/csrc -j8 FLAGS="$BASE $*" > $T/log.txt 2>&1 || { tail -20 $T/log.txt; exit 1; }
cp $T/cuda-akaze_amd/libhipakaze.so $R/build/ab/libhak_$NAME.so
rm -rf $T
echo "built build/ab/libhak_$NAME.so"
