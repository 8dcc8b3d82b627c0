#!/bin/bash
# build_variant_one.sh <name> <file.hip> <extra hipcc flags...>: like build_variant.sh, but only <file.hip> is recompiled with the extra
# flags; the other objects are the ones of the current build (cuda-akaze_amd/csrc/*.o).  -> build/ab/libhak_<name>.so
set -e
NAME=$1; FILE=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/cuda-akaze_amd/csrc
mkdir -p $R/build/ab
NOSLP=""
case $FILE in kernels_fed.hip|kernels_fedsf.hip|kernels_base_stream.hip|kernels_hessian_stream.hip) NOSLP="-fno-slp-vectorize";; esac
O=$(mktemp /tmp/hakvar_XXXX.o)
(cd $C && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result $NOSLP "$@" -c $FILE -o $O)
OBJS=""
for f in hak_api kernels_scalespace kernels_base kernels_base_stream kernels_fed kernels_fedsf kernels_level kernels_smoothflow kernels_hessian kernels_hessian_stream kernels_detect kernels_fast kernels_describe kernels_match; do
  if [ "$f.hip" = "$FILE" ]; then OBJS="$OBJS $O"; else OBJS="$OBJS $C/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $R/build/ab/libhak_$NAME.so $OBJS
rm -f $O
echo "built build/ab/libhak_$NAME.so"
