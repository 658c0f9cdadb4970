#!/bin/bash
# Whole-forward A/B of ONE compile-time switch with PRODUCT builds (a diagnostic build reads its switches from the environment on every launch and is host-bound at
# 256 frames: 11.5 ms per step against 8.8): copies csrc/ to a scratch directory, rewrites the default of GRNET_AB(NAME, ...) to VALUE, builds
# video-based-gait-analysis-for-dementia_amd/libgrnet_hip_NAME_VALUE.so (git-ignored; travels to the GPU box), to be loaded with GRNET_LIB_PATH.
#   tools/ab_build.sh BF16_S2_ROWS 0
set -e
NAME=$1; VALUE=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/video-based-gait-analysis-for-dementia_amd/csrc
TMP=$(mktemp -d /tmp/ab_${NAME}_XXXX)
mkdir -p $TMP/pkg/csrc $TMP/include
cp $ROOT/include/*.h $TMP/include/
C=$TMP/pkg/csrc
cp $SRC/*.hip $SRC/*.cpp $SRC/*.h $SRC/Makefile $C/
n=$(grep -l "GRNET_AB($NAME," $C/*.hip $C/*.cpp | wc -l)
[ "$n" -ge 1 ] || { echo "no GRNET_AB($NAME, ...) in csrc"; exit 1; }
sed -i -E "s/GRNET_AB\($NAME, *[^)]*\)/GRNET_AB($NAME, $VALUE)/g" $C/*.hip $C/*.cpp
make -C $C -j8 BUILD=$TMP/build LIB=$ROOT/video-based-gait-analysis-for-dementia_amd/libgrnet_hip_${NAME}_${VALUE}.so > $TMP/make.log 2>&1 || { tail -5 $TMP/make.log; exit 1; }
ls -la $ROOT/video-based-gait-analysis-for-dementia_amd/libgrnet_hip_${NAME}_${VALUE}.so
rm -rf $TMP
