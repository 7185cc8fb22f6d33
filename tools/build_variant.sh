#!/bin/bash
# Builds a variant of libwtalign.so from sed-edited copies of csrc/ (for A/B runs through WT_LIBWTALIGN).
#   tools/build_variant.sh NAME 'sed-script for wt_dtw.hip' ['sed-script for wt_cost.hip' ['sed-script for wt_logmel.hip']]
# Output: tools/variants/libwtalign_NAME.so (git-ignored; travels to the GPU box).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; W=/tmp/wt_variant_$name; rm -rf $W; mkdir -p $W/pkg/csrc $W/include $R/tools/variants
cp $R/include/*.h $W/include/
cp $R/whisper-timestamped_amd/csrc/*.hip $R/whisper-timestamped_amd/csrc/*.h $R/whisper-timestamped_amd/csrc/Makefile $R/whisper-timestamped_amd/csrc/wtalign.map $W/pkg/csrc/
[ -n "$2" ] && sed -i -E "$2" $W/pkg/csrc/wt_dtw.hip
[ -n "$3" ] && sed -i -E "$3" $W/pkg/csrc/wt_cost.hip
[ -n "$4" ] && sed -i -E "$4" $W/pkg/csrc/wt_logmel.hip
make -s -C $W/pkg/csrc -j8 OUT=$R/tools/variants/libwtalign_$name.so
echo built tools/variants/libwtalign_$name.so
