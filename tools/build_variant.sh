#!/bin/bash
# Builds a variant of libwtalign.so from sed-edited copies of csrc/ (for A/B runs through WT_LIBWTALIGN).
#   tools/build_variant.sh NAME 'sed-script for wt_dtw.hip' ['sed for wt_cost.hip' ['sed for wt_logmel.hip' ['sed for wt_logprob.hip' ['sed for wt_cost_core.h']]]]
# Output: tools/variants/libwtalign_NAME.so (git-ignored; travels to the GPU box).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; W=/tmp/wt_variant_$name; rm -rf $W; mkdir -p $W/pkg/csrc $W/include $R/tools/variants
cp $R/include/*.h $W/include/
cp $R/whisper-timestamped_amd/csrc/*.hip $R/whisper-timestamped_amd/csrc/*.h $R/whisper-timestamped_amd/csrc/Makefile $R/whisper-timestamped_amd/csrc/wtalign.map $W/pkg/csrc/
[ -n "$2" ] && sed -i -E "$2" $W/pkg/csrc/wt_dtw.hip
[ -n "$3" ] && sed -i -E "$3" $W/pkg/csrc/wt_cost.hip
[ -n "$4" ] && sed -i -E "$4" $W/pkg/csrc/wt_logmel.hip
[ -n "${5:-}" ] && sed -i -E "$5" $W/pkg/csrc/wt_logprob.hip
[ -n "${6:-}" ] && sed -i -E "$6" $W/pkg/csrc/wt_cost_core.h
make -s -C $W/pkg/csrc -j8 OUT=$R/tools/variants/libwtalign_$name.so
echo built tools/variants/libwtalign_$name.so
