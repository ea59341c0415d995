#!/bin/bash
# Kernel trace of the thin-batch CD step (csrc/mdbn_thin.hip) at the reference's own batch-20 shapes:
#   gpurun -- 'bash scripts/experiments/thin_trace.sh r05c'
set -o pipefail
TAG=${1:-r05x}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
for SHAPE in 784,500,20,1,0 19937,400,20,1,1; do
  export MDBN_AB_SHAPE=$SHAPE
  NAME=${SHAPE//,/_}
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_thin_$NAME -- python3 scripts/step_ab.py thin_fused 1 > $OUT/${TAG}_thin_$NAME.log 2>&1 || exit 2
  STATS=$(find $OUT/${TAG}_thin_$NAME -name '*kernel_stats.csv' | head -1)
  cp "$STATS" $OUT/${TAG}_thin_${NAME}_kernel_stats.csv
  rm -rf $OUT/${TAG}_thin_$NAME
  tail -3 $OUT/${TAG}_thin_$NAME.log
  python3 scripts/kernel_stats_print.py $OUT/${TAG}_thin_${NAME}_kernel_stats.csv
done
