#!/bin/bash
# Kernel trace of the CD step at the mid-size layers of BASELINE configs 4 / 5 (B = 512):
#   gpurun -- 'bash scripts/experiments/mid_trace.sh r05g'
set -o pipefail
TAG=${1:-r05x}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
for SHAPE in 2048,400,512,5,1 1024,256,512,1,0 256,200,512,5,1; do
  export MDBN_AB_SHAPE=$SHAPE
  NAME=${SHAPE//,/_}
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_mid_$NAME -- python3 scripts/step_ab.py small_fused 1 > $OUT/${TAG}_mid_$NAME.log 2>&1 || exit 2
  STATS=$(find $OUT/${TAG}_mid_$NAME -name '*kernel_stats.csv' | head -1)
  cp "$STATS" $OUT/${TAG}_mid_${NAME}_kernel_stats.csv
  rm -rf $OUT/${TAG}_mid_$NAME
  grep -E "median|GEMM" $OUT/${TAG}_mid_$NAME.log
  python3 scripts/kernel_stats_print.py $OUT/${TAG}_mid_${NAME}_kernel_stats.csv
done
