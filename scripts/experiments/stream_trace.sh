#!/bin/bash
# Kernel trace of the CD step at the mid-size layers (B = 512) with the streaming bf16x6 kernel (csrc/mdbn_stream.hip):
#   gpurun -- 'bash scripts/experiments/stream_trace.sh r05t 2'
set -o pipefail
TAG=${1:-r05x}
MODE=${2:-1}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
for SHAPE in 2048,400,512,5,1 1024,256,512,1,0 256,200,512,5,1 784,500,512,1,0; do
  export MDBN_AB_SHAPE=$SHAPE
  NAME=${SHAPE//,/_}
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stream_$NAME -- python3 scripts/step_ab.py stream_x6 $MODE > $OUT/${TAG}_stream_$NAME.log 2>&1 || exit 2
  STATS=$(find $OUT/${TAG}_stream_$NAME -name '*kernel_stats.csv' | head -1)
  cp "$STATS" $OUT/${TAG}_stream_${NAME}_kernel_stats.csv
  rm -rf $OUT/${TAG}_stream_$NAME
  echo "== $SHAPE"
  grep -E "median" $OUT/${TAG}_stream_$NAME.log
  python3 scripts/kernel_stats_print.py $OUT/${TAG}_stream_${NAME}_kernel_stats.csv
done
