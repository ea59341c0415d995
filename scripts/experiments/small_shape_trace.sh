#!/bin/bash
# Kernel-trace of the CD step at the reference's own small shapes (c1 RBM 784->500 B=20; a 19 937-gene GRBM at B=20):
#   gpurun -- 'bash scripts/experiments/small_shape_trace.sh r03u'
set -o pipefail
TAG=${1:-r03x}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
for SHAPE in 784,500,20,1,0 19937,400,20,1,1 512,40,512,5,1; do
  export MDBN_AB_SHAPE=$SHAPE
  NAME=${SHAPE//,/_}
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_small_$NAME -- python3 scripts/step_ab.py gather_ahead 1 > $OUT/${TAG}_small_$NAME.log 2>&1 || exit 2
  STATS=$(find $OUT/${TAG}_small_$NAME -name '*kernel_stats.csv' | head -1)
  cp "$STATS" $OUT/${TAG}_small_${NAME}_kernel_stats.csv
  rm -rf $OUT/${TAG}_small_$NAME
  cat $OUT/${TAG}_small_$NAME.log | tail -3
  cut -c1-150 $OUT/${TAG}_small_${NAME}_kernel_stats.csv | head -14
done
