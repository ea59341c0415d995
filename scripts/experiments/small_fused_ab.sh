#!/bin/bash
# One-launch step (mdbn_small.hip) against the multi-launch path at the LDS-resident layer shapes, interleaved A/B of the
# "small_fused" option on the whole step function, then a kernel trace of the fused path:
#   gpurun -- 'bash scripts/experiments/small_fused_ab.sh r04d'
set -o pipefail
TAG=${1:-r04x}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
: > $OUT/${TAG}_small_fused_ab.log
for SHAPE in 512,40,512,5,1 400,40,512,1,0 200,20,512,1,0 100,128,512,1,0 100,24,20,1,0; do
  export MDBN_AB_SHAPE=$SHAPE
  echo "== shape V,H,B,k,gauss = $SHAPE" >> $OUT/${TAG}_small_fused_ab.log
  python3 scripts/step_ab.py small_fused 0 1 2>/dev/null | grep median >> $OUT/${TAG}_small_fused_ab.log || exit 2
done
cat $OUT/${TAG}_small_fused_ab.log
for SHAPE in 512,40,512,5,1 100,128,512,1,0; do
  export MDBN_AB_SHAPE=$SHAPE
  NAME=${SHAPE//,/_}
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_small_$NAME -- python3 scripts/step_ab.py small_fused 1 > $OUT/${TAG}_small_$NAME.log 2>&1 || exit 3
  STATS=$(find $OUT/${TAG}_small_$NAME -name '*kernel_stats.csv' | head -1)
  cp "$STATS" $OUT/${TAG}_small_${NAME}_kernel_stats.csv
  rm -rf $OUT/${TAG}_small_$NAME
  cut -c1-150 $OUT/${TAG}_small_${NAME}_kernel_stats.csv | head -8
done
