#!/bin/bash
# A ragged hidden width on a padded leading dimension (plane path) against the exact-f32 tile kernels it otherwise falls to:
#   gpurun -- 'bash scripts/experiments/ragged_planes_ab.sh r04zg'
TAG=${1:-r04x}
OUT=gpurun_out/${TAG}_ragged_planes_ab.log
: > $OUT
for SHAPE in 2048,400,512,5,1 2048,400,512,1,1 1024,400,512,1,0 2048,200,512,1,1 4096,1000,512,1,1; do
  echo "== shape V,H,B,k,gauss = $SHAPE: dense ld (exact-f32 tiles)" >> $OUT
  MDBN_WEIGHT_LD_MIN=999999999999 MDBN_AB_SHAPE=$SHAPE python3 scripts/step_ab.py planes_min_work 0 2>/dev/null | grep -E "median|GEMM" >> $OUT
  echo "== shape V,H,B,k,gauss = $SHAPE: ld padded to 128 (planes_min_work 0 = always planes / 1073741824 = the default rule)" >> $OUT
  MDBN_WEIGHT_LD_MIN=0 MDBN_AB_SHAPE=$SHAPE python3 scripts/step_ab.py planes_min_work 0 1073741824 2>/dev/null | grep -E "median|GEMM" >> $OUT
done
cat $OUT
