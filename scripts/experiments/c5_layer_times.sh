#!/bin/bash
# Step time of every layer of BASELINE configs[4] (scripts/bench_mdbn.py) on one GPU, one by one (product defaults):
#   gpurun -- 'bash scripts/experiments/c5_layer_times.sh r04zz'
TAG=${1:-r04x}
OUT=gpurun_out/${TAG}_c5_layer_times.log
: > $OUT
for SHAPE in 2048,400,512,5,1 400,40,512,5,0 512,40,512,5,1 256,200,512,5,1 200,20,512,5,0 100,128,512,1,0 128,3,512,1,0; do
  echo "== shape V,H,B,k,gauss = $SHAPE" >> $OUT
  MDBN_AB_SHAPE=$SHAPE python3 scripts/step_ab.py small_fused 1 2>/dev/null | grep -E "median|GEMM" >> $OUT
done
cat $OUT
