#!/bin/bash
# How much of a streaming pass is its epilogue?  The same step with fused_epilogue = 0: the streaming kernel then only parks,
# reduces and stores the pre-activations (FUSED = 0) and act_epilogue_kernel runs as its own launch.
#   gpurun -- 'bash scripts/experiments/stream_unfused_trace.sh r05zh'
set -o pipefail
TAG=${1:-r05x}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
for SHAPE in 2048,400,512,1,1 1024,256,512,1,0; do
  export MDBN_AB_SHAPE=$SHAPE
  NAME=${SHAPE//,/_}
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_unf_$NAME -- python3 scripts/step_ab.py fused_epilogue 0 > $OUT/${TAG}_unf_$NAME.log 2>&1 || exit 2
  STATS=$(find $OUT/${TAG}_unf_$NAME -name '*kernel_stats.csv' | head -1)
  cp "$STATS" $OUT/${TAG}_unf_${NAME}_kernel_stats.csv
  rm -rf $OUT/${TAG}_unf_$NAME
  echo "== $SHAPE (fused_epilogue = 0)"
  python3 scripts/kernel_stats_print.py $OUT/${TAG}_unf_${NAME}_kernel_stats.csv
done
