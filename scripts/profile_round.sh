#!/bin/bash
# Evidence for one build: bench line, rocprofv3 kernel-trace stats, PMC passes (HBM traffic, MFMA busy), summaries.
#   gpurun -- 'bash scripts/profile_round.sh r02d'      (writes gpurun_out/<tag>_*; copy what is to be judged into profiles/)
# rocprofv3 is given the python3 binary directly (no env / bash -c hop in front of it).
set -o pipefail
TAG=${1:-r02x}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
# the profiled command runs the DEFAULT path only (no exact-f32 / bf16-input re-runs): per-step sums are then plain sums
CMD="python3 bench.py --steps 100 --warmup 10 --windows 5 --default-only"
python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- $CMD > $OUT/${TAG}_stats.log 2>&1 || exit 2
STATS=$(find $OUT/${TAG}_stats -name '*kernel_stats.csv' | head -1)
cp "$STATS" $OUT/${TAG}_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch -- $CMD > $OUT/${TAG}_fetch.log 2>&1 || exit 3
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write -- $CMD > $OUT/${TAG}_write.log 2>&1 || exit 4
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/${TAG}_mfma -- $CMD > $OUT/${TAG}_mfma.log 2>&1 || exit 5
F=$(find $OUT/${TAG}_fetch -name '*counter_collection.csv' | head -1)
W=$(find $OUT/${TAG}_write -name '*counter_collection.csv' | head -1)
M=$(find $OUT/${TAG}_mfma -name '*counter_collection.csv' | head -1)
python3 scripts/pmc_traffic.py "$F" "$W" $OUT/${TAG}_pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- $CMD" > /dev/null || exit 6
python3 scripts/pmc_mfma_util.py "$M" $OUT/${TAG}_kernel_stats.csv $OUT/${TAG}_pmc_mfma_util.json "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES -- $CMD" > /dev/null || exit 7
# the raw counter directories are large: keep the summaries
rm -rf $OUT/${TAG}_fetch $OUT/${TAG}_write $OUT/${TAG}_mfma $OUT/${TAG}_stats
head -c 1500 $OUT/${TAG}_bench.json; echo; head -12 $OUT/${TAG}_kernel_stats.csv | cut -c1-160
