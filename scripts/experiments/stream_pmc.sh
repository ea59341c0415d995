#!/bin/bash
# HBM traffic (L2 fills / write-backs) of the streaming kernel's launches at a mid-size layer: two PMC passes
# (FETCH_SIZE, WRITE_SIZE; corrected as MI355X_MICROARCH.md prescribes by scripts/pmc_traffic.py), per launch.
#   gpurun -- 'MDBN_AB_SHAPE=2048,400,512,1,1 bash scripts/experiments/stream_pmc.sh r05zp'
set -o pipefail
TAG=${1:-r05x}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
export MDBN_AB_SHAPE=${MDBN_AB_SHAPE:-2048,400,512,1,1}
CMD="python3 scripts/step_ab.py stream_x6 2"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_sfetch -- $CMD > $OUT/${TAG}_sfetch.log 2>&1 || exit 3
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_swrite -- $CMD > $OUT/${TAG}_swrite.log 2>&1 || exit 4
F=$(find $OUT/${TAG}_sfetch -name '*counter_collection.csv' | head -1)
W=$(find $OUT/${TAG}_swrite -name '*counter_collection.csv' | head -1)
python3 scripts/pmc_traffic.py "$F" "$W" $OUT/${TAG}_stream_pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- MDBN_AB_SHAPE=$MDBN_AB_SHAPE $CMD" > /dev/null || exit 6
rm -rf $OUT/${TAG}_sfetch $OUT/${TAG}_swrite
python3 - <<PY
import json
d = json.load(open("$OUT/${TAG}_stream_pmc_traffic.json"))
for n, v in sorted(d["per_launch_bytes"].items()):
    print("%-60s launches %5d  fetch %7.2f MB  write %7.2f MB" % (n[:60], v["launches"], v["fetch"] / 1e6, v["write"] / 1e6))
PY
