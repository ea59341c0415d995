#!/bin/bash
# HBM traffic of the thin-batch CD step (csrc/mdbn_thin.hip) at the 19 937-gene layer, batch 20: two PMC passes
# (FETCH_SIZE, WRITE_SIZE; corrected as MI355X_MICROARCH.md prescribes by scripts/pmc_traffic.py), per launch and per step.
#   gpurun -- 'bash scripts/experiments/thin_pmc.sh r05j'
set -o pipefail
TAG=${1:-r05x}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
export MDBN_AB_SHAPE=${MDBN_AB_SHAPE:-19937,400,20,1,1}
CMD="python3 scripts/step_ab.py thin_fused 1"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_tfetch -- $CMD > $OUT/${TAG}_tfetch.log 2>&1 || exit 3
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_twrite -- $CMD > $OUT/${TAG}_twrite.log 2>&1 || exit 4
F=$(find $OUT/${TAG}_tfetch -name '*counter_collection.csv' | head -1)
W=$(find $OUT/${TAG}_twrite -name '*counter_collection.csv' | head -1)
python3 scripts/pmc_traffic.py "$F" "$W" $OUT/${TAG}_thin_pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- MDBN_AB_SHAPE=$MDBN_AB_SHAPE $CMD" > /dev/null || exit 6
rm -rf $OUT/${TAG}_tfetch $OUT/${TAG}_twrite
python3 - <<PY
import json
d = json.load(open("$OUT/${TAG}_thin_pmc_traffic.json"))
k = {n: v for n, v in d["per_launch_bytes"].items() if "thin_" in n}
steps = max(v["launches"] for n, v in k.items() if "thin_update" in n)
tot = 0.0
for n, v in sorted(k.items()):
    per_step = v["total"] * v["launches"] / steps
    tot += per_step
    print("%-48s %5.2f launches/step  fetch %7.2f MB  write %7.2f MB  per step %7.2f MB" % (n[:48], v["launches"] / steps, v["fetch"] / 1e6, v["write"] / 1e6, per_step / 1e6))
d["thin_step_bytes"] = tot
d["thin_steps_profiled"] = steps
json.dump(d, open("$OUT/${TAG}_thin_pmc_traffic.json", "w"), indent=1)
print("HBM bytes per thin step: %.1f MB" % (tot / 1e6))
PY
