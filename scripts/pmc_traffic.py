#!/usr/bin/env python
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into per-kernel HBM bytes per launch.
    python scripts/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> <note>
Correction (MI355X_MICROARCH.md, HBM / rocprofv3): gfx950 FETCH_SIZE reports half of wide coalesced
streaming reads -> fetch bytes = 2 * FETCH_SIZE[KB] * 1024; WRITE_SIZE[KB] * 1024 as is."""
import csv, json, re, sys, collections

def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    per_dispatch = collections.defaultdict(float)
    names = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        d = r["Dispatch_Id"]
        per_dispatch[d] += float(r["Counter_Value"])
        names[d] = r["Kernel_Name"]
    for d, v in per_dispatch.items():
        n = re.sub(r"\(.*$", "", names[d]).replace("void ", "")
        acc[n][0] += v; acc[n][1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}

fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"source": sys.argv[4],
       "correction": "gfx950: fetch bytes = 2 * FETCH_SIZE KB * 1024; write bytes = WRITE_SIZE KB * 1024 (MI355X_MICROARCH.md, HBM)",
       "per_launch_bytes": {}}
gemm_bytes = gemm_n = 0
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("mdbn::"):
        continue
    f = 2 * 1024 * fetch.get(k, (0, 0))[0]
    w = 1024 * write.get(k, (0, 0))[0]
    n = max(fetch.get(k, (0, 0))[1], write.get(k, (0, 0))[1])
    out["per_launch_bytes"][k] = {"fetch": f, "write": w, "total": f + w, "launches": n}
    if "gemm" in k:
        gemm_bytes += (f + w) * n; gemm_n += n
out["gemm_avg_bytes_per_launch"] = gemm_bytes / max(gemm_n, 1)
# bytes per CD STEP over every kernel of the step: the profiled command runs ONE path (bench.py --default-only), whose
# steps each launch the statistics GEMM once; kernels launched less than once per two steps (one-off splits, the free-energy
# check) are not part of the step
# (one statistics GEMM per step; with the gather-ahead the gather kernel itself is launched once per run)
steps = max([v["launches"] for k, v in out["per_launch_bytes"].items() if k.startswith("mdbn::gemm_planes_kernel<1, 1")] or
            [out["per_launch_bytes"].get("mdbn::gather_planes_kernel", {}).get("launches", 0)])
if steps:
    out["steps_profiled"] = steps
    out["step_kernels"] = {k: {"launches_per_step": v["launches"] / steps, "bytes_per_step": v["total"] * v["launches"] / steps}
                           for k, v in out["per_launch_bytes"].items() if v["launches"] * 2 >= steps}
    out["step_bytes"] = sum(v["bytes_per_step"] for v in out["step_kernels"].values())
try:        # which library was profiled (bench.py compares it with the one it times)
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mdbn_amd import build
    out["source_hash"] = build.source_hash()
except Exception:
    pass
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
