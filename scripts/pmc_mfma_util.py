#!/usr/bin/env python
"""Summarise a rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES ...) plus a
kernel trace of the same command into per-kernel MFMA utilisation.
    python scripts/pmc_mfma_util.py <counter_collection.csv> <kernel_stats.csv> <out.json> <note>"""
import csv, json, re, sys, collections
cnt = collections.defaultdict(lambda: collections.defaultdict(float)); nl = collections.Counter(); seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "")
    if not k.startswith("mdbn::gemm"): continue
    cnt[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (r["Dispatch_Id"], k) not in seen:
        seen.add((r["Dispatch_Id"], k)); nl[k] += 1
dur = {}
for r in csv.DictReader(open(sys.argv[2])):
    k = re.sub(r"\(.*$", "", r["Name"]).replace("void ", "")
    dur[k] = float(r["AverageNs"]) / 1e3
out = {"source": sys.argv[4], "note": "mfma_util_vs_2p4ghz = SQ_VALU_MFMA_BUSY_CYCLES per launch / (2.4 GHz x avg duration (kernel trace, not the slower "
       "counter run) x 1024 SIMDs); busy cycles per instruction: 64 for v_mfma_f32_32x32x2_f32, 32 for v_mfma_f32_32x32x16_bf16", "kernels": {}}
for k in cnt:
    n = nl[k]; busy = cnt[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / n
    d = dur.get(k)
    out["kernels"][k] = {"launches": n, "avg_us_trace": d, "mfma_busy_cycles_per_launch": busy,
                         "mfma_util_vs_2p4ghz": busy / (2.4e3 * d * 1024) if d else None}
json.dump(out, open(sys.argv[3], "w"), indent=1); print(json.dumps(out, indent=1))
