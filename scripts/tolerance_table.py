#!/usr/bin/env python
"""Markdown table of the tolerances the GPU parity tests use, SURVEY 8d's value for the quantity and the worst value
measured on the box (tests/_margins.py writes gpurun_out/tolerance_margins.json at the end of a `pytest -m gpu` run).
    python scripts/tolerance_table.py [gpurun_out/tolerance_margins.json]"""
import json, re, sys
path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tolerance_margins.json"
d = json.load(open(path))
groups = {}
for name, r in d.items():
    key = re.sub(r"\s*\[[^\]]*\]", "", name)              # fold the per-shape entries of one quantity
    key = re.sub(r"^(CD-\d+ \d+->\d+ (GRBM|RBM)|100 steps \d+->\d+|product-default c2 (GRBM|RBM)|DP 2 ranks (small|c2)): ", lambda m: m.group(0), key)
    g = groups.setdefault(key, {"worst": 0.0, "tol": 0.0, "tol_min": 1e9, "survey": r.get("survey"), "n": 0, "at": None})
    if r["worst"] >= g["worst"]:
        g["worst"], g["at"] = r["worst"], name
    g["tol"] = max(g["tol"], r["tol"]); g["tol_min"] = min(g["tol_min"], r["tol"]); g["n"] += r["n"]
print("| quantity (device fp32 vs float64 oracle) | tolerance in the test | SURVEY 8d | worst measured | margin | where the worst was |")
print("|---|---|---|---|---|---|")
for key in sorted(groups):
    g = groups[key]
    tol = ("%.1e" % g["tol"]) if g["tol"] == g["tol_min"] else ("%.1e ... %.1e" % (g["tol_min"], g["tol"]))
    print("| %s | %s | %s | %.2e | %s | %s |" % (key, tol, ("%.0e" % g["survey"]) if g["survey"] else "--", g["worst"],
                                               ("%.1fx" % (g["tol"] / g["worst"])) if g["worst"] > 0 else "exact", g["at"]))
