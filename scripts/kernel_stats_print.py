#!/usr/bin/env python
"""Print (name, calls, average us) of the mdbn kernels in a rocprofv3 kernel_stats.csv."""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "mdbn::" in r["Name"]:
        print("  %-70s calls %6s  avg %8.2f us" % (r["Name"].replace("void ", "").split("(")[0][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
