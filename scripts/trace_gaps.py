#!/usr/bin/env python
"""Per-kernel duration and gap-to-previous from a rocprofv3 kernel-trace CSV (last 20 steps)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = [(r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("mdbn::", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
# steps begin with gather_rows_kernel
starts = [i for i, k in enumerate(ks) if k[0].startswith("gather_rows")]
starts = starts[-21:]
per = collections.OrderedDict()
tot = 0
for a, b in zip(starts[:-1], starts[1:]):
    for pos, i in enumerate(range(a, b)):
        name, s, e = ks[i]
        gap = s - ks[i - 1][2]
        d = per.setdefault((pos, name), [0, 0, 0])
        d[0] += e - s; d[1] += gap; d[2] += 1
    tot += ks[b][1] - ks[a][1]
n = len(starts) - 1
print("step %.1f us over %d steps" % (tot / n / 1e3, n))
sd = sg = 0
for (pos, name), (d, g, c) in per.items():
    print("%2d %-28s dur %6.2f us  gap-before %6.2f us" % (pos, name[:28], d / c / 1e3, g / c / 1e3))
    sd += d / c; sg += g / c
print("sum dur %.1f us, sum gaps %.1f us" % (sd / 1e3, sg / 1e3))
