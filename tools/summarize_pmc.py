#!/usr/bin/env python3
"""Summarise rocprofv3 output of tools/profile.sh: per-kernel time (kernel_stats.csv) and the
per-dispatch average of every PMC counter per kernel (pmc*.csv)."""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
ks = os.path.join(d, "kernel_stats.csv")
if os.path.exists(ks):
    print("== kernel trace stats ==")
    with open(ks) as fh:
        for row in csv.DictReader(fh):
            print("%-60s calls %6s  total %12s ns  avg %12s ns  %6s %%" % (
                row.get("Name", "")[:60], row.get("Calls"), row.get("TotalDurationNs"),
                row.get("AverageNs"), row.get("Percentage")))
for f in sorted(glob.glob(os.path.join(d, "pmc*.csv"))):
    acc = defaultdict(lambda: defaultdict(float))
    n = defaultdict(lambda: defaultdict(int))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")[:48]
            c = row.get("Counter_Name")
            acc[k][c] += float(row.get("Counter_Value", 0) or 0)
            n[k][c] += 1
    print("== %s (average per dispatch) ==" % os.path.basename(f))
    for k in acc:
        print("  " + k)
        for c in acc[k]:
            print("      %-28s %18.1f  (n=%d)" % (c, acc[k][c] / max(1, n[k][c]), n[k][c]))
