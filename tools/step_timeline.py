#!/usr/bin/env python3
"""Timeline of the LAST scan of a rocprofv3 --kernel-trace run of bench.py: every kernel between the last two k_pack_windows
launches... i.e. from the last k_pack_windows to the last k_finalize, start / end in microseconds relative to the pack kernel.
usage: step_timeline.py <trace dir>"""
import csv
import glob
import sys

ev = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", ""))))
ev.sort()
packs = [i for i, e in enumerate(ev) if "k_pack_windows" in e[2]]
fins = [i for i, e in enumerate(ev) if "k_finalize" in e[2]]
if not packs or not fins:
    sys.exit("no scan in the trace")
i1 = fins[-1]
i0 = max(i for i in packs if i < i1)
t0 = ev[i0][0]
for s, e, name, q in ev[i0:i1 + 1]:
    short = name.split("(")[0].replace("void ", "")[-58:]
    print("%8.1f %8.1f  %7.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, short))
