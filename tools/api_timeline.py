#!/usr/bin/env python3
"""Timeline of the LAST kit-auto batch call of a rocprofv3 --kernel-trace (+ --memory-copy-trace) run of
`bench.py --workload api4000`: every kernel and copy from the call's first upload to its last download, in microseconds.
usage: api_timeline.py <trace dir>"""
import csv
import glob
import sys

ev = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0].replace("void ", "")[-50:], r.get("Stream_Id", "")))
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r["Direction"], r.get("Stream_Id", "")))
ev.sort()
fins = [i for i, e in enumerate(ev) if "k_finalize" in e[2]]
i1 = fins[-1]
packs = [i for i, e in enumerate(ev) if "k_pack_windows" in e[2] and i < i1]
i0 = packs[-1]
while i0 > 0 and ev[i0 - 1][2].startswith("C ") or "fill" in ev[i0 - 1][2]:
    i0 -= 1
while i1 + 1 < len(ev) and ev[i1 + 1][2].startswith("C ") and ev[i1 + 1][0] - ev[i1][1] < 200000:
    i1 += 1
t0 = ev[i0][0]
busy = 0
for s, e, name, q in ev[i0:i1 + 1]:
    busy += e - s
    print("%8.1f %8.1f  %7.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, name))
print("span %.1f us, sum of durations %.1f us, %d events" % ((ev[i1][1] - t0) / 1e3, busy / 1e3, i1 - i0 + 1))
