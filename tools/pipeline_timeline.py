#!/usr/bin/env python3
"""Merge rocprofv3 kernel + memory-copy traces into one timeline (ms) -- used to check that the host-buffer
pipeline overlaps H2D copies with the kernels of the previous chunk.  usage: pipeline_timeline.py <dir> [n_last]"""
import csv, glob, sys
d = sys.argv[1]; n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 120
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:40], r.get("Stream_Id", r.get("Queue_Id", ""))))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r["Direction"][12:], r.get("Stream_Id", "")))
ev.sort()
t0 = ev[-n_last][0] if len(ev) >= n_last else ev[0][0]
for s, e, name, q in ev[-n_last:]:
    if e - s > 200000:
        print("%9.3f %9.3f  %8.3f ms  q%-3s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, name))
