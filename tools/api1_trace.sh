#!/bin/bash
# kernel + copy timeline of single-read calls (bench.py --workload api1 under rocprofv3 --kernel-trace --memory-copy-trace):
# gpurun_out/api1_trace/timeline.txt = 70 consecutive events of the named-kit leg with start offsets and durations
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/api1_trace
mkdir -p $out
rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/rp_api1 -o t --output-format csv -- python $R/bench.py --workload api1 --steps 1 --reads 300 > $out/trace_run.log 2>&1
ls /tmp/rp_api1/*/ | head
python - > $out/timeline.txt <<'PY'
import csv, glob
k = glob.glob('/tmp/rp_api1/**/*kernel_trace.csv', recursive=True)[0]
m = glob.glob('/tmp/rp_api1/**/*memory_copy_trace.csv', recursive=True)[0]
ev = []
for r in csv.DictReader(open(k)):
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60]))
for r in csv.DictReader(open(m)):
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', r.get('Name', ''))[:40] ))
ev.sort()
# two windows: inside the named-kit leg (the first 300 calls) and inside the kit-auto leg (the last 300)
n = len(ev)
for label, i0 in (("named kit", n // 6), ("kit auto", 3 * n // 4)):
    print("---- " + label)
    t0 = ev[i0][0]
    for s, e, name in ev[i0:i0 + 60]:
        print("%9.1f us  +%7.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, name))
PY

head -64 $out/timeline.txt
