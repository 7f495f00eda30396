#!/bin/bash
# kernel + copy timeline of the reference driver's call shape (bench.py --workload api4000 under rocprofv3 --kernel-trace --memory-copy-trace):
# gpurun_out/api4000_trace/timeline.txt = the events of two consecutive calls in the middle of the run, with start offsets and durations
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/api4000_trace
mkdir -p $out
python $R/bench.py --workload api4000 --steps 20 --warmup 3 > $out/bench_api4000.json 2>$out/bench.err
rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/rp_api4000 -o t --output-format csv -- python $R/bench.py --workload api4000 --steps 3 --warmup 1 --reads 40000 > $out/trace_run.log 2>&1
python - > $out/timeline.txt <<'PY'
import csv, glob
k = glob.glob('/tmp/rp_api4000/**/*kernel_trace.csv', recursive=True)[0]
m = glob.glob('/tmp/rp_api4000/**/*memory_copy_trace.csv', recursive=True)
ev = []
for r in csv.DictReader(open(k)):
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:70]))
if m:
    for r in csv.DictReader(open(m[0])):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', r.get('Name', ''))[:40]))
ev.sort()
n = len(ev)
# find call boundaries: an H2D copy after a D2H copy
i0 = 3 * n // 4
while i0 < n and not ev[i0][2].startswith('COPY'): i0 += 1
t0 = ev[i0][0]
for s, e, name in ev[i0:i0 + 150]:
    print("%9.1f us  +%7.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, name))
PY
head -120 $out/timeline.txt
cat $out/bench_api4000.json | head -c 900
