#!/bin/bash
# what the binary16 kernels still take of the shipped dual kit: the job histogram per (group, class) of one
# step, the timing marks, and the per-kernel timeline of one step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r06_dual_bins
mkdir -p $out
QCAT_HIP_DEBUG_BINS=1 python $R/bench.py --workload dual --steps 1 --warmup 0 --no-cpu-baseline > $out/bins.json 2> $out/bins.err
grep "group" $out/bins.err | sort | uniq -c | sort -k3n -k5n > $out/bins.txt
python $R/bench.py --workload dual --steps 20 --warmup 3 --no-cpu-baseline > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace -d /tmp/rp_dual -o t --output-format csv -- python $R/bench.py --workload dual --steps 2 --warmup 1 --no-cpu-baseline > $out/trace_run.log 2>&1
python - > $out/timeline.txt <<'PY'
import csv, glob
k = glob.glob('/tmp/rp_dual/**/*kernel_trace.csv', recursive=True)[0]
ev = []
for r in csv.DictReader(open(k)):
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60], r.get('Grid_Size',''), r.get('Queue_Id','')))
ev.sort()
# the last step: from the last k_pack_windows on
i0 = max(i for i, e in enumerate(ev) if 'k_pack_windows' in e[2])
t0 = ev[i0][0]
for s, e, name, g, q in ev[i0:]:
    print("%9.1f us  +%8.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, name))
PY
cat $out/bins.txt; tail -50 $out/timeline.txt; head -c 600 $out/bench.json
