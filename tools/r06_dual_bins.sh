#!/bin/bash
# what the binary16 kernels still take of the shipped dual kit: the job histogram per (group, class) of one
# step, the timing marks, and the per-kernel timeline of one step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r06_dual_bins
mkdir -p $out
rocprofv3 --kernel-trace -d /tmp/rp_dual -o t --output-format csv -- python $R/bench.py --workload dual --steps 2 --warmup 1 --no-cpu-baseline > $out/trace_run.log 2>&1
python - > $out/timeline.txt <<'PY'
import csv, glob
k = glob.glob('/tmp/rp_dual/**/*kernel_trace.csv', recursive=True)[0]
ev = []
for r in csv.DictReader(open(k)):
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60], r.get('Grid_Size',''), r.get('Queue_Id','')))
ev.sort()
# a full-size step: from the LONGEST k_pack_windows to the next one
packs = [i for i, e in enumerate(ev) if 'k_pack_windows' in e[2]]
i0 = max(packs, key=lambda i: ev[i][1] - ev[i][0])
i1 = min([i for i in packs if i > i0] + [len(ev)])
t0 = ev[i0][0]
for s, e, name, g, q in ev[i0:i1]:
    print("%9.1f us  +%8.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, name))
PY
cat $out/timeline.txt
