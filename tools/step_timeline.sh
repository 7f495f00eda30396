#!/bin/bash
# kernel timeline of one steady-state step of a resident workload: tools/step_timeline.sh <workload> [bench args]
# -> gpurun_out/timeline_<workload>.txt (start offset, duration, hardware queue, kernel)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
w=$1; shift
rm -rf /tmp/rp_tl
rocprofv3 --kernel-trace -d /tmp/rp_tl -o t --output-format csv -- python $R/bench.py --workload $w --steps 6 --warmup 3 --no-cpu-baseline "$@" > /tmp/tl_run.log 2>&1
python - > $R/gpurun_out/timeline_$w.txt <<'PY'
import csv, glob
k = glob.glob('/tmp/rp_tl/**/*kernel_trace.csv', recursive=True)[0]
ev = []
for r in csv.DictReader(open(k)):
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:64], r.get('Queue_Id', '')))
ev.sort()
packs = [i for i, e in enumerate(ev) if 'k_pack_windows' in e[2]]
# full-size steps: the pack kernels of the longest duration class; take the second to last of them
big = max(ev[i][1] - ev[i][0] for i in packs)
full = [i for i in packs if ev[i][1] - ev[i][0] > 0.7 * big]
i0 = full[-2] if len(full) > 1 else full[-1]
i1 = min([i for i in packs if i > i0] + [len(ev)])
t0 = ev[i0][0]
for s, e, name, q in ev[i0:i1]:
    print("%9.1f us  +%8.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, name))
PY
cat $R/gpurun_out/timeline_$w.txt
