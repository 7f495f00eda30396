#!/bin/bash
# kernel-trace stats of one bench workload: tools/ktrace.sh <tag> <workload> [bench args]  -> gpurun_out/ktrace_<tag>/kernel_stats.csv
tag=$1; wl=$2; shift 2
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/ktrace_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/rp_kt_$tag -o t --output-format csv -- python $R/bench.py --workload $wl --no-cpu-baseline --no-host-inclusive "$@" > $out/bench.log 2>&1
find /tmp/rp_kt_$tag -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
python - $out/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:28]:
    print("%-70s calls %5s avg %10.1f us total %9.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
# ... and the timeline of the last step: start offset and duration of every kernel of more than 20 us
find /tmp/rp_kt_$tag -name "*kernel_trace.csv" -exec cp {} /tmp/rp_kt_$tag/kt.csv \;
python - /tmp/rp_kt_$tag/kt.csv > $out/last_step_timeline.txt <<'PY'
import csv, sys
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:64]) for r in csv.DictReader(open(sys.argv[1]))]
ev.sort()
last_pack = max(i for i, e in enumerate(ev) if "k_pack_windows" in e[2])
t0 = ev[last_pack][0]
for s, e, name in ev[last_pack:]:
    if e - s > 20000:
        print("%9.1f us .. %9.1f us  (%8.1f us)  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, name))
PY
cat $out/last_step_timeline.txt
