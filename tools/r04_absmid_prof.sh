#!/bin/bash
# kernel trace of the --detect-middle workload (bit-sliced interior adapter scan on)
out=gpurun_out/r04_absmid_prof; mkdir -p $GRAFT_REPO_ROOT/$out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cmd="python $R/bench.py --workload middle --steps 5 --warmup 2 --no-host-inclusive --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d /tmp/rp_mid/trace -o trace --output-format csv -- $cmd > $R/$out/bench_under_trace.log 2>&1
find /tmp/rp_mid/trace -name "*kernel_stats.csv" -exec cp {} $R/$out/kernel_stats.csv \;
python - "$R/$out/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:30]:
    print("%-100s calls %5s avg %10.1f us total %8.2f ms" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
