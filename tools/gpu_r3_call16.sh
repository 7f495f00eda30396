#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/c16
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rp_c16 -o trace --output-format csv -- python $R/bench.py --workload api4000 --steps 1 --warmup 1 --reads 80000 > $R/gpurun_out/c16/trace.log 2>&1
find /tmp/rp_c16 -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/c16/kernel_stats.csv \;
find /tmp/rp_c16 -name "*kernel_trace.csv" -exec cp {} $R/gpurun_out/c16/kernel_trace.csv \;
tail -2 $R/gpurun_out/c16/trace.log | cut -c1-600
python - <<PY
import csv
rows=list(csv.DictReader(open('$R/gpurun_out/c16/kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows); calls=sum(int(r['Calls']) for r in rows)
print('kernels total ms', tot/1e6, 'dispatches', calls)
for r in rows[:22]: print(r['Name'][:70], r['Calls'], r['AverageNs'])
PY
