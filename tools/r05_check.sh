#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r05_check
for v in 0 1; do
QCAT_BENCH_SMALL_WRITES=$v QCAT_BENCH_STREAM_REPEATS=6 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --reads 1000000 > gpurun_out/r05_check/bench_writes$v.json 2>/dev/null
python - $v <<'PY'
import json, sys
d = json.load(open("gpurun_out/r05_check/bench_writes%s.json" % sys.argv[1]))
f = d["host_inclusive"]["from_fastq"]
print("small_writes", sys.argv[1], f["value"], f["stream"], f["whole_file"]["value"])
PY
done
grep -i "AnonHugePages\|FileHugePages\|ShmemHuge" /proc/meminfo; cat /sys/kernel/mm/transparent_hugepage/enabled; df -T /tmp | tail -1
