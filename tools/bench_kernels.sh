#!/bin/bash
# usage: tools/bench_kernels.sh <workload> [extra bench args] -- prints value + per-kernel ms
out=$(python bench.py --workload "$1" --no-cpu-baseline --no-host-inclusive "${@:2}" 2>&1 | tail -1)
python - "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[1])
print("%-8s %12.0f reads/s  %s" % (d["config"]["kit"], d["value"], d["roofline"]["kernels_avg_ms"]))
PY
