#!/bin/bash
# round 6: the host-buffer calls -- hardware queues 4 (runtime default) against 12, results through the pinned block
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r06_api; mkdir -p $out
for q in 0 12; do
  for w in api4000 api1 config2 dual config3; do
    extra="--steps 20 --warmup 3"; [ $w = config3 ] && extra="--steps 8 --warmup 2"
    QCAT_HIP_HW_QUEUES=$q timeout 300 python bench.py --workload $w $extra --no-cpu-baseline --no-host-inclusive > $out/${w}_q$q.json 2>$out/${w}_q$q.err
    python - $out/${w}_q$q.json $w q$q <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[2], sys.argv[3], round(d['value'] / 1e6, 3), d['ms_per_step'], d.get('split_ms_per_call', ''))
except Exception as e:
    print(sys.argv[2], sys.argv[3], 'failed', e)
PY
  done
done
timeout 900 python -m pytest tests/test_scan_api_gpu.py tests/test_batch_auto_gpu.py -x -q -m gpu 2>&1 | tail -4
bash tools/api4000_trace.sh > $out/trace.log 2>&1; head -60 gpurun_out/api4000_trace/timeline.txt
