#!/bin/bash
# issue-priority rotation of the read ends' bit-sliced adapter kernels (QCAT_HIP_ABS_PRIO) at the final kernels
cd ${GRAFT_REPO_ROOT:-.}
for wl in config3 config2 dual; do
  for p in 0 1 2; do
    QCAT_HIP_ABS_PRIO=$p timeout 600 python bench.py --no-host-inclusive --no-cpu-baseline --workload $wl --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl prio $p', d['ms_per_step'], d['roofline']['kernels_avg_ms']['k_adapter_bitslice'])"
  done
done
