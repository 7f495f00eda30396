#!/bin/bash
# round 6 A/B: barcodes of a unit round robin (0) / drawn from a counter (1), same library, same box
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r06_draw; mkdir -p $out
for r in 1 2; do
  for d in 0 1; do
    QCAT_HIP_BS_DRAW=$d timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-inclusive > $out/draw${d}_$r.json 2>/dev/null
    python - $out/draw${d}_$r.json draw$d <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
k = d['roofline']['kernels_avg_ms']
print(sys.argv[2], round(d['value'] / 1e6, 2), d['ms_per_step'], {x: round(v, 3) for x, v in k.items() if v > 0.3})
PY
  done
done
for d in 0 1; do
QCAT_HIP_BS_DRAW=$d timeout 300 python bench.py --workload config2 --steps 20 --warmup 3 --no-cpu-baseline --no-host-inclusive > $out/c2_draw${d}.json 2>/dev/null
QCAT_HIP_BS_DRAW=$d timeout 300 python bench.py --workload dual --steps 20 --warmup 3 --no-cpu-baseline --no-host-inclusive > $out/dual_draw${d}.json 2>/dev/null
python - $out/c2_draw${d}.json $out/dual_draw${d}.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    d = json.load(open(f)); print(f, round(d['value'] / 1e6, 2), d['ms_per_step'])
PY
done
QCAT_HIP_BS_TRACE=1 QCAT_HIP_BS_DRAW=1 timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-inclusive > $out/trace_draw1.json 2> $out/trace_draw1.txt
QCAT_HIP_BS_DRAW=1 timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -3
