#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_mid; mkdir -p $out
timeout 1500 python -m pytest tests -x -q -m gpu -k "middle" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest.log
B="python bench.py --no-host-inclusive --no-cpu-baseline"
for i in 1 2; do $B --workload middle --steps 10 --warmup 2 > $out/mid_$i.json 2>/dev/null; done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r04_mid/*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    k = (d.get('roofline') or {}).get('kernels_avg_ms') or {}
    print(os.path.basename(f), round(d['value'] / 1e6, 3), d['ms_per_step'], {x: round(v, 3) for x, v in k.items()})
PY
