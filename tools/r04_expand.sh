#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_expand; mkdir -p $out
timeout 1800 python -m pytest tests/test_hip_parity.py tests/test_hip_fuzz.py -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
B="python bench.py --no-host-inclusive --no-cpu-baseline"
for i in 1 2; do $B --workload config3 --steps 5 --warmup 2 > $out/c3_$i.json 2>/dev/null; done
$B --workload config2 --steps 20 --warmup 3 > $out/c2.json 2>/dev/null
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r04_expand/*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    k = (d.get('roofline') or {}).get('kernels_avg_ms') or {}
    print(os.path.basename(f), round(d['value'] / 1e6, 3), d['ms_per_step'], {x: round(v, 3) for x, v in k.items()})
PY
