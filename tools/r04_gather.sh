#!/bin/bash
# round 4: k_mid_gather without the covered tiles, k_finalize at 1024 blocks (the tile images beside the bit-sliced launches: measured in an earlier form of this script and dropped)
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_gather; mkdir -p $out
timeout 2400 python -m pytest tests/test_hip_parity.py tests/test_static_kernels.py tests/test_hip_fullsize.py tests/test_batch_auto_gpu.py -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest.log
B="python bench.py --no-host-inclusive --no-cpu-baseline"
for i in 1 2 3; do
  $B --workload config2 --steps 20 --warmup 3 > $out/c2_new_$i.json 2>/dev/null
  true
done
for i in 1 2; do $B --workload middle --steps 10 --warmup 2 > $out/mid_$i.json 2>/dev/null; done
$B --workload dual --steps 10 --warmup 2 > $out/dual.json 2>/dev/null
$B --workload config3 --steps 5 --warmup 2 > $out/c3.json 2>/dev/null
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r04_gather/*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    k = (d.get('roofline') or {}).get('kernels_avg_ms') or {}
    print(os.path.basename(f), round(d['value'] / 1e6, 3), d['ms_per_step'], {x: round(v, 3) for x, v in k.items()})
PY
