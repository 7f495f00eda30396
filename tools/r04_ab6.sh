#!/bin/bash
# round 4, call 12: device-side kit choice fixed?, lazy byte windows parity, interior barcode jobs on the bit-sliced kernels
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_ab6; mkdir -p $out
QCAT_HIP_DEBUG_VOTE=1 python tools/dbg_vote.py 2>&1 | grep -E "vote:|python" | head -3
timeout 1800 python -m pytest tests/test_batch_auto_gpu.py tests/test_hip_parity.py tests/test_hip_fuzz.py tests/test_static_kernels.py tests/test_cli_gpu.py -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest.log
B="python bench.py --no-host-inclusive --no-cpu-baseline"
for i in 1 2; do
  $B --workload middle --steps 10 --warmup 2 > $out/middle_new_$i.json 2>/dev/null
  QCAT_HIP_MIDDLE_NO_BITSLICE=1 $B --workload middle --steps 10 --warmup 2 > $out/middle_nobs_$i.json 2>/dev/null
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r04_ab6/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e); continue
    k = (d.get('roofline') or {}).get('kernels_avg_ms', {})
    print(os.path.basename(f), round(d['value'] / 1e6, 2), d['ms_per_step'], {x: round(v, 3) for x, v in k.items()})
PY
