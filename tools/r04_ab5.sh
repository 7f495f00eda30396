#!/bin/bash
# round 4, call 11: the device-side kit choice (diagnostics), lazy byte windows, api latency cuts
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_ab5; mkdir -p $out
QCAT_HIP_DEBUG_VOTE=1 python tools/dbg_vote.py 2>&1 | grep -E "vote:|scan_auto|python" | head -8
timeout 1200 python -m pytest tests/test_batch_auto_gpu.py tests/test_hip_parity.py tests/test_hip_fuzz.py -x -q -m gpu -k "not adapter_kernels" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest.log
B="python bench.py --no-host-inclusive --no-cpu-baseline"
for i in 1 2; do
  $B --workload config2 --steps 20 --warmup 3 > $out/c2_new_$i.json 2>/dev/null
  QCAT_HIP_EAGER_BYTES=1 $B --workload config2 --steps 20 --warmup 3 > $out/c2_eager_$i.json 2>/dev/null
  $B --workload config3 --steps 8 --warmup 2 > $out/c3_new_$i.json 2>/dev/null
  QCAT_HIP_EAGER_BYTES=1 $B --workload config3 --steps 8 --warmup 2 > $out/c3_eager_$i.json 2>/dev/null
  python bench.py --workload api4000 --steps 3 --warmup 1 > $out/api_$i.json 2>/dev/null
done
$B --workload dual --steps 10 --warmup 2 > $out/dual_new.json 2>/dev/null
$B --workload middle --steps 10 --warmup 2 > $out/middle_new.json 2>/dev/null
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r04_ab5/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e); continue
    k = (d.get('roofline') or {}).get('kernels_avg_ms', {})
    print(os.path.basename(f), round(d['value'] / 1e6, 2), d['ms_per_step'], {x: round(v, 3) for x, v in k.items()}, d.get('split_ms_per_call', ''))
PY
