#!/bin/bash
# round 4, call 14: plane kernels at eight waves per tile (default) against four; adapter parity on the new default; config 3 once
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_ab8; mkdir -p $out
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_static_kernels.py -x -q -m gpu -k "adapter or golden or static" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
B="python bench.py --no-host-inclusive --no-cpu-baseline"
AB=$PWD/qcat_amd/csrc/build/ab
for i in 1 2 3; do
  $B --workload config2 --steps 20 --warmup 3 > $out/c2_pw8_$i.json 2>/dev/null
  QCAT_HIP_LIBRARY=$AB/pw4.so $B --workload config2 --steps 20 --warmup 3 > $out/c2_pw4_$i.json 2>/dev/null
done
$B --workload config3 --steps 5 --warmup 2 > $out/c3_pw8.json 2>/dev/null
QCAT_HIP_LIBRARY=$AB/pw4.so $B --workload config3 --steps 5 --warmup 2 > $out/c3_pw4.json 2>/dev/null
QCAT_HIP_LEFTOVER_SIDE=1 $B --workload config3 --steps 5 --warmup 2 > $out/c3_side.json 2>/dev/null
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r04_ab8/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e); continue
    k = (d.get('roofline') or {}).get('kernels_avg_ms', {})
    print(os.path.basename(f), round(d['value'] / 1e6, 2), d['ms_per_step'], {x: round(v, 3) for x, v in k.items()})
PY
