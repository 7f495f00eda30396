#!/bin/bash
# A/B of the bit-sliced interior adapter scan: workgroups per CU and launch
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_absmid_ab; mkdir -p $out
B="timeout 600 python bench.py --no-host-inclusive --no-cpu-baseline --workload middle --steps 10 --warmup 2"
for w in default 1 2 3 4; do
  if [ $w = default ]; then $B > $out/wgs_$w.json 2>/dev/null; else QCAT_HIP_MIDDLE_ABS_WGS=$w $B > $out/wgs_$w.json 2>/dev/null; fi
done
QCAT_HIP_MIDDLE_NO_ABS=1 $B > $out/f16.json 2>/dev/null
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r04_absmid_ab/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "no line", e); continue
    k = (d.get('roofline') or {}).get('kernels_avg_ms') or {}
    print(os.path.basename(f), round(d['value'] / 1e6, 3), d['ms_per_step'], {x: round(v, 3) for x, v in k.items() if 'middle' in x})
PY
