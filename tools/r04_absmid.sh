#!/bin/bash
# the bit-sliced interior adapter scan (kernels_abs_mid.inc): parity tests, then the --detect-middle workload with the path on / off
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_absmid; mkdir -p $out
timeout 1200 python -m pytest tests -x -q -m gpu -k "middle" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $out/pytest.log
B="timeout 600 python bench.py --no-host-inclusive --cpu-seconds 2"
for i in 1 2; do
  $B --workload middle --steps 10 --warmup 2 > $out/mid_abs_$i.json 2>$out/mid_abs_$i.err
  QCAT_HIP_MIDDLE_NO_ABS=1 $B --workload middle --steps 10 --warmup 2 > $out/mid_f16_$i.json 2>$out/mid_f16_$i.err
  QCAT_HIP_MIDDLE_ABS_EARLY=1 $B --workload middle --steps 10 --warmup 2 > $out/mid_abs_e1_$i.json 2>$out/mid_abs_e1_$i.err
done
tail -3 $out/mid_abs_1.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r04_absmid/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "no line", e); continue
    k = (d.get('roofline') or {}).get('kernels_avg_ms') or {}
    print(os.path.basename(f), round(d['value'] / 1e6, 3), d['ms_per_step'], d.get('parity'), {x: round(v, 3) for x, v in k.items()})
PY
