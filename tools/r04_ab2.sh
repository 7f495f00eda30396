#!/bin/bash
# round 4, call 3: the lane-split transposition + left-over tiles beside the bit-sliced launches; A/B against round 3's
# transposition (build oldt) on one box; phase stamps of the generated kernels
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_ab2; mkdir -p $out
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "bit_sliced_barcode" > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log; tail -3 $out/pytest.log
B="python bench.py --no-host-inclusive --no-cpu-baseline"
OLD=$PWD/qcat_amd/csrc/build/ab/oldt.so
for i in 1 2; do
  $B --workload config2 --steps 20 --warmup 3 > $out/c2_new_$i.json 2>/dev/null
  QCAT_HIP_LEFTOVER_SIDE=0 $B --workload config2 --steps 20 --warmup 3 > $out/c2_noside_$i.json 2>/dev/null
  QCAT_HIP_LIBRARY=$OLD QCAT_HIP_LEFTOVER_SIDE=0 $B --workload config2 --steps 20 --warmup 3 > $out/c2_oldt_$i.json 2>/dev/null
  QCAT_HIP_LIBRARY=$OLD QCAT_HIP_LEFTOVER_SIDE=0 QCAT_HIP_BS_NO_SOLO=1 QCAT_HIP_BS_STATIC_MIN=1024 $B --workload config2 --steps 20 --warmup 3 > $out/c2_r03_$i.json 2>/dev/null
  $B --workload config3 --steps 8 --warmup 2 > $out/c3_new_$i.json 2>/dev/null
  QCAT_HIP_LIBRARY=$OLD $B --workload config3 --steps 8 --warmup 2 > $out/c3_oldt_$i.json 2>/dev/null
  $B --workload dual --steps 10 --warmup 2 > $out/dual_new_$i.json 2>/dev/null
  QCAT_HIP_LIBRARY=$OLD QCAT_HIP_BS_NO_SOLO=1 $B --workload dual --steps 10 --warmup 2 > $out/dual_r03_$i.json 2>/dev/null
done
QCAT_HIP_LEFTOVER_SIDE=0 QCAT_HIP_BS_TRACE=1 $B --workload config2 --steps 1 --warmup 1 > $out/c2_trace.json 2> $out/c2_trace.err
QCAT_HIP_BS_TRACE=1 $B --workload config3 --reads 2000000 --steps 1 --warmup 1 > $out/c3_trace.json 2> $out/c3_trace.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r04_ab2/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e); continue
    k = (d.get('roofline') or {}).get('kernels_avg_ms', {})
    print(os.path.basename(f), round(d['value'] / 1e6, 2), d['ms_per_step'], {x: round(v, 3) for x, v in k.items()})
PY
grep "launch . unit [01] " $out/c2_trace.err | head -70
grep "launch . unit [01] " $out/c3_trace.err | head -40
