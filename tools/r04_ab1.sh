#!/bin/bash
# round 4, call 2: parity of the reworked bit-sliced barcode kernel (register transposition, producer wave for the shared
# columns, generated kernels at every batch size), then A/B on one box against the round-3 forms
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_ab1; mkdir -p $out
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_static_kernels.py tests/test_jit.py -x -q -m gpu -k "bit_sliced or bitslice or static or jit" > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log; tail -3 $out/pytest.log
B="python bench.py --no-host-inclusive --no-cpu-baseline"
OLD=$PWD/qcat_amd/csrc/build/ab/oldt.so
run() { name=$1; shift; env "$@" > /dev/null 2>&1; }
for i in 1 2; do
  $B --workload config2 --steps 20 --warmup 3 > $out/c2_new_$i.json 2>/dev/null
  QCAT_HIP_BS_NO_SOLO=1 $B --workload config2 --steps 20 --warmup 3 > $out/c2_nosolo_$i.json 2>/dev/null
  QCAT_HIP_BS_STATIC_MIN=1024 $B --workload config2 --steps 20 --warmup 3 > $out/c2_dyn_$i.json 2>/dev/null
  QCAT_HIP_LIBRARY=$OLD $B --workload config2 --steps 20 --warmup 3 > $out/c2_oldt_$i.json 2>/dev/null
  QCAT_HIP_LIBRARY=$OLD QCAT_HIP_BS_NO_SOLO=1 QCAT_HIP_BS_STATIC_MIN=1024 $B --workload config2 --steps 20 --warmup 3 > $out/c2_r03_$i.json 2>/dev/null
  $B --workload config3 --steps 8 --warmup 2 > $out/c3_new_$i.json 2>/dev/null
  QCAT_HIP_LIBRARY=$OLD $B --workload config3 --steps 8 --warmup 2 > $out/c3_oldt_$i.json 2>/dev/null
  $B --workload dual --steps 10 --warmup 2 > $out/dual_new_$i.json 2>/dev/null
  QCAT_HIP_LIBRARY=$OLD QCAT_HIP_BS_NO_SOLO=1 $B --workload dual --steps 10 --warmup 2 > $out/dual_r03_$i.json 2>/dev/null
  $B --workload dual96 --steps 10 --warmup 2 > $out/dual96_new_$i.json 2>/dev/null
  QCAT_HIP_LIBRARY=$OLD QCAT_HIP_BS_NO_SOLO=1 $B --workload dual96 --steps 10 --warmup 2 > $out/dual96_r03_$i.json 2>/dev/null
done
QCAT_HIP_BS_STATIC_MIN=1024 QCAT_HIP_BS_TRACE=1 $B --workload config2 --steps 1 --warmup 1 > $out/c2_dyn_trace.json 2> $out/c2_dyn_trace.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r04_ab1/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e); continue
    k = (d.get('roofline') or {}).get('kernels_avg_ms', {})
    print(os.path.basename(f), round(d['value'] / 1e6, 2), d['ms_per_step'], {x: round(v, 3) for x, v in k.items()})
PY
head -34 $out/c2_dyn_trace.err
