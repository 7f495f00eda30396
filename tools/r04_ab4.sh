#!/bin/bash
# round 4, call 6: unit log of config 2 (is the tail split doing anything?), the merged fills, VMK001's wide plan, run-time
# generated adapter plans of custom kits
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_ab4; mkdir -p $out
B="python bench.py --no-host-inclusive --no-cpu-baseline"
QCAT_HIP_LEFTOVER_SIDE=0 QCAT_HIP_BS_TRACE=1 $B --workload config2 --steps 1 --warmup 1 > $out/c2_trace.json 2> $out/c2_trace.err
QCAT_HIP_LEFTOVER_SIDE=0 QCAT_HIP_BS_NO_TAIL_SPLIT=1 QCAT_HIP_BS_TRACE=1 $B --workload config2 --steps 1 --warmup 1 > $out/c2_trace_nosplit.json 2> $out/c2_trace_nosplit.err
QCAT_HIP_BS_TRACE=1 $B --workload config2 --steps 1 --warmup 1 > $out/c2_trace_side.json 2> $out/c2_trace_side.err
grep "bs units" $out/c2_trace.err | tail -12; echo; grep "bs units" $out/c2_trace_nosplit.err | tail -12; echo; grep "bs units" $out/c2_trace_side.err | tail -12
for i in 1 2; do
  $B --workload config2 --steps 20 --warmup 3 > $out/c2_new_$i.json 2>/dev/null
  QCAT_HIP_NO_FILL_MERGE=1 $B --workload config2 --steps 20 --warmup 3 > $out/c2_nomerge_$i.json 2>/dev/null
  QCAT_HIP_BS_NO_TAIL_SPLIT=1 $B --workload config2 --steps 20 --warmup 3 > $out/c2_nosplit_$i.json 2>/dev/null
done
$B --workload config3 --steps 8 --warmup 2 > $out/c3_new_1.json 2>/dev/null
QCAT_HIP_NO_FILL_MERGE=1 $B --workload config3 --steps 8 --warmup 2 > $out/c3_nomerge_1.json 2>/dev/null
python bench.py --workload api4000 --steps 3 --warmup 1 > $out/api4000.json 2>/dev/null
QCAT_HIP_NO_FILL_MERGE=1 python bench.py --workload api4000 --steps 3 --warmup 1 > $out/api4000_nomerge.json 2>/dev/null
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r04_ab4/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e); continue
    k = (d.get('roofline') or {}).get('kernels_avg_ms', {})
    print(os.path.basename(f), round(d['value'] / 1e6, 2), d['ms_per_step'], {x: round(v, 3) for x, v in k.items()}, d.get('split_ms_per_call', ''))
PY
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_jit.py tests/test_batch_auto_gpu.py tests/test_static_kernels.py -x -q -m gpu -k "bit_sliced_adapter or generated_bit_sliced or batch or middle" > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log; tail -3 $out/pytest.log
