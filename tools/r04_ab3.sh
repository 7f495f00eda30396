#!/bin/bash
# round 4, call 4: tail split of the long units, priority rotation by the SIMD's barcode waves, parallel k_bs_plan; barcodes of
# unequal length in simple mode; the N = 8 rehearsal; the statistics rules of qcat_sg_align
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_ab3; mkdir -p $out
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_simple_gpu.py tests/test_sg_align_gpu.py tests/test_comm_gpu.py -x -q -m gpu -k "bit_sliced_barcode or simple or sg_align or device_alignments or rehearsal or known_answer" -s > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log; grep -E "rehearsal:|passed|failed" $out/pytest.log | tail -5
B="python bench.py --no-host-inclusive --no-cpu-baseline"
for i in 1 2; do
  $B --workload config2 --steps 20 --warmup 3 > $out/c2_new_$i.json 2>/dev/null
  QCAT_HIP_BS_NO_TAIL_SPLIT=1 $B --workload config2 --steps 20 --warmup 3 > $out/c2_nosplit_$i.json 2>/dev/null
  $B --workload config2 --reads 500000 --steps 20 --warmup 3 > $out/c2h_new_$i.json 2>/dev/null
  QCAT_HIP_BS_NO_TAIL_SPLIT=1 $B --workload config2 --reads 500000 --steps 20 --warmup 3 > $out/c2h_nosplit_$i.json 2>/dev/null
  $B --workload config3 --steps 8 --warmup 2 > $out/c3_new_$i.json 2>/dev/null
  $B --workload dual --steps 10 --warmup 2 > $out/dual_new_$i.json 2>/dev/null
  QCAT_HIP_BS_NO_TAIL_SPLIT=1 $B --workload dual --steps 10 --warmup 2 > $out/dual_nosplit_$i.json 2>/dev/null
  $B --workload dual96 --steps 10 --warmup 2 > $out/dual96_new_$i.json 2>/dev/null
done
QCAT_HIP_LEFTOVER_SIDE=0 QCAT_HIP_BS_TRACE=1 $B --workload config2 --steps 1 --warmup 1 > $out/c2_trace.json 2> $out/c2_trace.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r04_ab3/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e); continue
    k = (d.get('roofline') or {}).get('kernels_avg_ms', {})
    print(os.path.basename(f), round(d['value'] / 1e6, 2), d['ms_per_step'], {x: round(v, 3) for x, v in k.items()})
PY
grep "launch 1 unit 0 " $out/c2_trace.err | head -16
