#!/bin/bash
# round 4, first GPU call: where config 2's step goes (job classes, phase stamps of a bit-sliced unit, static-letter kernels
# forced for the medium batch, side streams), and this box's baseline lines of the other workloads
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_diag1; mkdir -p $out
B="python bench.py --no-host-inclusive --no-cpu-baseline"
QCAT_HIP_DEBUG_BINS=1 $B --workload config2 --steps 1 --warmup 1 > $out/c2_bins.json 2> $out/c2_bins.err
QCAT_HIP_BS_TRACE=1 $B --workload config2 --steps 1 --warmup 1 > $out/c2_trace.json 2> $out/c2_trace.err
for i in 1 2; do
$B --workload config2 --steps 20 --warmup 3 > $out/c2_default_$i.json 2>/dev/null
QCAT_HIP_BITSLICE_MIN=2048 $B --workload config2 --steps 20 --warmup 3 > $out/c2_static_$i.json 2>/dev/null
QCAT_HIP_BITSLICE_MIN=2048 QCAT_HIP_BS_SIDE=1 $B --workload config2 --steps 20 --warmup 3 > $out/c2_static_side_$i.json 2>/dev/null
QCAT_HIP_ABS_STAGES=2 $B --workload config2 --steps 20 --warmup 3 > $out/c2_abs2_$i.json 2>/dev/null
QCAT_HIP_NO_ADAPTER_BITSLICE=1 $B --workload config2 --steps 20 --warmup 3 > $out/c2_noabs_$i.json 2>/dev/null
done
QCAT_HIP_BITSLICE_MIN=2048 QCAT_HIP_BS_TRACE=1 $B --workload config2 --steps 1 --warmup 1 > $out/c2_static_trace.json 2> $out/c2_static_trace.err
$B --workload config3 --steps 10 --warmup 2 > $out/c3.json 2>/dev/null
$B --workload middle --steps 10 --warmup 2 > $out/middle.json 2>/dev/null
$B --workload dual96 --steps 10 --warmup 2 > $out/dual96.json 2>/dev/null
$B --workload dual --steps 10 --warmup 2 > $out/dual.json 2>/dev/null
python bench.py --workload api4000 --steps 3 --warmup 1 > $out/api4000.json 2>/dev/null
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r04_diag1/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e); continue
    k = (d.get('roofline') or {}).get('kernels_avg_ms', {})
    print(os.path.basename(f), round(d['value'] / 1e6, 2), d['ms_per_step'], {x: round(v, 3) for x, v in k.items()}, d.get('split_ms_per_call', ''))
PY
