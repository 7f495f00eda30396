#!/bin/bash
# round 4: (1) the occupancy experiment (tools/occupancy_pair.hip), (2) more hardware queues for the side streams
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_occ; mkdir -p $out
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/occupancy_pair.hip -o /tmp/occupancy_pair 2>$out/occ_build.log
for i in 1 2 3; do timeout 120 /tmp/occupancy_pair; done | tee $out/occupancy_pair.txt
B="python bench.py --no-host-inclusive --no-cpu-baseline"
for q in 4 8 16; do
  for i in 1 2; do
    GPU_MAX_HW_QUEUES=$q $B --workload config2 --steps 20 --warmup 3 > $out/c2_q${q}_$i.json 2>/dev/null
    GPU_MAX_HW_QUEUES=$q timeout 600 python bench.py --workload api4000 > $out/api_q${q}_$i.json 2>/dev/null
  done
  GPU_MAX_HW_QUEUES=$q $B --workload dual --steps 10 --warmup 2 > $out/dual_q${q}.json 2>/dev/null
done
GPU_MAX_HW_QUEUES=8 $B --workload config3 --steps 5 --warmup 2 > $out/c3_q8.json 2>/dev/null
$B --workload config3 --steps 5 --warmup 2 > $out/c3_q4.json 2>/dev/null
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r04_occ/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e); continue
    k = (d.get('roofline') or {}).get('kernels_avg_ms') or d.get('split_ms_per_call')
    print(os.path.basename(f), round(d['value'] / 1e6, 3), d['ms_per_step'], {x: round(v, 3) for x, v in (k or {}).items()})
PY
