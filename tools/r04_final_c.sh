#!/bin/bash
# round 4, after the bit-sliced interior adapter scan: rocprofv3 evidence for --detect-middle (trace + counter passes), its bench line,
# the A/B of tools/r04_absmid.sh, the default bench line, the full GPU suite, smoke
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04c
bash tools/profile.sh r04_absmid middle --steps 5 --warmup 1 > gpurun_out/r04c/prof_middle.log 2>&1
cd $GRAFT_REPO_ROOT
for wl in middle config2 dual; do
  timeout 900 python bench.py --workload $wl > gpurun_out/r04c/bench_$wl.json 2> gpurun_out/r04c/bench_$wl.err
  python -c "
import json; d=json.loads(open('gpurun_out/r04c/bench_$wl.json').read().strip().splitlines()[-1]); print('$wl', d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('kernels_avg_ms'))" 2>&1 | cut -c1-400
done
bash tools/r04_absmid.sh 2>&1 | tail -8
bash tools/gpu_final_validation.sh
