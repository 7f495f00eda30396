#!/bin/bash
# round 4, final tree: fuzz sweep of the bit-sliced interior adapter scan, rocprofv3 evidence for --detect-middle, bench lines, full GPU suite, smoke
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04d
(timeout 1200 python tools/fuzz_middle.py 0 48) > gpurun_out/r04d/fuzz_middle_0_47.txt 2>&1; tail -2 gpurun_out/r04d/fuzz_middle_0_47.txt
bash tools/profile.sh r04_absmid middle --steps 5 --warmup 1 > gpurun_out/r04d/prof_middle.log 2>&1
cd $GRAFT_REPO_ROOT
for wl in middle; do
  timeout 900 python bench.py --workload $wl > gpurun_out/r04d/bench_$wl.json 2> gpurun_out/r04d/bench_$wl.err
  python -c "
import json; d=json.loads(open('gpurun_out/r04d/bench_$wl.json').read().strip().splitlines()[-1]); print('$wl', d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('kernels_avg_ms'))" 2>&1 | cut -c1-400
done
bash tools/r04_absmid.sh 2>&1 | tail -7 | cut -c1-60
bash tools/gpu_final_validation.sh
