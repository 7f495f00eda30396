#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/all
bash tools/profile.sh r03g_config3 config3 --steps 3 --warmup 1 > gpurun_out/all/prof_config3.log 2>&1
bash tools/profile.sh r03g_config2 config2 --steps 10 --warmup 2 > gpurun_out/all/prof_config2.log 2>&1
bash tools/profile.sh r03g_dual dual --steps 5 --warmup 1 > gpurun_out/all/prof_dual.log 2>&1
cd $GRAFT_REPO_ROOT
for wl in config3 config2 dual config4 dual96 middle api4000; do
  timeout 600 python bench.py --workload $wl > gpurun_out/all/bench_$wl.json 2> gpurun_out/all/bench_$wl.err
  python -c "
import json; d=json.load(open('gpurun_out/all/bench_$wl.json')); print('$wl', d['value'], d['ms_per_step'])" 2>&1 | cut -c1-200
done
