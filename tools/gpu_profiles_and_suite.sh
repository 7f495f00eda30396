#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/run
bash tools/profile.sh r03f_config3 config3 --steps 3 --warmup 1 > gpurun_out/run/prof_config3.log 2>&1
bash tools/profile.sh r03f_config2 config2 --steps 10 --warmup 2 > gpurun_out/run/prof_config2.log 2>&1
bash tools/profile.sh r03f_dual dual --steps 5 --warmup 1 > gpurun_out/run/prof_dual.log 2>&1
cd $GRAFT_REPO_ROOT
for wl in config3 config2 dual config4 api4000; do
  timeout 600 python bench.py --workload $wl > gpurun_out/run/bench_$wl.json 2> gpurun_out/run/bench_$wl.err
  python -c "
import json; d=json.load(open('gpurun_out/run/bench_$wl.json')); print('$wl', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'), d.get('valu_issue'))" 2>&1 | cut -c1-400
done
(time timeout 1500 python -m pytest tests -x -q -m gpu) > gpurun_out/run/tests_full.log 2>&1; tail -5 gpurun_out/run/tests_full.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
