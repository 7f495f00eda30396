#!/bin/bash
# tools/gpu_round_end.sh <tag>: one gpurun call at the end of a round -- rocprofv3 trace + counter passes (tools/profile.sh) of the
# main workloads, every bench workload, the GPU suite and smoke().  Summaries land under gpurun_out/prof_<tag>_* and
# gpurun_out/<tag>_final/; copy what is to be judged into profiles/ (tools/update_traffic.py, tools/update_pmc.py read them there).
tag=${1:-r05}
cd $GRAFT_REPO_ROOT
out=gpurun_out/${tag}_final
mkdir -p $out
bash tools/profile.sh ${tag}_config3 config3 --steps 3 --warmup 1 > $out/prof_config3.log 2>&1
bash tools/profile.sh ${tag}_config2 config2 --steps 10 --warmup 2 > $out/prof_config2.log 2>&1
bash tools/profile.sh ${tag}_dual dual --steps 5 --warmup 1 > $out/prof_dual.log 2>&1
bash tools/profile.sh ${tag}_middle middle --steps 5 --warmup 1 > $out/prof_middle.log 2>&1
cd $GRAFT_REPO_ROOT
for wl in config3 config2 dual dual96 config4 middle api4000 api1; do
  timeout 900 python bench.py --workload $wl > $out/bench_$wl.json 2> $out/bench_$wl.err
  python -c "
import json; d=json.load(open('$out/bench_$wl.json')); print('$wl', d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('frac'), d.get('parity'))" 2>&1 | cut -c1-300
done
(time timeout 1800 python -m pytest tests -x -q -m gpu) > $out/tests_full.log 2>&1; tail -5 $out/tests_full.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
