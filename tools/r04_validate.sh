#!/bin/bash
# round 4: the whole GPU suite, the fuzz sweep, smoke(), the default bench line
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_validate; mkdir -p $out
(time timeout 2400 python -m pytest tests -x -q -m gpu) > $out/tests_full.log 2>&1; echo "pytest rc=$?"; tail -5 $out/tests_full.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log
(time timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err) 2>&1 | grep real
python -c "
import json; d=json.loads(open('$out/bench_default.json').read().strip().splitlines()[-1]); print(d['metric'], d['value'], d['ms_per_step'], d['roofline'], d['cpu_baseline'])
print({k:(v.get('ms'),v.get('issue_util')) for k,v in (d.get('valu_issue') or {}).get('marks',{}).items()})" | cut -c1-1500
timeout 1200 bash tools/gpu_fuzz.sh > $out/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -5 $out/fuzz.log
