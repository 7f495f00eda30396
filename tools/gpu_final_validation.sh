#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
(time timeout 600 python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err) 2>&1 | grep real
python -c "
import json; d=json.load(open('gpurun_out/final/bench_default.json')); print(d['metric'], d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], d['host_inclusive']['value'], d['host_inclusive']['from_fastq']['value'])
print({k:(v['ms'],v['issue_util']) for k,v in d['valu_issue']['marks'].items() if v.get('ms',0)>0.3})"
(time timeout 1500 python -m pytest tests -x -q -m gpu) > gpurun_out/final/tests_full.log 2>&1; tail -4 gpurun_out/final/tests_full.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
