#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r05_check
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
(time timeout 1800 python -m pytest tests -x -q -m gpu) > gpurun_out/r05_check/tests_full.log 2>&1; tail -5 gpurun_out/r05_check/tests_full.log
timeout 300 python bench.py --workload api1 > gpurun_out/r05_check/bench_api1.json 2>gpurun_out/r05_check/bench_api1.err; python -c "
import json; d=json.load(open('gpurun_out/r05_check/bench_api1.json')); print(d['legs'])"
timeout 300 python bench.py --workload api4000 > gpurun_out/r05_check/bench_api4000.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r05_check/bench_api4000.json')); print(d['value'], d['ms_per_step'], d['split_ms_per_call'])"
