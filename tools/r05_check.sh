#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r05_check
timeout 1500 python -m pytest tests/test_tiny_gpu.py -x -q -m gpu > gpurun_out/r05_check/tests_tiny.log 2>&1; tail -3 gpurun_out/r05_check/tests_tiny.log
timeout 300 python bench.py --workload api1 > gpurun_out/r05_check/bench_api1.json 2>gpurun_out/r05_check/bench_api1.err; python -c "
import json; d=json.load(open('gpurun_out/r05_check/bench_api1.json')); print(d['legs'])"
bash tools/api1_trace.sh > gpurun_out/r05_check/api1_trace.log 2>&1; head -20 gpurun_out/api1_trace/timeline.txt
(time timeout 1800 python -m pytest tests -x -q -m gpu) > gpurun_out/r05_check/tests_full.log 2>&1; tail -5 gpurun_out/r05_check/tests_full.log
