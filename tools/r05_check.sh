#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r05_check
timeout 1500 python -m pytest tests/test_tiny_gpu.py tests/test_scan_api_gpu.py -x -q -m gpu > gpurun_out/r05_check/tests_tiny.log 2>&1; tail -3 gpurun_out/r05_check/tests_tiny.log
for i in 1 2 3; do timeout 600 python -m pytest tests/test_tiny_gpu.py -x -q -m gpu 2>&1 | tail -1; done
timeout 300 python bench.py --workload api1 > gpurun_out/r05_check/bench_api1.json 2>gpurun_out/r05_check/bench_api1.err; python -c "
import json; d=json.load(open('gpurun_out/r05_check/bench_api1.json')); print(d['legs'])"
bash tools/api1_trace.sh > gpurun_out/r05_check/api1_trace.log 2>&1; sed -n 8,14p gpurun_out/api1_trace/timeline.txt
