#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r05_api1
mkdir -p $out
timeout 2400 python -m pytest tests/test_scan_api_gpu.py tests/test_stream_gpu.py tests/test_batch_auto_gpu.py tests/test_hip_parity.py tests/test_cli_gpu.py -x -q -m gpu > $out/tests.log 2>&1
tail -6 $out/tests.log
timeout 600 python bench.py --workload api1 --steps 3 > $out/bench_api1.json 2> $out/bench_api1.err; tail -3 $out/bench_api1.err
python -c "
import json; d=json.load(open('$out/bench_api1.json')); print(json.dumps(d['legs']))"
