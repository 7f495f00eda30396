#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r05_stream${1:+_$1}
mkdir -p $out
nproc; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket|L3|L2" ; free -g | head -2
timeout 900 python tools/bench_stream.py 1000000 $out/bench_stream.json 2>&1 | tail -120
timeout 2400 python -m pytest tests/test_stream_gpu.py tests/test_cli_gpu.py tests/test_simple_gpu.py -x -q -m gpu > $out/tests.log 2>&1
tail -15 $out/tests.log
