#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r05_stream${1:+_$1}
mkdir -p $out
cat /sys/fs/cgroup/cpu.max 2>/dev/null
timeout 2400 python -m pytest tests/test_stream_gpu.py tests/test_cli_gpu.py tests/test_simple_gpu.py -x -q -m gpu > $out/tests.log 2>&1
tail -15 $out/tests.log
