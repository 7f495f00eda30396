#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_quick; mkdir -p $out
timeout 1800 python -m pytest tests/test_batch_auto_gpu.py tests/test_cli_gpu.py tests/test_scan_api_gpu.py -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
