#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c9
(time timeout 900 python -m pytest tests/test_sg_align_gpu.py tests/test_abi.py tests/test_simple_gpu.py -x -q) > gpurun_out/c9/tests.log 2>&1; tail -15 gpurun_out/c9/tests.log
