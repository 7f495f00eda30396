#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c41
(time timeout 1500 python -m pytest tests/test_hip_fullsize.py -x -q -m gpu -k "config2 or medium") > gpurun_out/c41/tests.log 2>&1; tail -5 gpurun_out/c41/tests.log
