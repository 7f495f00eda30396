#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c19
(timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_static_kernels.py -x -q -m gpu) > gpurun_out/c19/tests.log 2>&1; tail -3 gpurun_out/c19/tests.log
bash tools/ab_run.sh gpurun_out/c19 3 --steps 10 -- qcat_amd/csrc/build/ab/base.so qcat_amd/csrc/build/ab/deficit2.so
