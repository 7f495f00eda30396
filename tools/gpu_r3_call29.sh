#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c29
A=qcat_amd/csrc/build/ab
(timeout 900 python -m pytest tests/test_hip_parity.py tests/test_scan_api_gpu.py -x -q -m gpu -k "golden or dual or counts or api or middle") > gpurun_out/c29/tests.log 2>&1; tail -3 gpurun_out/c29/tests.log
bash tools/ab_run.sh gpurun_out/c29 2 --steps 8 -- $A/cur.so $A/dyn.so
bash tools/ab_run.sh gpurun_out/c29/dual 2 --workload dual -- $A/cur.so $A/dyn.so
