#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c28
A=qcat_amd/csrc/build/ab
(timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_batch_auto_gpu.py tests/test_scan_api_gpu.py tests/test_cli_gpu.py tests/test_simple_gpu.py -x -q -m gpu) > gpurun_out/c28/tests.log 2>&1; tail -3 gpurun_out/c28/tests.log
bash tools/ab_run.sh gpurun_out/c28 2 --steps 8 -- $A/cur.so $A/fin.so
bash tools/ab_run.sh gpurun_out/c28/config2 2 --workload config2 -- $A/cur.so $A/fin.so
bash tools/ab_run.sh gpurun_out/c28/dual 2 --workload dual -- $A/cur.so $A/fin.so
