#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c47
A=qcat_amd/csrc/build/ab
(timeout 1800 python -m pytest tests/test_hip_parity.py tests/test_static_kernels.py tests/test_batch_auto_gpu.py tests/test_hip_fuzz.py tests/test_jit.py tests/test_scan_api_gpu.py -x -q -m gpu) > gpurun_out/c47/tests.log 2>&1; tail -3 gpurun_out/c47/tests.log
bash tools/ab_run.sh gpurun_out/c47 2 --steps 8 -- $A/cur.so $A/hot4.so
bash tools/ab_run.sh gpurun_out/c47/config2 2 --workload config2 -- $A/cur.so $A/hot4.so
bash tools/ab_run.sh gpurun_out/c47/dual 2 --workload dual -- $A/cur.so $A/hot4.so
bash tools/ab_run.sh gpurun_out/c47/c3_125k 1 --workload config3 --reads 125000 -- $A/cur.so $A/hot4.so
bash tools/ab_run.sh gpurun_out/c47/c3_1m 1 --workload config3 --reads 1000000 -- $A/cur.so $A/hot4.so
