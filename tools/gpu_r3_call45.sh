#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c45
A=qcat_amd/csrc/build/ab
(timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_static_kernels.py tests/test_batch_auto_gpu.py tests/test_hip_fuzz.py tests/test_jit.py -x -q -m gpu) > gpurun_out/c45/tests.log 2>&1; tail -3 gpurun_out/c45/tests.log
bash tools/ab_run.sh gpurun_out/c45/config2 2 --workload config2 -- $A/cur.so $A/partial.so
bash tools/ab_run.sh gpurun_out/c45/dual 2 --workload dual -- $A/cur.so $A/partial.so
bash tools/ab_run.sh gpurun_out/c45 2 --steps 8 -- $A/cur.so $A/partial.so
bash tools/ab_run.sh gpurun_out/c45/c3_125k 1 --workload config3 --reads 125000 -- $A/cur.so $A/partial.so
bash tools/ab_run.sh gpurun_out/c45/c2_500k 1 --workload config2 --reads 500000 -- $A/cur.so $A/partial.so
