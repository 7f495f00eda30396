#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c39
A=qcat_amd/csrc/build/ab
(timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "bit_sliced or golden or zero") > gpurun_out/c39/tests.log 2>&1; tail -3 gpurun_out/c39/tests.log
bash tools/ab_run.sh gpurun_out/c39 2 --steps 8 -- $A/cur.so $A/win2.so
bash tools/ab_run.sh gpurun_out/c39/config2 2 --workload config2 -- $A/cur.so $A/win2.so
bash tools/ab_run.sh gpurun_out/c39/dual 2 --workload dual -- $A/cur.so $A/win2.so
