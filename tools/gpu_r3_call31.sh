#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c31
A=qcat_amd/csrc/build/ab
bash tools/ab_run.sh gpurun_out/c31 2 --steps 8 -- $A/cur.so $A/direct.so $A/direct2.so
bash tools/ab_run.sh gpurun_out/c31/config2 2 --workload config2 -- $A/cur.so $A/direct2.so
bash tools/ab_run.sh gpurun_out/c31/dual 2 --workload dual -- $A/cur.so $A/direct2.so
(timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "bit_sliced or golden or zero") > gpurun_out/c31/tests.log 2>&1; tail -3 gpurun_out/c31/tests.log
