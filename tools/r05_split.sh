#!/bin/bash
# round 5: the split barcode DP (both contexts shared) -- parity first, then A/B against the round-4 library on one box
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r05_split
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "bit_sliced" > gpurun_out/r05_split/parity.log 2>&1
tail -5 gpurun_out/r05_split/parity.log
bash tools/ab_run.sh gpurun_out/r05_split/config3 2 --steps 10 -- qcat_amd/csrc/build/ab/r4.so qcat_amd/csrc/libqcat_hip.so
bash tools/ab_run.sh gpurun_out/r05_split/dual 2 --workload dual --steps 20 -- qcat_amd/csrc/build/ab/r4.so qcat_amd/csrc/libqcat_hip.so
bash tools/ab_run.sh gpurun_out/r05_split/config2 2 --workload config2 --steps 30 -- qcat_amd/csrc/build/ab/r4.so qcat_amd/csrc/libqcat_hip.so
