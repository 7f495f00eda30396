#!/bin/bash
# round 5: the barcode's last letter as variant columns of the trailing pass -- parity, then A/B against the build before it
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r05_var
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "bit_sliced or front" > gpurun_out/r05_var/parity.log 2>&1
tail -4 gpurun_out/r05_var/parity.log
bash tools/ab_run.sh gpurun_out/r05_var/config3 2 --steps 10 -- qcat_amd/csrc/build/ab/novar.so qcat_amd/csrc/libqcat_hip.so
bash tools/ab_run.sh gpurun_out/r05_var/dual 2 --workload dual --steps 20 -- qcat_amd/csrc/build/ab/novar.so qcat_amd/csrc/libqcat_hip.so
bash tools/ab_run.sh gpurun_out/r05_var/config2 2 --workload config2 --steps 30 -- qcat_amd/csrc/build/ab/novar.so qcat_amd/csrc/libqcat_hip.so
