#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c30
A=qcat_amd/csrc/build/ab
(timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_static_kernels.py tests/test_batch_auto_gpu.py tests/test_jit.py -x -q -m gpu) > gpurun_out/c30/tests.log 2>&1; tail -3 gpurun_out/c30/tests.log
bash tools/ab_run.sh gpurun_out/c30 2 --steps 8 -- $A/cur.so $A/direct.so
bash tools/ab_run.sh gpurun_out/c30/config2 2 --workload config2 -- $A/cur.so $A/direct.so
bash tools/ab_run.sh gpurun_out/c30/dual 2 --workload dual -- $A/cur.so $A/direct.so
QCAT_HIP_BS_FROM_TILES=1 QCAT_HIP_LIBRARY=$PWD/$A/direct.so python bench.py --no-host-inclusive --no-cpu-baseline --steps 8 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('direct build, tiles switch', d['value'], d['ms_per_step'], d['roofline']['kernels_avg_ms'])"
