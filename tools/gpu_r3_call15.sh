#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c15
(time timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_hip_fuzz.py tests/test_batch_auto_gpu.py tests/test_static_kernels.py tests/test_scan_api_gpu.py tests/test_cli_gpu.py tests/test_jit.py -x -q -m gpu) > gpurun_out/c15/tests.log 2>&1; tail -6 gpurun_out/c15/tests.log
for v in slim noslim; do
  if [ $v = noslim ]; then export QCAT_HIP_NO_SLIM=1; fi
  timeout 300 python bench.py --no-host-inclusive --cpu-seconds 3 --steps 10 > gpurun_out/c15/bench_$v.json 2>/dev/null
  python - <<PY
import json
d=json.load(open('gpurun_out/c15/bench_$v.json'))
print('$v', d['value'], d['ms_per_step'], d['roofline']['kernels_avg_ms'], d.get('parity'))
PY
done
unset QCAT_HIP_NO_SLIM
(timeout 900 python -m pytest tests/test_hip_fullsize.py -x -q -m gpu) > gpurun_out/c15/fullsize.log 2>&1; tail -3 gpurun_out/c15/fullsize.log
