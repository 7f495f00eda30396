#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c12
(timeout 900 python -m pytest tests/test_hip_parity.py tests/test_batch_auto_gpu.py tests/test_cli_gpu.py -x -q -m gpu -k "bit_sliced_adapter or auto or config1 or ragged") > gpurun_out/c12/parity.log 2>&1; tail -4 gpurun_out/c12/parity.log
for v in fused separate; do
  if [ $v = separate ]; then export QCAT_HIP_NO_PACK_PLANES=1; fi
  timeout 300 python bench.py --no-host-inclusive --cpu-seconds 3 --steps 10 > gpurun_out/c12/bench_$v.json 2>/dev/null
  python - <<PY
import json
d=json.load(open('gpurun_out/c12/bench_$v.json'))
print('$v', d['value'], d['ms_per_step'], d['roofline']['kernels_avg_ms'], d.get('parity'))
PY
done
unset QCAT_HIP_NO_PACK_PLANES
(timeout 900 python -m pytest tests/test_hip_fullsize.py -x -q -m gpu -k "config3 or config4 or shards") > gpurun_out/c12/fullsize.log 2>&1; tail -4 gpurun_out/c12/fullsize.log
