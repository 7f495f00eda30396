#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c10
(timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "bit_sliced_adapter") > gpurun_out/c10/parity.log 2>&1; tail -3 gpurun_out/c10/parity.log
for v in split nosplit off; do
  if [ $v = nosplit ]; then export QCAT_HIP_ABS_NO_SPLIT=1; fi
  if [ $v = off ]; then unset QCAT_HIP_ABS_NO_SPLIT; export QCAT_HIP_NO_ADAPTER_BITSLICE=1; fi
  timeout 300 python bench.py --workload config2 --no-host-inclusive --cpu-seconds 2 --steps 50 > gpurun_out/c10/bench_config2_$v.json 2>/dev/null
  python - <<PY
import json
d=json.load(open('gpurun_out/c10/bench_config2_$v.json'))
print('$v', d['value'], d['ms_per_step'], d['roofline']['kernels_avg_ms'], d.get('parity'))
PY
done
unset QCAT_HIP_NO_ADAPTER_BITSLICE
timeout 300 python bench.py --workload dual --no-host-inclusive --cpu-seconds 2 > gpurun_out/c10/bench_dual.json 2>/dev/null
python - <<PY
import json
d=json.load(open('gpurun_out/c10/bench_dual.json'))
print('dual', d['value'], d['ms_per_step'], d['roofline']['kernels_avg_ms'], d.get('parity'))
PY
