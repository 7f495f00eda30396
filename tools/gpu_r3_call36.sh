#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c36
(timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_batch_auto_gpu.py tests/test_scan_api_gpu.py -x -q -m gpu) > gpurun_out/c36/tests.log 2>&1; tail -3 gpurun_out/c36/tests.log
for wl in config2 dual config3; do for r in 1 2; do
timeout 300 python bench.py --workload $wl --no-host-inclusive --no-cpu-baseline > gpurun_out/c36/${wl}_$r.json 2>/dev/null
python - gpurun_out/c36/${wl}_$r.json $wl <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); k = d['roofline']['kernels_avg_ms']
print(sys.argv[2], round(d['value'] / 1e6, 2), d['ms_per_step'], {x: round(v, 3) for x, v in k.items() if v > 0.2})
PY
done; done
timeout 300 python bench.py --workload config2 --reads 500000 --no-host-inclusive --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('config2 500k', d['value'], d['ms_per_step'], d['roofline']['kernels_avg_ms'])"
QCAT_HIP_NO_ADAPTER_BITSLICE=1 timeout 300 python bench.py --workload config2 --reads 500000 --no-host-inclusive --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('config2 500k b16', d['value'], d['ms_per_step'], d['roofline']['kernels_avg_ms'])"
timeout 300 python bench.py --workload config2 --reads 800000 --no-host-inclusive --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('config2 800k', d['value'], d['ms_per_step'], d['roofline']['kernels_avg_ms'])"
QCAT_HIP_NO_ADAPTER_BITSLICE=1 timeout 300 python bench.py --workload config2 --reads 800000 --no-host-inclusive --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('config2 800k b16', d['value'], d['ms_per_step'], d['roofline']['kernels_avg_ms'])"
