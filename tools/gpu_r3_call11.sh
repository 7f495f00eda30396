#!/bin/bash
# round 3: the committed profiles (rocprofv3 stats + PMC) of configs 3, 2 and the dual kit with the round's final kernels
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c11
timeout 900 bash tools/profile.sh r03_config3 config3 --steps 3 --warmup 1 > gpurun_out/c11/profile3.log 2>&1
timeout 600 bash tools/profile.sh r03_config2 config2 --steps 10 --warmup 2 > gpurun_out/c11/profile2.log 2>&1
timeout 600 bash tools/profile.sh r03_dual dual --steps 5 --warmup 2 > gpurun_out/c11/profiled.log 2>&1
for w in config3 config2 dual dual96 middle; do
  timeout 600 python bench.py --workload $w --cpu-seconds 4 > gpurun_out/c11/bench_$w.json 2> gpurun_out/c11/bench_$w.err
  python - <<PY
import json
try:
    d=json.load(open('gpurun_out/c11/bench_$w.json'))
    print('$w', d['value'], d['ms_per_step'], d['roofline']['kernels_avg_ms'], d.get('parity'), (d.get('host_inclusive') or {}).get('value'))
except Exception as e: print('$w failed', e)
PY
done
QCAT_HIP_NO_ADAPTER_BITSLICE=1 timeout 300 python bench.py --workload dual --no-host-inclusive --no-cpu-baseline > gpurun_out/c11/bench_dual_noabs.json 2>/dev/null
QCAT_HIP_ADAPTER_BITSLICE_MIN=1 timeout 300 python bench.py --workload dual --no-host-inclusive --no-cpu-baseline > gpurun_out/c11/bench_dual_abs.json 2>/dev/null
python - <<PY
import json
for v in ('noabs','abs'):
    d=json.load(open('gpurun_out/c11/bench_dual_%s.json'%v)); print('dual', v, d['value'], d['roofline']['kernels_avg_ms'])
PY
