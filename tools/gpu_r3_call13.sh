#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c13
(timeout 900 python -m pytest tests/test_hip_parity.py tests/test_batch_auto_gpu.py -x -q -m gpu -k "bit_sliced_adapter or auto") > gpurun_out/c13/parity.log 2>&1; tail -4 gpurun_out/c13/parity.log
timeout 300 python bench.py --no-host-inclusive --cpu-seconds 3 --steps 10 > gpurun_out/c13/bench.json 2>/dev/null
python - <<PY
import json
d=json.load(open('gpurun_out/c13/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['kernels_avg_ms'], d.get('parity'))
PY
timeout 300 python bench.py --workload dual --no-host-inclusive --cpu-seconds 2 > gpurun_out/c13/bench_dual.json 2>/dev/null
python - <<PY
import json
d=json.load(open('gpurun_out/c13/bench_dual.json'))
print('dual', d['value'], d['roofline']['kernels_avg_ms'], d.get('parity'))
PY
(timeout 900 python -m pytest tests/test_hip_fullsize.py -x -q -m gpu -k "config3") > gpurun_out/c13/fullsize.log 2>&1; tail -3 gpurun_out/c13/fullsize.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rp_c13 -o trace --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-inclusive --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/c13/trace.log 2>&1
find /tmp/rp_c13 -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/c13/kernel_stats.csv \;
grep -E "k_abs_planes|k_adapter_bs|k_pack" $GRAFT_REPO_ROOT/gpurun_out/c13/kernel_stats.csv | awk -F'",' '{print substr($1,1,50), $2}' | cut -c1-140
