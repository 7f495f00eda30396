#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c6
(timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "bit_sliced_adapter") > gpurun_out/c6/parity.log 2>&1; tail -3 gpurun_out/c6/parity.log
timeout 300 python bench.py --no-host-inclusive --no-cpu-baseline --steps 10 > gpurun_out/c6/bench.json 2>/dev/null
python - <<PY
import json
d=json.load(open('gpurun_out/c6/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['kernels_avg_ms'])
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rp_c6 -o trace --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-inclusive --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/c6/trace.log 2>&1
find /tmp/rp_c6 -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/c6/kernel_stats.csv \;
grep -E "k_abs_planes|k_adapter_bs|k_pack" $GRAFT_REPO_ROOT/gpurun_out/c6/kernel_stats.csv | cut -c1-40,150-260
