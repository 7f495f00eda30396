#!/bin/bash
# round 3, GPU call 2: bit-sliced adapter kernels -- parity first, then speed
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c2
(time timeout 900 python -m pytest tests/test_hip_parity.py tests/test_batch_auto_gpu.py -x -q -m gpu -k "bit_sliced_adapter or auto") > gpurun_out/c2/parity.log 2>&1
tail -25 gpurun_out/c2/parity.log
timeout 300 python bench.py --no-host-inclusive --cpu-seconds 3 > gpurun_out/c2/bench_config3.json 2> gpurun_out/c2/bench_config3.err; tail -c 1500 gpurun_out/c2/bench_config3.err; python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/c2/bench_config3.json'))
    print(d['value'], d['ms_per_step'], d['roofline']['kernels_avg_ms'], d.get('parity'))
except Exception as e: print('bench parse failed', e)
PY
QCAT_HIP_NO_ADAPTER_BITSLICE=1 timeout 300 python bench.py --no-host-inclusive --no-cpu-baseline > gpurun_out/c2/bench_config3_noabs.json 2>/dev/null; python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/c2/bench_config3_noabs.json'))
    print('no abs:', d['value'], d['ms_per_step'], d['roofline']['kernels_avg_ms'])
except Exception as e: print('bench parse failed', e)
PY
timeout 300 python bench.py --workload config2 --no-host-inclusive --cpu-seconds 3 > gpurun_out/c2/bench_config2.json 2> gpurun_out/c2/bench_config2.err; python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/c2/bench_config2.json'))
    print('config2:', d['value'], d['ms_per_step'], d['roofline']['kernels_avg_ms'], d.get('parity'))
except Exception as e: print('bench parse failed', e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rp_c2 -o trace --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-inclusive --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/c2/trace.log 2>&1
find /tmp/rp_c2 -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/c2/kernel_stats.csv \;
head -12 $GRAFT_REPO_ROOT/gpurun_out/c2/kernel_stats.csv | cut -c1-200
