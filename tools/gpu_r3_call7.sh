#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c7
(timeout 900 python -m pytest tests/test_hip_parity.py tests/test_cli_gpu.py -x -q -m gpu -k "bit_sliced_adapter or cli or config1") > gpurun_out/c7/tests.log 2>&1; tail -5 gpurun_out/c7/tests.log
timeout 300 python bench.py --no-host-inclusive --no-cpu-baseline --steps 10 > gpurun_out/c7/bench.json 2>gpurun_out/c7/bench.err
python - <<PY
import json
d=json.load(open('gpurun_out/c7/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['kernels_avg_ms'])
PY
timeout 600 python tools/bench_cli.py 2000000 100000 > gpurun_out/c7/bench_cli.json 2> gpurun_out/c7/bench_cli.err; cat gpurun_out/c7/bench_cli.json; tail -5 gpurun_out/c7/bench_cli.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rp_c7 -o trace --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-inclusive --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/c7/trace.log 2>&1
find /tmp/rp_c7 -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/c7/kernel_stats.csv \;
grep -E "k_abs_planes|k_adapter_bs|k_pack" $GRAFT_REPO_ROOT/gpurun_out/c7/kernel_stats.csv | awk -F'",' '{print substr($1,1,50), $2}' | cut -c1-140
