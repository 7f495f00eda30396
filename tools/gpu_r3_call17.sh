#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c17
(timeout 900 python -m pytest tests/test_batch_auto_gpu.py tests/test_scan_api_gpu.py tests/test_hip_parity.py -x -q -m gpu -k "auto or golden or api or batch") > gpurun_out/c17/tests.log 2>&1; tail -3 gpurun_out/c17/tests.log
timeout 300 python bench.py --workload api4000 --steps 2 --warmup 2 > gpurun_out/c17/bench_api4000.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/c17/bench_api4000.json')); print(d['value'], d['ms_per_step'], d['split_ms_per_call'], d['other_python_ms_per_call'])"
timeout 300 python bench.py --no-host-inclusive --no-cpu-baseline --steps 10 > gpurun_out/c17/bench.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/c17/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernels_avg_ms'])"
timeout 300 python bench.py --workload dual --no-host-inclusive --no-cpu-baseline > gpurun_out/c17/bench_dual.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/c17/bench_dual.json')); print('dual', d['value'], d['ms_per_step'], d['roofline']['kernels_avg_ms'])"
