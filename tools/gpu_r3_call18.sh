#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c18
(timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_static_kernels.py tests/test_jit_gpu.py -x -q -m gpu) > gpurun_out/c18/tests.log 2>&1; tail -3 gpurun_out/c18/tests.log
timeout 300 python bench.py --no-host-inclusive --no-cpu-baseline --steps 10 > gpurun_out/c18/bench.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/c18/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernels_avg_ms'])"
timeout 300 python bench.py --workload dual --no-host-inclusive --no-cpu-baseline > gpurun_out/c18/bench_dual.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/c18/bench_dual.json')); print('dual', d['value'], d['ms_per_step'], d['roofline']['kernels_avg_ms'])"
timeout 300 python bench.py --workload config2 --no-host-inclusive --no-cpu-baseline > gpurun_out/c18/bench_config2.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/c18/bench_config2.json')); print('config2', d['value'], d['ms_per_step'], d['roofline']['kernels_avg_ms'])"
