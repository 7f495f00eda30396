#!/bin/bash
# round 3, GPU call 1: the new full-coverage tests, the default bench line, the API-shape bench, config-3 profile
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c1
(time timeout 1500 python -m pytest tests/test_hip_fullsize.py tests/test_cli_gpu.py tests/test_comm_gpu.py tests/test_hip_parity.py -x -q -m gpu) > gpurun_out/c1/tests.log 2>&1
tail -5 gpurun_out/c1/tests.log
timeout 600 python bench.py > gpurun_out/c1/bench_config3.json 2> gpurun_out/c1/bench_config3.err; tail -c 3000 gpurun_out/c1/bench_config3.json
timeout 600 python bench.py --workload config2 --cpu-seconds 4 > gpurun_out/c1/bench_config2.json 2> gpurun_out/c1/bench_config2.err
timeout 600 python bench.py --workload api4000 --steps 2 --warmup 2 > gpurun_out/c1/bench_api4000.json 2> gpurun_out/c1/bench_api4000.err; cat gpurun_out/c1/bench_api4000.json
timeout 900 bash tools/profile.sh r03a_config3 config3 --steps 3 --warmup 1 > gpurun_out/c1/profile.log 2>&1
tail -30 gpurun_out/c1/profile.log
