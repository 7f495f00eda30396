#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c8
(time timeout 1700 python -m pytest tests -x -q -m gpu) > gpurun_out/c8/tests.log 2>&1; tail -8 gpurun_out/c8/tests.log
timeout 300 python bench.py --workload api4000 --steps 2 --warmup 2 > gpurun_out/c8/bench_api4000.json 2> gpurun_out/c8/bench_api4000.err; cat gpurun_out/c8/bench_api4000.json
timeout 600 python tools/bench_cli.py 2000000 50000 > gpurun_out/c8/bench_cli.json 2> gpurun_out/c8/bench_cli.err; cat gpurun_out/c8/bench_cli.json
