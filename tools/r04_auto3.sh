#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_auto3; mkdir -p $out
timeout 1500 python -m pytest tests/test_cli_gpu.py tests/test_batch_auto_gpu.py -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $out/pytest.log
QCAT_BENCH_TMP=/dev/shm timeout 900 python tools/bench_auto_file.py 2000000 > $out/auto_file.json 2>$out/auto_file.err; tail -3 $out/auto_file.err; cat $out/auto_file.json
QCAT_BENCH_TMP=/dev/shm timeout 1500 python tools/bench_cli.py 3000000 20000 > $out/bench_cli.json 2>$out/bench_cli.err; tail -3 $out/bench_cli.err; cat $out/bench_cli.json
