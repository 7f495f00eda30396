#!/bin/bash
# round 4: the kit-auto file loop on several contexts -- parity, then the driver at size (6 M reads) with one worker and four
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_auto; mkdir -p $out
timeout 1500 python -m pytest tests/test_cli_gpu.py tests/test_batch_auto_gpu.py -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $out/pytest.log
QCAT_BENCH_TMP=/dev/shm timeout 1500 python tools/bench_cli.py 3000000 20000 > $out/bench_cli.json 2>$out/bench_cli.err; tail -3 $out/bench_cli.err; cat $out/bench_cli.json
