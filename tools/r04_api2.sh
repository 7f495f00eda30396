#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r04_api2; mkdir -p $out
cd $R; timeout 900 python -m pytest tests/test_batch_auto_gpu.py tests/test_scan_api_gpu.py -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/tl_api -o t --output-format csv -- python $R/bench.py --workload api4000 --steps 2 --warmup 1 > $out/bench_traced.log 2>&1
python $R/tools/api_timeline.py /tmp/tl_api > $out/timeline.txt 2>&1; cat $out/timeline.txt
