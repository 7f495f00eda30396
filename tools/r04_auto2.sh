#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_auto2; mkdir -p $out
QCAT_BENCH_TMP=/dev/shm timeout 900 python tools/bench_auto_file.py 1000000 > $out/auto_file.json 2>$out/auto_file.err; tail -3 $out/auto_file.err; cat $out/auto_file.json
QCAT_HIP_NO_GRAPH=1 QCAT_BENCH_TMP=/dev/shm timeout 900 python tools/bench_auto_file.py 1000000 > $out/auto_file_nograph.json 2>$out/auto_file_nograph.err; cat $out/auto_file_nograph.json
