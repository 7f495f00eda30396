#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r05_check
timeout 1500 python -m pytest tests -x -q -m gpu -k "unique or middle or comm or stream or cli or fastq or api" > gpurun_out/r05_check/tests.log 2>&1; tail -3 gpurun_out/r05_check/tests.log
for wl in api4000 middle; do timeout 600 python bench.py --workload $wl > gpurun_out/r05_check/bench_$wl.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r05_check/bench_$wl.json')); print('$wl', d['value'], d['ms_per_step'], d.get('split_ms_per_call'))"; done
bash tools/stream_diag.sh
