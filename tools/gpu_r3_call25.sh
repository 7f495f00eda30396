#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c25
(timeout 900 python -m pytest tests/test_cli_gpu.py tests/test_scan_api_gpu.py -x -q -m gpu) > gpurun_out/c25/tests.log 2>&1; tail -3 gpurun_out/c25/tests.log
timeout 900 python tools/bench_cli.py 6000000 20000 > gpurun_out/c25/bench_cli6m.json 2> gpurun_out/c25/bench_cli6m.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/c25/bench_cli6m.json'))
for k in ('outputs_identical','ingest','native_tsv','native_per_barcode_fastq'): print(k, d[k])
PY
timeout 600 python tools/prof_cli.py 6000000 > gpurun_out/c25/prof_cli.txt 2>&1; grep -A24 "run 1" gpurun_out/c25/prof_cli.txt | cut -c1-150
