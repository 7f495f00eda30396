#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c24
(timeout 1500 python -m pytest tests/test_cli_gpu.py tests/test_scan_api_gpu.py tests/test_batch_auto_gpu.py tests/test_hip_parity.py tests/test_static_kernels.py -x -q -m gpu) > gpurun_out/c24/tests.log 2>&1; tail -3 gpurun_out/c24/tests.log
QCAT_HIP_PIPELINE_TRACE=1 timeout 900 python tools/bench_cli.py 6000000 20000 > gpurun_out/c24/bench_cli6m.json 2> gpurun_out/c24/bench_cli6m.err; grep "qcat pipeline" gpurun_out/c24/bench_cli6m.err | tail -8 | cut -c1-300
python - <<'PY'
import json
d=json.load(open('gpurun_out/c24/bench_cli6m.json'))
for k in ('outputs_identical','ingest','native_tsv','native_per_barcode_fastq'): print(k, d[k])
PY
timeout 600 python bench.py --steps 10 > gpurun_out/c24/bench.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/c24/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernels_avg_ms'], d.get('host_inclusive'))"
