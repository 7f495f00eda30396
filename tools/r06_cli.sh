#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r06_cli; mkdir -p $out
timeout 1200 python -m pytest tests/test_cli_gpu.py tests/test_stream_gpu.py tests/test_fastq_native.py -x -q -m gpu 2>&1 | tail -6
QCAT_HIP_PIPELINE_TRACE=1 timeout 600 python tools/bench_cli.py 1000000 100000 > $out/bench_cli.json 2>$out/bench_cli.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r06_cli/bench_cli.json'))
for k in ('kit_PBC096_100000_reads','outputs_identical','outputs_identical_with_flags','native_tsv','native_per_barcode_fastq'):
    print(k, d.get(k))
PY
grep "writers:" $out/bench_cli.err | tail -4
