#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r05_check
timeout 1500 python -m pytest tests/test_stream_gpu.py tests/test_cli_gpu.py tests/test_fastq_native.py tests/test_batch_auto_gpu.py -x -q -m gpu > gpurun_out/r05_check/tests_stream.log 2>&1; tail -3 gpurun_out/r05_check/tests_stream.log
(time timeout 600 python bench.py > gpurun_out/r05_check/bench_default.json 2> gpurun_out/r05_check/bench_default.err) 2>&1 | grep real
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_check/bench_default.json"))
f = d["host_inclusive"]["from_fastq"]
print(d["value"], d["ms_per_step"], f["value"], f["stream"], f["whole_file"]["value"])
PY
timeout 300 python bench.py --workload api4000 > gpurun_out/r05_check/bench_api4000.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r05_check/bench_api4000.json')); print(d['value'], d['ms_per_step'], d['split_ms_per_call'])"
