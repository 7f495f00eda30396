#!/bin/bash
# round 5: the streamed file loop -- its GPU tests, the flags on the native path, and the bench line's from_fastq leg
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r05_stream${1:+_$1}
mkdir -p $out
timeout 2400 python -m pytest tests/test_stream_gpu.py tests/test_cli_gpu.py tests/test_simple_gpu.py -x -q -m gpu > $out/tests.log 2>&1
tail -15 $out/tests.log
timeout 900 python bench.py --steps 5 --no-cpu-baseline > $out/bench_config3.json 2> $out/bench_config3.err
python -c "
import json
d=json.load(open('$out/bench_config3.json'))
print(d['value'], d['ms_per_step'])
print(json.dumps(d['host_inclusive']['from_fastq'], indent=1))
"
