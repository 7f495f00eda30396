#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r06_r1; mkdir -p $out
timeout 1500 python -m pytest tests/test_r1_rule_gpu.py -x -q -m gpu 2>&1 | tail -15 > $out/r1_tests.txt; cat $out/r1_tests.txt
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_tiny_gpu.py tests/test_sg_align_gpu.py -x -q -m gpu 2>&1 | tail -5 > $out/parity_tests.txt; cat $out/parity_tests.txt
timeout 300 python bench.py --steps 10 --warmup 2 --no-host-inclusive --cpu-seconds 5 > $out/bench.json 2>$out/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r06_r1/bench.json')); print(d['value']/1e6, d['ms_per_step'], d['parity'], d['roofline']['kernels_avg_ms'])
PY
