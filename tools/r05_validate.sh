#!/bin/bash
# round 5: the whole GPU suite, the bit-sliced fuzz sweep (custom shapes through hipRTC) and the default bench line on one box
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r05_validate${1:+_$1}
mkdir -p $out
timeout 1500 python -m pytest tests -x -q -m gpu > $out/gpu_suite.log 2>&1
tail -3 $out/gpu_suite.log
QCAT_HIP_BITSLICE_MIN=2048 timeout 900 python tools/fuzz_bitslice.py 0 ${FUZZ_N:-60} > $out/fuzz_bitslice.txt 2>&1
tail -2 $out/fuzz_bitslice.txt
timeout 600 python bench.py > $out/bench_config3.json 2> $out/bench_config3.err
python -c "
import json,sys
d=json.load(open('$out/bench_config3.json'))
print(d['value'], d['ms_per_step'], d.get('parity'), d['roofline']['frac'])
"
