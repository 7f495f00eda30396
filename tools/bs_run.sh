#!/bin/bash
# Quick A/B loop for the bit-sliced barcode kernels on the GPU box: the parity tests that force the path, then one
# bench line per workload with static letters and (config 3 / 2) with the letters from memory.
python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "bit_sliced or zero or summary" 2>&1 | tail -3
for w in config3 config2 dual dual96; do python bench.py --workload $w --steps 5 --warmup 2 --no-host-inclusive 2>gpurun_out/bs_$w.err | tail -1 > gpurun_out/bs_$w.json; python -c "
import json
d=json.load(open('gpurun_out/bs_$w.json')); print('$w', d['value'], d['ms_per_step'], d['parity'], d['roofline']['kernels_avg_ms'].get('k_barcode_bitslice'))"; done
for w in config3 config2; do QCAT_HIP_NO_BS_STATIC=1 python bench.py --workload $w --steps 5 --warmup 2 --no-host-inclusive --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bsd_$w.json; python -c "
import json
d=json.load(open('gpurun_out/bsd_$w.json')); print('$w letters from memory', d['value'], d['ms_per_step'], d['roofline']['kernels_avg_ms'].get('k_barcode_bitslice'))"; done
