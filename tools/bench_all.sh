#!/bin/bash
# Round snapshot on the GPU box: one bench line per workload -> gpurun_out/bench_<workload>.json
mkdir -p gpurun_out
python bench.py --workload config2 --steps 20 --warmup 3 2>/dev/null | tail -1 > gpurun_out/bench_config2.json
python bench.py --workload config3 --steps 10 --warmup 2 2>/dev/null | tail -1 > gpurun_out/bench_config3.json
python bench.py --workload dual --steps 10 --warmup 2 2>/dev/null | tail -1 > gpurun_out/bench_dual.json
python bench.py --workload dual96 --steps 10 --warmup 2 2>/dev/null | tail -1 > gpurun_out/bench_dual96.json
python bench.py --workload middle --steps 10 --warmup 2 2>/dev/null | tail -1 > gpurun_out/bench_middle.json
python bench.py --workload config3 --reads 12500000 --steps 3 --warmup 1 --no-host-inclusive 2>/dev/null | tail -1 > gpurun_out/bench_config4_shard.json
for f in config2 config3 dual dual96 middle config4_shard; do python -c "
import json
d=json.load(open('gpurun_out/bench_$f.json')); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['valu']['frac_of_valu_peak'], d['cpu_baseline']['value'], d['parity'], d.get('host_inclusive',{}).get('value'))"; done
python tools/bench_auto.py 1000000 2>&1 | tail -2
