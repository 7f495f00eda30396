#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c37
run() { label=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-host-inclusive --no-cpu-baseline "$@" > gpurun_out/c37/$label.json 2>/dev/null
  python - gpurun_out/c37/$label.json $label <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); k = d['roofline']['kernels_avg_ms']
print(sys.argv[2], round(d['value'] / 1e6, 2), d['ms_per_step'], {x: round(v, 3) for x, v in k.items() if v > 0.05})
PY
}
for n in 500000 250000 125000 62500; do
run c2_${n}_def A=1 -- --workload config2 --reads $n
run c2_${n}_bs QCAT_HIP_BITSLICE_MIN=1 -- --workload config2 --reads $n
done
for n in 250000 125000 62500 31250; do
run c3_${n}_def A=1 -- --workload config3 --reads $n
run c3_${n}_bs QCAT_HIP_BITSLICE_MIN=1 -- --workload config3 --reads $n
done
