#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c32
(QCAT_HIP_ABS_STAGES=4 timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "bit_sliced_adapter") > gpurun_out/c32/tests4.log 2>&1; tail -3 gpurun_out/c32/tests4.log
run() { # label, env..., -- bench args
  label=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-host-inclusive --no-cpu-baseline "$@" > gpurun_out/c32/$label.json 2>/dev/null
  python - gpurun_out/c32/$label.json $label <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); k = d['roofline']['kernels_avg_ms']
print(sys.argv[2], round(d['value'] / 1e6, 2), d['ms_per_step'], {x: round(v, 3) for x, v in k.items() if v > 0.2})
PY
}
for r in 1 2; do
run c3_s2_$r QCAT_HIP_ABS_STAGES=2 -- --steps 6
run c3_s4_$r QCAT_HIP_ABS_STAGES=4 -- --steps 6
done
for r in 1 2; do
run c2_b16_$r A=1 -- --workload config2
run c2_s2_$r QCAT_HIP_ADAPTER_BITSLICE_MIN=1 QCAT_HIP_ABS_STAGES=2 -- --workload config2
run c2_s4_$r QCAT_HIP_ADAPTER_BITSLICE_MIN=1 QCAT_HIP_ABS_STAGES=4 -- --workload config2
run c2_s4ns_$r QCAT_HIP_ADAPTER_BITSLICE_MIN=1 QCAT_HIP_ABS_STAGES=4 QCAT_HIP_ABS_NO_SPLIT=1 -- --workload config2
done
for r in 1 2; do
run dual_s2_$r QCAT_HIP_ABS_STAGES=2 -- --workload dual
run dual_s4_$r QCAT_HIP_ABS_STAGES=4 -- --workload dual
run dual_s4ns_$r QCAT_HIP_ABS_STAGES=4 QCAT_HIP_ABS_NO_SPLIT=1 -- --workload dual
done
