#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c14
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --workload config2 --no-host-inclusive --no-cpu-baseline --steps 50 > gpurun_out/c14/$tag.json 2>/dev/null; python - <<PY
import json
d=json.load(open('gpurun_out/c14/$tag.json'))
print('$tag', d['value'], d['ms_per_step'], d['roofline']['kernels_avg_ms'])
PY
}
run default X=1
run static QCAT_HIP_BITSLICE_MIN=524288
run static_side QCAT_HIP_BITSLICE_MIN=524288 QCAT_HIP_BS_SIDE=1
run default2 X=1
