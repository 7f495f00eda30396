#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04e
B="timeout 600 python bench.py --no-host-inclusive --no-cpu-baseline --workload middle --steps 5 --warmup 1"
$B --reads 4000000 > gpurun_out/r04e/default_4M.json 2>gpurun_out/r04e/default_4M.err
QCAT_HIP_MIDDLE_NO_ABS=1 $B --reads 4000000 > gpurun_out/r04e/f16_4M.json 2>/dev/null
python - <<'PY'
import json
for k in ("default", "f16"):
    try:
        d = json.loads(open('gpurun_out/r04e/%s_4M.json' % k).read().strip().splitlines()[-1])
        print("4000000", k, d['ms_per_step'], round(d['value'] / 1e6, 1), d.get('parity'), d['roofline']['kernels_avg_ms'].get('k_middle_packed'))
    except Exception as e:
        print(k, "failed", e)
PY
timeout 600 python bench.py --workload middle > gpurun_out/r04e/bench_middle.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r04e/bench_middle.json').read().strip().splitlines()[-1]); print('middle', d['value'], d['ms_per_step'])"
bash tools/gpu_final_validation.sh
