#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c49
(timeout 900 python -m pytest tests/test_hip_parity.py tests/test_static_kernels.py -x -q -m gpu -k "not fuzz") > gpurun_out/c49/tests.log 2>&1; tail -2 gpurun_out/c49/tests.log
for wl in config3 config2 dual; do for r in 1 2; do
timeout 300 python bench.py --workload $wl --no-host-inclusive --no-cpu-baseline > gpurun_out/c49/${wl}_$r.json 2>/dev/null
python - gpurun_out/c49/${wl}_$r.json $wl <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); k = d['roofline']['kernels_avg_ms']
print(sys.argv[2], round(d['value'] / 1e6, 2), d['ms_per_step'], {x: round(v, 3) for x, v in k.items() if v > 0.05})
PY
done; done
for n in 50000; do timeout 300 python bench.py --workload config3 --reads $n --no-host-inclusive --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c3', $n, d['value'], d['ms_per_step'], d['roofline']['kernels_avg_ms'].get('k_job_sort'))"; done
