#!/bin/bash
# round 4, call 8: the reference driver's call shape (api4000) with the device-side kit choice, the pointer hand-over and the
# C helper; parity of the kit-auto paths
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_api; mkdir -p $out
timeout 900 python -m pytest tests/test_batch_auto_gpu.py tests/test_cli_gpu.py tests/test_scan_api_gpu.py -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
for i in 1 2; do
  python bench.py --workload api4000 --steps 3 --warmup 1 > $out/api_new_$i.json 2>/dev/null
  QCAT_AMD_NO_PYGLUE=1 python bench.py --workload api4000 --steps 3 --warmup 1 > $out/api_nopyglue_$i.json 2>/dev/null
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r04_api/*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), round(d['value'] / 1e6, 3), d['ms_per_step'], d.get('split_ms_per_call'), d.get('other_python_ms_per_call'), d.get('python_helper'))
PY
