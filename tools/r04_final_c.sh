#!/bin/bash
# round 4: after the last host-side changes (process-wide allocation generation, detect_barcode_batch without the unused count):
# the kit-auto / file-loop / API tests again, api4000, the default bench line
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_final_c; mkdir -p $out
timeout 1800 python -m pytest tests/test_batch_auto_gpu.py tests/test_cli_gpu.py tests/test_scan_api_gpu.py tests/test_comm_gpu.py -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
for i in 1 2; do timeout 600 python bench.py --workload api4000 > $out/api_$i.json 2>$out/api_$i.err; done
timeout 600 python bench.py > $out/bench_default.json 2>$out/bench_default.err
QCAT_BENCH_TMP=/dev/shm timeout 900 python tools/bench_auto_file.py 2000000 > $out/auto_file.json 2>$out/auto_file.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r04_final_c/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e); continue
    if 'value' in d:
        print(os.path.basename(f), round(d['value'] / 1e6, 3), d['ms_per_step'], d.get('split_ms_per_call'), d.get('other_python_ms_per_call'), (d.get('roofline') or {}).get('frac'), (d.get('cpu_baseline') or {}).get('value'))
    else:
        print(os.path.basename(f), {k: v.get('reads_per_s_warm') for k, v in d.items() if isinstance(v, dict)})
PY
