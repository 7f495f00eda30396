#!/bin/bash
# round 4, last validation of the final tree: the whole GPU suite, smoke(), the default bench line, config 2 / middle / api4000 once more
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_final_d; mkdir -p $out
(time timeout 2400 python -m pytest tests -x -q -m gpu) > $out/tests_full.log 2>&1; echo "pytest rc=$?"; tail -5 $out/tests_full.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $out/smoke.log
timeout 600 python bench.py > $out/bench_default.json 2>$out/bench_default.err
for wl in config2 middle api4000 dual; do timeout 600 python bench.py --workload $wl > $out/bench_$wl.json 2>$out/bench_$wl.err; done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r04_final_d/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e); continue
    print(os.path.basename(f), round(d['value'] / 1e6, 3), d['ms_per_step'], d.get('split_ms_per_call'), (d.get('roofline') or {}).get('frac'), (d.get('cpu_baseline') or {}).get('value'), (d.get('parity') or {}))
PY
