#!/bin/bash
# round 4: the kit-auto call as a captured graph -- parity of the replays, api4000 with and without
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_graph; mkdir -p $out
timeout 1200 python -m pytest tests/test_batch_auto_gpu.py -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $out/pytest.log
for i in 1 2; do
  timeout 600 python bench.py --workload api4000 > $out/api_graph_$i.json 2>$out/api_graph_$i.err
  QCAT_HIP_NO_GRAPH=1 timeout 600 python bench.py --workload api4000 > $out/api_plain_$i.json 2>$out/api_plain_$i.err
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r04_graph/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e); continue
    print(os.path.basename(f), round(d['value'] / 1e6, 3), d['ms_per_step'], d.get('split_ms_per_call'), d.get('other_python_ms_per_call'))
PY
