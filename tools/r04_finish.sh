#!/bin/bash
# round 4: blocks of k_adapter_finish (every block flushes its job histogram with global atomics); k_job_offsets through the LDS
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_finish; mkdir -p $out
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "golden or bit_sliced" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
B="python bench.py --no-host-inclusive --no-cpu-baseline"
for fb in 2048 1024 512 256; do
  for i in 1 2; do QCAT_HIP_FINISH_BLOCKS=$fb $B --workload config2 --steps 20 --warmup 3 > $out/c2_fb${fb}_$i.json 2>/dev/null; done
  QCAT_HIP_FINISH_BLOCKS=$fb $B --workload config3 --steps 5 --warmup 2 > $out/c3_fb${fb}.json 2>/dev/null
  QCAT_HIP_FINISH_BLOCKS=$fb $B --workload dual --steps 10 --warmup 2 > $out/dual_fb${fb}.json 2>/dev/null
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r04_finish/*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    k = (d.get('roofline') or {}).get('kernels_avg_ms') or {}
    print(os.path.basename(f), round(d['value'] / 1e6, 3), d['ms_per_step'], 'job_sort', round(k.get('k_job_sort', 0), 4), 'adapter', round(k.get('k_adapter_bitslice', 0), 4))
PY
