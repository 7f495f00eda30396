#!/bin/bash
# round 4: blocks of k_finalize (every block flushes its LDS histogram with global atomics)
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_fin; mkdir -p $out
B="python bench.py --no-host-inclusive --no-cpu-baseline"
for fb in 4096 2048 1024 512 256; do
  for i in 1 2; do QCAT_HIP_FIN_BLOCKS=$fb $B --workload config2 --steps 20 --warmup 3 > $out/c2_fb${fb}_$i.json 2>/dev/null; done
  QCAT_HIP_FIN_BLOCKS=$fb $B --workload config3 --steps 5 --warmup 2 > $out/c3_fb${fb}.json 2>/dev/null
  QCAT_HIP_FIN_BLOCKS=$fb $B --workload dual --steps 10 --warmup 2 > $out/dual_fb${fb}.json 2>/dev/null
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r04_fin/*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    k = (d.get('roofline') or {}).get('kernels_avg_ms') or {}
    print(os.path.basename(f), round(d['value'] / 1e6, 3), d['ms_per_step'], 'finalize', round(k.get('k_finalize', 0), 4), 'select', round(k.get('k_barcode_select', 0), 4))
PY
