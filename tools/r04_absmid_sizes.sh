#!/bin/bash
# from which batch size does the bit-sliced interior adapter scan pay?  (one wave per tile: a tile's rows are one sequential chain)
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_absmid_sizes; mkdir -p $out
B="timeout 600 python bench.py --no-host-inclusive --no-cpu-baseline --workload middle --steps 10 --warmup 2"
for n in 150000 250000 400000 600000 800000; do
  QCAT_HIP_MIDDLE_ABS_MIN=1 $B --reads $n > $out/abs_$n.json 2>/dev/null
  QCAT_HIP_MIDDLE_ABS_MIN=1 QCAT_HIP_MIDDLE_ABS_ONE_WAVE=0 $B --reads $n > $out/abs2w_$n.json 2>/dev/null
  QCAT_HIP_MIDDLE_NO_ABS=1 $B --reads $n > $out/f16_$n.json 2>/dev/null
done
python - <<'PY'
import json, glob, os
for n in (150000, 250000, 400000, 600000, 800000):
    row = []
    for k in ("abs", "abs2w", "f16"):
        try:
            d = json.loads(open('gpurun_out/r04_absmid_sizes/%s_%d.json' % (k, n)).read().strip().splitlines()[-1])
            row.append("%s %.3f ms (middle %.3f)" % (k, d['ms_per_step'], d['roofline']['kernels_avg_ms'].get('k_middle_packed', 0)))
        except Exception as e:
            row.append("%s failed" % k)
    print(n, " | ".join(row))
PY
