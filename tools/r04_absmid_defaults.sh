#!/bin/bash
# the size-dependent defaults of the bit-sliced interior adapter scan against the binary16 kernel
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_absmid_defaults; mkdir -p $out
timeout 1200 python -m pytest tests -x -q -m gpu -k "middle" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $out/pytest.log
B="timeout 600 python bench.py --no-host-inclusive --no-cpu-baseline --workload middle --steps 10 --warmup 2"
for n in 250000 400000 600000 800000 1000000; do
  $B --reads $n > $out/default_$n.json 2>/dev/null
  QCAT_HIP_MIDDLE_NO_ABS=1 $B --reads $n > $out/f16_$n.json 2>/dev/null
done
python - <<'PY'
import json
for n in (250000, 400000, 600000, 800000, 1000000):
    row = []
    for k in ("default", "f16"):
        try:
            d = json.loads(open('gpurun_out/r04_absmid_defaults/%s_%d.json' % (k, n)).read().strip().splitlines()[-1])
            row.append("%s %.3f ms = %.1f M reads/s (middle %.3f)" % (k, d['ms_per_step'], d['value'] / 1e6, d['roofline']['kernels_avg_ms'].get('k_middle_packed', 0)))
        except Exception as e:
            row.append("%s failed" % k)
    print(n, " | ".join(row))
PY
