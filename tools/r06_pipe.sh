#!/bin/bash
# round 6 A/B: PIPE mode of k_bs_barcode (next unit staged beside the row loops) against the round-5 flow, same library, same box
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r06_pipe; mkdir -p $out
QCAT_HIP_BS_PIPE=1 timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -5
for r in 1 2; do
  for d in 0 1; do
    QCAT_HIP_BS_PIPE=$d timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-inclusive > $out/pipe${d}_$r.json 2>$out/pipe${d}_$r.err
    python - $out/pipe${d}_$r.json pipe$d <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    k = d['roofline']['kernels_avg_ms']
    print(sys.argv[2], round(d['value'] / 1e6, 2), d['ms_per_step'], {x: round(v, 3) for x, v in k.items() if v > 0.3})
except Exception as e:
    print(sys.argv[2], 'failed', e)
PY
  done
done
QCAT_HIP_BS_TRACE=1 QCAT_HIP_BS_PIPE=1 timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-inclusive > $out/trace_pipe1.json 2> $out/trace_pipe1.txt
QCAT_HIP_BS_PIPE=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-host-inclusive --cpu-seconds 5 > $out/pipe1_parity.json 2>$out/pipe1_parity.err; tail -c 600 $out/pipe1_parity.json
