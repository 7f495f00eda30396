#!/bin/bash
# BS_PAD_ROWS 5 -> 12: parity of the front-padded units, then the resident workloads
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_pad12; mkdir -p $out
timeout 1200 python -m pytest tests/test_hip_parity.py -q -x -k "padded_at_the_front or bit_sliced or bitslice" 2>&1 | tail -5 > $out/tests.txt
for w in dual config3 config2 dual96; do
  python bench.py --workload $w --no-cpu-baseline > $out/bench_$w.json 2> $out/bench_$w.err
  python - <<PY
import json
d = json.load(open("$out/bench_$w.json"))
print("$w", d["ms_per_step"], d["value"], d.get("parity", d.get("parity_check")))
PY
done
cat $out/tests.txt
