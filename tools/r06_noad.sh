#!/bin/bash
# is the barcode phase of the 4000-read call its 150-row tiles (reads without an adapter)?  (no: profiles/r06_ab_multi.txt)
cd $GRAFT_REPO_ROOT
for f in 0.05 0.0 0.05 0.0 0.2; do
  python bench.py --workload api4000 --steps 20 --warmup 3 --no-adapter-fraction $f 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('no_adapter_fraction', '$f', d['ms_per_step'], d['split_ms_per_call']['native_call_ms'])"
done
