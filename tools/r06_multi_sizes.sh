#!/bin/bash
# the merged launches of small batches across batch sizes: named kits, resident batches (bench.py --reads), both ways
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_multi_sizes; mkdir -p $out
for w in config2 config3; do
for n in 64000 100000 150000; do
  for v in own 64; do
    if [ $v = own ]; then e="QCAT_HIP_NO_ADAPTER_MULTI=1 QCAT_HIP_NO_BARCODE_MULTI=1"; elif [ $v = one ]; then e="A=1"; else e="QCAT_X_BM=$v"; fi
    env $e python bench.py --workload $w --reads $n --steps 30 --warmup 5 --no-cpu-baseline > $out/$w.$n.$v.json 2> $out/$w.$n.$v.err
    python - <<PY
import json
d = json.load(open("$out/$w.$n.$v.json")); print("$w", $n, "$v", d["ms_per_step"])
PY
  done
done
done
