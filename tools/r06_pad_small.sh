#!/bin/bash
# the 4000-read call with its barcode jobs on padded bit-sliced super-tiles (QCAT_HIP_BITSLICE_PAD) against the binary16 kernels
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_pad_small; mkdir -p $out
run() {
  name=$1; shift
  env "$@" python bench.py --workload $W --steps 20 --warmup 3 > $out/$W.$name.json 2> $out/$W.$name.err
  python - <<PY
import json
d = json.load(open("$out/$W.$name.json")); print("$W", "$name", d["ms_per_step"], d.get("split_ms_per_call", {}).get("native_call_ms"))
PY
}
W=api4000
for i in 1 2 3; do run base A=1; run pad128 QCAT_HIP_BITSLICE_PAD=128; run pad1024 QCAT_HIP_BITSLICE_PAD=1024; done
