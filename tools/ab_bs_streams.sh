#!/bin/bash
# A/B of the launch arrangement of the bit-sliced barcode phase on the GPU box (run through gpurun):
# the per-family launches in line on the context's stream (QCAT_HIP_BS_SERIAL=1) or on side streams (default for batches
# of >= 16 super-tiles per CU).  Results of round 2: profiles/r02_ab_bs_streams.txt.
out=gpurun_out/ab_bs
mkdir -p $out
run() { # tag env workload
  env $2 timeout 150 python bench.py --workload $3 --no-cpu-baseline --no-host-inclusive > $out/$1.json 2> $out/$1.err
}
for i in 1 2; do
run c3_side_$i X=1 config3
run c3_ser_$i QCAT_HIP_BS_SERIAL=1 config3
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/ab_bs/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[0]);k=d["roofline"]["kernels_avg_ms"]
    print(f.split('/')[-1],round(d["value"]/1e6,2),"M reads/s",d["ms_per_step"],"ms/step, bit-sliced phase",k.get("k_barcode_bitslice"),"ms")
PY
