#!/bin/bash
# where the bit-sliced barcode path crosses the binary16 kernels (round 3 put the switch at 70 000 + 3 500 000 / barcodes jobs):
# resident batches of config 3 (PBC096, 2 jobs per read) and config 2 (12 barcodes, 1 job per read), forced onto the path against the default
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_bs_threshold; mkdir -p $out
for w in ${WORKLOADS:-config3 config2}; do
  if [ $w = config3 ]; then sizes="8000 12000 16000 24000 32000 40000 48000 56000"; elif [ $w = dual ]; then sizes="8000 12000 16000 24000 32000"; else sizes="100000 150000 200000 250000 300000 350000 400000"; fi
  for n in $sizes; do
    for v in default forced binary16; do
      if [ $v = forced ]; then e="QCAT_HIP_BITSLICE_MIN=2048"; elif [ $v = binary16 ]; then e="QCAT_HIP_NO_BITSLICE=1"; else e="A=1"; fi
      env $e python bench.py --workload $w --reads $n --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w', $n, '$v', d['ms_per_step'])"
    done
  done
done
