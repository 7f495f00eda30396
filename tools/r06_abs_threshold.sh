#!/bin/bash
# where the bit-sliced adapter scan crosses the binary16 kernels: resident batches; default / QCAT_HIP_ADAPTER_BITSLICE_MIN=1 (forced) /
# QCAT_HIP_NO_ADAPTER_BITSLICE=1 (binary16)
cd $GRAFT_REPO_ROOT
for w in ${WORKLOADS:-config3 config2}; do
  if [ $w = config2 ]; then sizes="200000 400000 600000 800000 1000000 1500000"; else sizes="100000 200000 300000 400000 600000 900000"; fi
  for n in $sizes; do
    line="$w $n"
    for v in default forced binary16; do
      if [ $v = forced ]; then e="QCAT_HIP_ADAPTER_BITSLICE_MIN=1"; elif [ $v = binary16 ]; then e="QCAT_HIP_NO_ADAPTER_BITSLICE=1"; else e="A=1"; fi
      ms=$(env $e python bench.py --workload $w --reads $n --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
print(json.loads(sys.stdin.read())['ms_per_step'])")
      line="$line $v $ms"
    done
    echo $line
  done
done
