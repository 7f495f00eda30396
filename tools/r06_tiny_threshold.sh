#!/bin/bash
# where the one-wave-per-alignment kernels cross the throughput kernels after round 6's merged small-batch launches:
# resident batches; default / QCAT_HIP_NO_TINY=1 / QCAT_HIP_TINY_MAX_ENDS=4096 (forced while the batch has at most 4096 read ends)
cd $GRAFT_REPO_ROOT
for w in config3 config2; do
  for n in 16 32 64 100 150 200 300 500 800 1200; do
    line="$w $n"
    for v in default throughput one-wave; do
      if [ $v = throughput ]; then e="QCAT_HIP_NO_TINY=1"; elif [ $v = one-wave ]; then e="QCAT_HIP_TINY_MAX_ENDS=4096"; else e="A=1"; fi
      ms=$(env $e python bench.py --workload $w --reads $n --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
print(json.loads(sys.stdin.read())['ms_per_step'])")
      line="$line $v $ms"
    done
    echo $line
  done
done
