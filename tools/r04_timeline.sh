#!/bin/bash
# kernel timelines of one config-2 step (rocprofv3 --kernel-trace): with / without the tail split, left-over tiles beside / after
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r04_timeline; mkdir -p $out
for v in default nosplit noside; do
  env=""
  [ $v = nosplit ] && env="QCAT_HIP_BS_NO_TAIL_SPLIT=1"
  [ $v = noside ] && env="QCAT_HIP_LEFTOVER_SIDE=0"
  env $env rocprofv3 --kernel-trace -d /tmp/tl_$v -o t --output-format csv -- python $R/bench.py --workload config2 --steps 4 --warmup 2 --no-cpu-baseline --no-host-inclusive > $out/bench_$v.log 2>&1
  python $R/tools/step_timeline.py /tmp/tl_$v > $out/timeline_$v.txt 2>&1
  echo "== $v"; cat $out/timeline_$v.txt
done
