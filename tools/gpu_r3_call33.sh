#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c33
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
prof() { # tag env...
  tag=$1; shift
  env "$@" rocprofv3 --kernel-trace --stats -d /tmp/rp_$tag -o t --output-format csv -- python $R/bench.py --workload config2 --no-cpu-baseline --no-host-inclusive --steps 20 --warmup 2 > $R/gpurun_out/c33/$tag.log 2>&1
  find /tmp/rp_$tag -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/c33/$tag.csv \;
  echo "== $tag"; head -9 $R/gpurun_out/c33/$tag.csv | cut -d, -f1-4 | cut -c1-150
}
prof s2split QCAT_HIP_ADAPTER_BITSLICE_MIN=1 QCAT_HIP_ABS_STAGES=2
prof s4split QCAT_HIP_ADAPTER_BITSLICE_MIN=1 QCAT_HIP_ABS_STAGES=4
prof s2fused QCAT_HIP_ADAPTER_BITSLICE_MIN=1 QCAT_HIP_ABS_STAGES=2 QCAT_HIP_ABS_NO_SPLIT=1
