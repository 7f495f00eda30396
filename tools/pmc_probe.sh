#!/bin/bash
# one extra PMC pass over a bench workload: tools/pmc_probe.sh <tag> <workload> "<counters>" [bench args]
tag=$1; wl=$2; pmc=$3; shift 3
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $pmc -d /tmp/rpp_$tag -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --no-cpu-baseline --no-host-inclusive --steps 3 --warmup 1 $* > $out/log.txt 2>&1
find /tmp/rpp_$tag -name "*counter_collection.csv" -exec cp {} $out/pmc1.csv \;
python $GRAFT_REPO_ROOT/tools/summarize_pmc.py $out 2>&1 | grep -A12 "k_bs_barcode" | head -40
