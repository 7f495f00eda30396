#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c44
QCAT_HIP_DEBUG_BINS=1 QCAT_HIP_DEBUG_REDO=1 timeout 300 python bench.py --no-host-inclusive --no-cpu-baseline --steps 1 --warmup 0 > gpurun_out/c44/b.json 2> gpurun_out/c44/bins.txt
sort gpurun_out/c44/bins.txt | uniq -c | sort -rn | head -30
