#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c4
timeout 900 bash tools/profile.sh r03b_config3 config3 --steps 3 --warmup 1 > gpurun_out/c4/profile.log 2>&1
grep -A12 "k_abs_planes" gpurun_out/prof_r03b_config3/summary.txt | head -80
grep -A12 "k_adapter_bs" gpurun_out/prof_r03b_config3/summary.txt | head -80
