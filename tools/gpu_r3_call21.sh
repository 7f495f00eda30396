#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c21
A=qcat_amd/csrc/build/ab
bash tools/ab_run.sh gpurun_out/c21 2 --steps 8 -- $A/base.so $A/nf7.so $A/nf7_shared.so $A/nf7_def2.so $A/full.so
timeout 600 python tools/prof_cli.py 2000000 > gpurun_out/c21/prof_cli.txt 2>&1; grep -A30 "run 1" gpurun_out/c21/prof_cli.txt | cut -c1-150
