#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c23
A=qcat_amd/csrc/build/ab
bash tools/ab_run.sh gpurun_out/c23 3 --steps 8 -- $A/base.so $A/nf7_shared.so $A/nf8_shared.so
QCAT_HIP_PIPELINE_TRACE=1 timeout 900 python tools/bench_cli.py 6000000 20000 > gpurun_out/c23/bench_cli6m.json 2> gpurun_out/c23/bench_cli6m.err; grep "qcat pipeline" gpurun_out/c23/bench_cli6m.err | cut -c1-400
