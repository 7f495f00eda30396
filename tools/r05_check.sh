#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/fuzz2
(timeout 900 python tools/fuzz_tiny.py 600 1600) > gpurun_out/fuzz2/tiny.txt 2>&1; tail -1 gpurun_out/fuzz2/tiny.txt
(QCAT_HIP_BITSLICE_MIN=2048 timeout 1300 python tools/fuzz_bitslice.py 60 120) > gpurun_out/fuzz2/barcode.txt 2>&1; tail -1 gpurun_out/fuzz2/barcode.txt
(timeout 1200 python tools/fuzz_middle.py 0 40) > gpurun_out/fuzz2/middle.txt 2>&1; tail -1 gpurun_out/fuzz2/middle.txt
