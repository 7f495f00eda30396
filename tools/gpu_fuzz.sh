#!/bin/bash
# parity sweeps on the GPU box: tools/fuzz_tiny.py (the one-wave-per-alignment kernels, every intermediate) and
# tools/fuzz_bitslice.py twice over the same seeds -- the barcode kernels forced for small batches, and the adapter kernels
# forced as well (two-stage / four-stage plans by the batch's tile count);  tools/gpu_fuzz.sh [bit-sliced seeds, 120] [tiny seeds, 600]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fuzz
(timeout 900 python tools/fuzz_tiny.py 0 ${2:-600}) > gpurun_out/fuzz/tiny.txt 2>&1; tail -1 gpurun_out/fuzz/tiny.txt
(QCAT_HIP_BITSLICE_MIN=2048 timeout 1300 python tools/fuzz_bitslice.py 0 ${1:-120}) > gpurun_out/fuzz/barcode.txt 2>&1; tail -1 gpurun_out/fuzz/barcode.txt
(QCAT_HIP_BITSLICE_MIN=2048 QCAT_HIP_ADAPTER_BITSLICE_MIN=1 timeout 1300 python tools/fuzz_bitslice.py 0 ${1:-120}) > gpurun_out/fuzz/adapter_forced.txt 2>&1; tail -1 gpurun_out/fuzz/adapter_forced.txt
grep -c " ok$" gpurun_out/fuzz/tiny.txt gpurun_out/fuzz/barcode.txt gpurun_out/fuzz/adapter_forced.txt; grep -h "MISMATCH\|Error\|error" gpurun_out/fuzz/*.txt | head -5
