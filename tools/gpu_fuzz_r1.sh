#!/bin/bash
# the parity sweeps of tools/gpu_fuzz.sh under BOTH end-position rules (include/qcat_hip.h QCAT_R1_*; QCAT_R1_RULE is read by
# qcat_amd/native.py when a descriptor is built, the oracle follows the descriptor): tools/gpu_fuzz_r1.sh [bit-sliced seeds] [tiny seeds]
cd $GRAFT_REPO_ROOT
out=gpurun_out/fuzz_r1; mkdir -p $out
for rule in scalar striped; do
  export QCAT_R1_RULE=$rule
  (timeout 900 python tools/fuzz_tiny.py 0 ${2:-600}) > $out/tiny_$rule.txt 2>&1; echo "$rule tiny: $(tail -1 $out/tiny_$rule.txt)"
  (QCAT_HIP_BITSLICE_MIN=2048 timeout 1300 python tools/fuzz_bitslice.py 0 ${1:-60}) > $out/barcode_$rule.txt 2>&1; echo "$rule bit-sliced barcode: $(tail -1 $out/barcode_$rule.txt)"
  (QCAT_HIP_BITSLICE_MIN=2048 QCAT_HIP_ADAPTER_BITSLICE_MIN=1 timeout 1300 python tools/fuzz_bitslice.py 0 ${1:-60}) > $out/adapter_forced_$rule.txt 2>&1; echo "$rule bit-sliced adapter forced: $(tail -1 $out/adapter_forced_$rule.txt)"
  (timeout 900 python tools/fuzz_middle.py 0 30) > $out/middle_$rule.txt 2>&1; echo "$rule middle: $(tail -1 $out/middle_$rule.txt)"
done
grep -c " ok$" $out/*.txt; grep -h "MISMATCH\|Error\|error" $out/*.txt | head -5
