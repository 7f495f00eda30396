#!/bin/bash
# round 4, final validation: the whole GPU suite, smoke(), the rest of the adapter-forced fuzz sweep, the default bench line
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_final_a; mkdir -p $out
(time timeout 2400 python -m pytest tests -x -q -m gpu) > $out/tests_full.log 2>&1; echo "pytest rc=$?"; tail -5 $out/tests_full.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log
(QCAT_HIP_BITSLICE_MIN=2048 QCAT_HIP_ADAPTER_BITSLICE_MIN=1 timeout 1500 python tools/fuzz_bitslice.py 67 53) > $out/fuzz_adapter_forced_67.txt 2>&1; tail -2 $out/fuzz_adapter_forced_67.txt
