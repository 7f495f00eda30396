#!/bin/bash
# round 6: the shape probe and the phase stamps of k_bs_barcode in one gpurun call
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r06_probe; mkdir -p $out
timeout 300 tools/bs_shape_probe.bin 1 > $out/shape_probe.txt 2>&1
timeout 300 tools/bs_shape_probe_norot.bin 1 > $out/shape_probe_norot.txt 2>&1
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-inclusive > $out/bench_base.json 2> $out/bench_base.err
for sel in "0 0" "0 16" "131 4" "255 24"; do
  set -- $sel
  QCAT_HIP_BS_TRACE=1 QCAT_HIP_BS_TRACE_WG=$1 QCAT_HIP_BS_TRACE_SEQ=$2 timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-inclusive > $out/trace_wg$1_seq$2.json 2> $out/trace_wg$1_seq$2.txt
done
tail -5 $out/shape_probe.txt
