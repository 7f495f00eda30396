#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c20
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $GRAFT_REPO_ROOT/gpurun_out/c20/avail.txt 2>&1
grep -o -i "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_INST[A-Z_]*\|SQC_TC_INST[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*" $GRAFT_REPO_ROOT/gpurun_out/c20/avail.txt | sort -u | tr '\n' ' '; echo
probe() {  # tag, counters
  out=$GRAFT_REPO_ROOT/gpurun_out/c20/$1; mkdir -p $out
  timeout 300 rocprofv3 --kernel-trace --pmc $2 -d /tmp/rpp_$1 -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-inclusive --steps 2 --warmup 1 > $out/log.txt 2>&1
  find /tmp/rpp_$1 -name "*counter_collection.csv" -exec cp {} $out/pmc1.csv \;
  python $GRAFT_REPO_ROOT/tools/summarize_pmc.py $out 2>&1 | grep -A8 "k_bs_barcode" | head -30
}
probe p1 "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES"
probe p2 "SQ_IFETCH SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU"
probe p3 "SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM"
