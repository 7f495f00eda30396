#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c22
A=qcat_amd/csrc/build/ab
bash tools/ab_run.sh gpurun_out/c22 2 --steps 8 -- $A/nf7_shared.so $A/w12.so $A/row1.so
cd /tmp && export TMPDIR=/tmp
probe() {  # tag lib counters
  out=$GRAFT_REPO_ROOT/gpurun_out/c22/$1; mkdir -p $out
  QCAT_HIP_LIBRARY=$GRAFT_REPO_ROOT/$2 timeout 300 rocprofv3 --kernel-trace --pmc $3 -d /tmp/rpp_$1 -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-inclusive --steps 2 --warmup 1 > $out/log.txt 2>&1
  find /tmp/rpp_$1 -name "*counter_collection.csv" -exec cp {} $out/pmc1.csv \;
  echo "== $1"; python $GRAFT_REPO_ROOT/tools/summarize_pmc.py $out 2>&1 | grep -A5 "k_bs_barcode" | head -14
}
for v in base nf7 nf7_shared full w12 row1; do probe $v $A/$v.so "SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_INSTS_VALU"; done
cd $GRAFT_REPO_ROOT; timeout 900 python tools/bench_cli.py 6000000 50000 > gpurun_out/c22/bench_cli6m.json 2> gpurun_out/c22/bench_cli6m.err; cat gpurun_out/c22/bench_cli6m.json | cut -c1-1500
