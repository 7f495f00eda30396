#!/bin/bash
# Collect the rocprofv3 evidence for one bench workload on the GPU box (run through gpurun):
#   tools/profile.sh <tag> <workload> [bench args]
# 1) kernel trace + stats, 2) PMC passes (own runs, --kernel-trace only), summaries -> gpurun_out/prof_<tag>/
set -u
tag=$1; wl=$2; shift 2
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cmd="python $R/bench.py --workload $wl --no-cpu-baseline --no-host-inclusive $*"
# 0) the same command without a profiler: the durations of the timing marks the counters will be replayed against (bench.py: stale_reason)
$cmd > $R/$out/bench_plain.log 2>/dev/null
rocprofv3 --kernel-trace --stats -d /tmp/rp_$tag/trace -o trace --output-format csv -- $cmd > $R/$out/bench_under_trace.log 2>&1
find /tmp/rp_$tag/trace -name "*kernel_stats.csv" -exec cp {} $R/$out/kernel_stats.csv \;
i=0
for pmc in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pmc -d /tmp/rp_$tag/pmc$i -o pmc --output-format csv -- $cmd > $R/$out/bench_under_pmc$i.log 2>&1
  find /tmp/rp_$tag/pmc$i -name "*counter_collection.csv" -exec cp {} $R/$out/pmc$i.csv \;
done
python $R/tools/summarize_pmc.py $R/$out > $R/$out/summary.txt 2>&1
cat $R/$out/summary.txt
