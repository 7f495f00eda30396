#!/bin/bash
# round 4: timeline of a REPLAYED kit-auto call (captured graph) against a plain one
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r04_api3; mkdir -p $out
rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/tl_g -o t --output-format csv -- python $R/bench.py --workload api4000 --steps 8 --warmup 2 > $out/bench_graph.log 2>&1
python $R/tools/api_timeline.py /tmp/tl_g > $out/timeline_graph.txt 2>&1; tail -60 $out/timeline_graph.txt
QCAT_HIP_NO_GRAPH=1 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/tl_p -o t --output-format csv -- python $R/bench.py --workload api4000 --steps 8 --warmup 2 > $out/bench_plain.log 2>&1
python $R/tools/api_timeline.py /tmp/tl_p > $out/timeline_plain.txt 2>&1; tail -3 $out/timeline_plain.txt
