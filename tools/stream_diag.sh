#!/bin/bash
# where the file loop's time goes, in a process of its own and inside bench.py (one gpurun call): stage traces + CPU seconds per stage
# (QCAT_HIP_PIPELINE_TRACE), the container's CPU-quota throttling around the streamed calls (bench.py: from_fastq.stream.host)
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/stream_diag
mkdir -p $out
cat /sys/fs/cgroup/cpu.max /sys/fs/cgroup/memory.max > $out/cgroup.txt 2>&1; nproc >> $out/cgroup.txt
QCAT_HIP_PIPELINE_TRACE=1 timeout 600 python tools/bench_stream.py 1000000 $out/standalone.json > /dev/null 2> $out/standalone.trace
python - <<'PY'
import json
d = json.load(open("gpurun_out/stream_diag/standalone.json"))
for r in d["demux"]: print("standalone", r)
print("whole", d["whole_file"])
PY
grep "CPU seconds" $out/standalone.trace | tail -1
show() { python - "$1" "$2" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
f = d["host_inclusive"]["from_fastq"]
print(sys.argv[2], f["value"], f["stream"], "whole", f["whole_file"]["value"])
PY
}
timeout 600 python bench.py --steps 3 --warmup 1 > $out/bench_default.json 2> $out/bench_default.err; show $out/bench_default.json default
timeout 600 python bench.py --steps 3 --warmup 1 --reads 1000000 > $out/bench_1m.json 2> $out/bench_1m.err; show $out/bench_1m.json reads_1m
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_nocpu.json 2> $out/bench_nocpu.err; show $out/bench_nocpu.json no_cpu_baseline
QCAT_HIP_PIPELINE_TRACE=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_trace.json 2> $out/bench_trace.err; show $out/bench_trace.json traced
grep "split of\|populate\|CPU seconds" $out/bench_trace.err | tail -12
